"""The PGD step of the reference trainers restated with torch autograd on CPU
(oracle side; test infrastructure only).

Follows the per-step order of `attack_model.py:276-373` (single model) and
`crossattack_models.py:301-432` (several models, gradients summed).  The VLM is
abstracted away: the caller either supplies the upstream gradient d(CE)/d(pixel_values)
directly (isolated pixel path, the benchmark's workload) or a callable that maps
pixel_values to a scalar loss (end-to-end parity on a tiny random model).

Randomness is an INPUT here (unit normal noise, blur sigma, crop window) so that the HIP
path can be compared on identical draws; RNG-stream parity with torch is not a goal.
"""
import torch

from . import pixel_ops as P


class PGDOracle:
    def __init__(self, x0, processors, epsilon=0.5, lr=1e-2, sigma0=1e-3, mask=None,
                 scheduler_step_size=100, scheduler_gamma=1.0, grad_accum_steps=1,
                 blur_kernel=None, model_weights=None, optimizer="adamw", cross_mode=False):
        if not isinstance(processors, (list, tuple)):
            processors = [processors]
        self.x0 = x0.clone().float()
        self.procs = list(processors)
        self.eps = epsilon
        self.p = torch.zeros_like(self.x0, requires_grad=True)          # attack_model.py:182
        self.optimizer_kind = optimizer
        if optimizer == "adamw":
            self.opt = torch.optim.AdamW([self.p], lr=lr)               # :184 (wd 0.01 default)
        elif optimizer == "sign":
            self.opt = None                                             # not in the reference
        else:
            raise ValueError(optimizer)
        self.lr0 = lr
        self.step_size, self.gamma = scheduler_step_size, scheduler_gamma
        self.sched = (torch.optim.lr_scheduler.StepLR(self.opt, scheduler_step_size, scheduler_gamma)
                      if self.opt is not None else None)                # :216
        self.opt_steps = 0
        self.mask = torch.ones_like(self.x0) if mask is None else mask.float()
        self.sigma = torch.tensor(float(sigma0))                        # :261 resave_error_std
        self.accum = grad_accum_steps
        self.blur_kernel = blur_kernel
        self.weights = list(model_weights) if model_weights is not None else [1.0] * len(self.procs)
        self.iteration = 0
        self._graph = None
        # crossattack_models.py quirks: the per-model loss is NOT divided by
        # grad_accum_steps (:369) and p.grad is zeroed at the start of every iteration
        # (:349-350, :384, :391) so "accumulation" only changes the optimiser cadence.
        self.cross_mode = cross_mode

    # ---- forward half of the step: p -> list of pixel_values [B_i, ...]
    def forward(self, batches, unit_noises=None, blur_sigma=None, crop=None):
        if not isinstance(batches, (list, tuple)):
            batches = [batches] * len(self.procs)
        if unit_noises is None:
            unit_noises = [None] * len(self.procs)
        elif not isinstance(unit_noises, (list, tuple)):
            unit_noises = [unit_noises]
        x = P.tanh_reparam(self.p, self.eps)                            # :300
        if self.blur_kernel is not None:
            x = P.gaussian_blur(x, self.blur_kernel, blur_sigma)        # :303-304
        s = self.x0 + x
        H, W = s.shape[1:]
        arg = P.resized_crop(s, *crop, (H, W)) if crop is not None else s   # :307-312
        pvs, singles = [], []
        for proc, B, z in zip(self.procs, batches, unit_noises):
            pv1 = proc.process(arg)["pixel_values"]                     # :314
            singles.append(pv1)
            pvs.append(P.broadcast_with_noise(pv1, B, z, self.sigma))   # :316-321
        img_loss = P.image_fit_loss(self.x0, x)                         # :329
        self._graph = dict(x=x, s=s, pvs=pvs, img_loss=img_loss)
        self.last_single = [t.detach() for t in singles]
        return [t.detach() for t in pvs]

    # ---- backward half: upstream grads (or loss callables) -> masked grad, update, sigma
    def backward_update(self, upstreams=None, loss_fns=None, negate=False):
        g = self._graph
        n = len(self.procs)
        total = 0.0
        model_losses = []
        for i in range(n):
            if loss_fns is not None:
                li = loss_fns[i](g["pvs"][i])
            else:
                li = (g["pvs"][i] * upstreams[i]).sum()                 # linear surrogate of CE
            model_losses.append(float(li.detach()))
            li = -li if negate else li                                  # :328 (refuse flag)
            # crossattack_models.py:369 adds img_loss once per model; with one model this
            # is attack_model.py:330.
            total = total + self.weights[i] * li + g["img_loss"]
        if self.cross_mode:
            if self.p.grad is not None:
                self.p.grad.zero_()                                     # cross :349-350
        else:
            total = total / self.accum                                  # :330
        total.backward()                                                # :332
        with torch.no_grad():
            self.p.grad.mul_(self.mask)                                 # :336
            grad_norm = self.p.grad.norm().item()                       # :340
            grad_snapshot = self.p.grad.clone()
        stepped = False
        if (self.iteration + 1) % self.accum == 0:                      # :343-346
            if self.opt is not None:
                self.opt.step()
                self.opt.zero_grad()
                self.sched.step()
            else:
                with torch.no_grad():
                    lr = self.lr0 * self.gamma ** (self.opt_steps // self.step_size)
                    # sign with a dead zone below the smallest normal fp32: at the 2^-149 level two correctly rounded
                    # evaluations of the gradient differ between 0 and 1.4e-45 (separable vs product-kernel blur), and a
                    # bare sign() would make that ulp a whole step - same rule as csrc/advx_kernels.h sign_direction()
                    gp = self.p.grad
                    gp = torch.where(gp.abs() >= torch.finfo(torch.float32).tiny, gp, torch.zeros_like(gp))
                    self.p.sub_(lr * torch.sign(gp))
                    self.p.grad = None
            self.opt_steps += 1
            stepped = True
        with torch.no_grad():                                           # :366-373
            s = g["s"].detach()
            std, mean, l1 = P.quantise_error_stats(s)
            self.sigma = std
        self.iteration += 1
        x = g["x"].detach()
        out = dict(model_losses=model_losses, img_loss=float(g["img_loss"].detach()),
                   grad_norm=grad_norm, grad=grad_snapshot, stepped=stepped,
                   sigma_next=float(std), qerr_mean=float(mean), qerr_l1=float(l1),
                   x_mean=float(x.mean()), x_std=float(x.std()), s=s)
        self._graph = None
        return out

    def current_lr(self):
        if self.opt is not None:
            return self.opt.param_groups[0]["lr"]
        return self.lr0 * self.gamma ** (self.opt_steps // self.step_size)
