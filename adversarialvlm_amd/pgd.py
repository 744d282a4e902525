"""PixelPGD: the owned half of one PGD step, driven through the C ABI.

Host-side mirror of the per-step order of `attack_model.py:300-346,366-373` (single model)
and `crossattack_models.py:329-406,425-432` (several models):

    forward()          p -> x = eps*tanh(p) -> [blur] -> s = x0+x -> [crop] -> per model:
                       process -> repeat(B) -> + sigma*noise            (HIP)
    <VLM forward/backward under PyTorch-ROCm - not owned>
    backward_update()  per model: sum_b, un-tile, /std, resize^T ; crop^T, + imgfit',
                       blur^T, tanh' -> grad_p ; [all-reduce over the DP group] ;
                       mask, ||g||, AdamW|sign, StepLR                   (HIP)

Four kernel chains implement this, chosen at construction (`self.mode`):
  * generic  - any plan(s), blur, crop, gradient accumulation (advx_forward_multi / advx_collect[_multi] /
               advx_image_bwd_update): five to eight launches per step; with ONE plan a crop window is composed
               with the plan's own resize (advx_crop_composes: one gather each way, advx_collect_crop);
  * prepared - ONE plan (LLaVA from a non-native image, Mllama, Qwen2-VL, Phi-3.5) without
               blur/crop/accumulation: the backward leaves the next step's canvas behind, four
               launches per step (advx_prepared_fwd / advx_prepared_bwd); under data parallelism
               the tail is split around the gradient exchange (advx_prepared_bwd_dp, or
               advx_prepared_bwd_grad + all-reduce + advx_prepared_update);
  * pair     - one identity-resize LLaVA plan without blur/crop/accumulation: two launches
               per step (advx_fused_fwd / advx_fused_bwd); the form data parallelism uses,
               with the gradient all-reduce between the backward and advx_update;
  * step     - same plans on one GPU: ONE launch per step (advx_fused_step): the backward of
               step t and the forward of step t+1 run in the same kernel, so forward() of
               step t+1 only hands out the tensor that backward_update() of step t produced.
               `auto` takes it when the caller's batch_hint is at most 16 prompts (the pair is then
               bound by the host's launch rate), else the pair.
All statistics stay on the device (`self.stats`); nothing here synchronises the stream.
"""
import torch

from . import _lib as L
from . import dp, ops


class PixelPGD:
    STEP_CHAIN_MAX_BATCH = 16      # fused_mode="auto" with a batch_hint up to this many prompts: the one-launch chain

    def __init__(self, x0, plans, epsilon=0.5, lr=1e-2, sigma0=1e-3, mask=None, scheduler_step_size=100,
                 scheduler_gamma=1.0, grad_accum_steps=1, blur_kernel=None, use_crop=False, model_weights=None,
                 optimizer="adamw", cross_mode=False, betas=(0.9, 0.999), adam_eps=1e-8, weight_decay=1e-2, seed=0,
                 process_group=None, allow_fused=True, fused_mode="auto", grad_prescale=None, force_exchange=False,
                 io_dtype=torch.float32, exchange_transport="auto", noise_on_padding=True, exchange_timeout_s=5.0,
                 batch_hint=None, step_fusion=False):
        """io_dtype: dtype of the pixel_values handed to the VLM (every chain but `step`).  float32 is
        the reference's own boundary; float16 / bfloat16 emit the tensor already cast to the
        model's dtype (the cast the model's first layer would apply) and let backward_update
        read the half gradient directly.
        exchange_transport: how the data-parallel all-reduce of the image gradient travels -
        "peer" (advx_comm_*: IPC-mapped segments over xGMI, in-library kernels), "rccl"
        (torch.distributed) or "auto" (peer if it sets up and passes its self-test here).
        noise_on_padding: True = the reference's tensor, noise also on the constant padding tiles
        of Mllama / Phi-3.5; False = a deviation from it (Llama-3.2's vision encoder does attend to its padding
        tiles, tests/test_mllama_padding_visibility.py): those tiles stay exact zeros in
        pixel_values buffers the engine keeps across steps and rewrites only where an image is
        (no generator work and no traffic for 3/4 resp. 2/7 of the tensor; the tensor returned by
        forward() is then only valid until the next forward()).
        exchange_timeout_s: wall-clock bound of every wait of the peer exchange; a wait that gives up sets
        a sticky error word (`self.peer.timed_out()`, `dp.check_replicas`) and lets its kernel go on.
        batch_hint: the prompt batch forward() will be called with, if the caller knows it: fused_mode="auto" then
        picks the one-launch `step` chain for small batches on a single rank (see STEP_CHAIN_MAX_BATCH).
        step_fusion: blur chains on one rank - let backward_update(next_blur_sigma=, next_crop=) run the NEXT forward's image kernel
        inside the backward's last launch (advx_image_step, bit-identical).  Off by default: measured on MI355X the one launch
        (25.6 us at 512 x 512, kernel 9) is no faster than the two it replaces (14.1 + 10.7) - these kernels are bound by
        instruction issue, and the halo recompute adds a third more of it (profiles/r04)."""
        if not x0.is_cuda:
            raise L.AdvxError("PixelPGD needs x0 on a ROCm device (there is no CPU fallback)")
        if not isinstance(plans, (list, tuple)):
            plans = [plans]
        self.plans = list(plans)
        self.x0 = x0.detach().float().contiguous()
        dev = self.x0.device
        C, H, W = self.x0.shape
        for pl in self.plans:
            if (pl.in_h, pl.in_w) != (H, W):
                raise L.AdvxError("plan geometry does not match x0")
        self.H, self.W = H, W
        self.eps = float(epsilon)
        self.p = torch.zeros_like(self.x0)                       # attack_model.py:182
        self.m = torch.zeros_like(self.x0)
        self.v = torch.zeros_like(self.x0)
        self.grad = torch.zeros_like(self.x0)
        self.mask = (torch.ones_like(self.x0) if mask is None else mask.to(dev).float().contiguous())
        self.stats = torch.zeros(L.STATS_N, dtype=torch.float32, device=dev)
        self.blur_kernel = blur_kernel
        self.use_crop = bool(use_crop)
        self.weights = list(model_weights) if model_weights is not None else [1.0] * len(self.plans)
        self.opt_kind = {"adamw": L.OPT_ADAMW, "sign": L.OPT_SIGN}[optimizer]
        self.lr = float(lr)                                     # chained StepLR value (double, like torch)
        self.step_size, self.gamma = int(scheduler_step_size), float(scheduler_gamma)
        self.beta1, self.beta2 = float(betas[0]), float(betas[1])
        self.adam_eps, self.wd = float(adam_eps), float(weight_decay)
        self.accum = int(grad_accum_steps)
        self.cross_mode = bool(cross_mode)
        self.opt_steps = 0
        self.iteration = 0
        self.seed = int(seed)
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # run the data-parallel chain (gradient-only backward, all-reduce, separate update) even
        # for a group of one rank: lets a single GPU exercise the RCCL path
        self.exchange = self.world > 1 or bool(force_exchange)
        self.peer = None
        self.exchange_report = None
        if self.exchange:
            self.peer = dp.make_exchange(self.x0.numel(), dev, process_group, exchange_transport,
                                         timeout_s=float(exchange_timeout_s))
            self.exchange_report = dp.last_exchange_report()
        # factor every rank applies to its contribution before the SUM all-reduce: 1/world gives
        # the data-parallel average; cross-model groups pass 1/group_size (average inside a
        # model's group, sum across models - crossattack_models.py:391)
        self.prescale = (1.0 / self.world) if grad_prescale is None else float(grad_prescale)
        self.scales = dp.Scales(len(self.plans), self.weights, self.accum, self.cross_mode, self.prescale)
        self.fused = bool(allow_fused and len(self.plans) == 1 and self.plans[0].fused_supported()
                          and blur_kernel is None and not self.use_crop and self.accum == 1)
        self.upd_scratch = ops.update_scratch(self.p.numel(), dev)
        if fused_mode not in ("auto", "pair", "step", "prepared"):
            raise ValueError("fused_mode must be auto, pair, step or prepared")
        # the prepared chain: any one-stage plan (it resamples), same restrictions otherwise
        can_prepare = bool(allow_fused and len(self.plans) == 1 and self.plans[0].prepared_supported()
                           and blur_kernel is None and not self.use_crop and self.accum == 1)
        if fused_mode == "prepared":
            if not can_prepare:
                raise L.AdvxError("fused_mode='prepared' needs one one-stage plan, no blur / crop / accumulation")
            self.fused = False
            self.mode = "prepared"
        elif not self.fused:
            if fused_mode in ("pair", "step") and allow_fused:
                raise L.AdvxError(f"fused_mode='{fused_mode}' needs an identity-resize LLaVA plan without blur / crop / accumulation")
            self.mode = "prepared" if can_prepare else "generic"
        elif fused_mode == "auto":
            # measured on MI355X (profiles/r03/small_batch.log, tools/small_batch_bench.py): from 32 prompts up the
            # two-launch pair is the faster chain (34.3 vs 38.8 us per step at 64); up to 16 prompts a step of the pair
            # is two launches the host cannot issue faster than 18.7-19.3 us, and the one-launch step runs in
            # 13.8-16.7 us (0.73-0.89 of it).  The reference's presets run batch 1-4 (attack_clamp_tanh_llava.sh:30-32).
            # The step chain has a float32 boundary only and cannot host a gradient exchange.
            small = batch_hint is not None and int(batch_hint) <= self.STEP_CHAIN_MAX_BATCH
            self.mode = "step" if (small and not self.exchange and io_dtype == torch.float32) else "pair"
        else:
            if fused_mode == "step" and self.exchange:
                raise L.AdvxError("the one-launch step cannot host the gradient all-reduce: use fused_mode='pair'")
            self.mode = fused_mode
        self._set_io(io_dtype)
        if self.peer is not None and self.mode in ("pair", "prepared"):
            # the masked, all-reduced gradient of the last step lives in the exchange's recv buffer
            self.grad = self.peer.recv[:self.x0.numel()].view_as(self.x0)
        if self.fused:
            # the forward reads its sigma from slot QERR_STD (the previous image's quantise error)
            self.stats[L.STAT_QERR_STD] = float(sigma0)
            self.par = 0                  # row set holding the statistics rows of the current image
            self.rows_fwd, self.rows_step = ops.fused_step_rows(self.plans[0])
            self.img_rows = 0
            self.norm_rows = 0
            self._out_next = None
            self.fused_scratch = ops.fused_scratch(self.plans[0], dev)
            self.s_bufs = [torch.empty_like(self.x0), torch.empty_like(self.x0)]
            self.s_cur = 0               # s_bufs[s_cur] = image of the prepared / latest forward
            self.v_buf = torch.empty_like(self.x0)
            self.prepared = False
            self.s = self.s_bufs[0]
        else:
            self.stats[L.STAT_QERR_STD] = float(sigma0)         # resave_error_std, attack_model.py:261
            self.s = torch.empty_like(self.x0)
            self.argument = torch.empty_like(self.x0)
            self.garg = torch.empty_like(self.x0)
            self._collect_update = None       # whether ops.collect_update takes this engine's steps (asked on the first one)
            self.img_scratch = ops.image_scratch(H, W, blur_kernel or 0, dev)
            self.workspaces = [torch.empty(pl.workspace_floats, dtype=torch.float32, device=dev) for pl in self.plans]
            self.noise_on_padding = bool(noise_on_padding)
            self._outs = [None] * len(self.plans)     # persistent pixel_values (noise_on_padding=False)
            # step-to-step fusion of the blur chains (advx_image_step): the backward of step t also runs the first image
            # kernel of step t+1 when the caller names that step's blur sigma / crop window (backward_update(next_...)).
            # p, m, v and s are double-buffered for it: a tile recomputes its halo from the OLD state.
            self.step_fusion = bool(step_fusion and self.mode == "generic" and blur_kernel is not None and not self.exchange
                                    and self.accum == 1 and ops.image_step_supported(H, W, blur_kernel))
            if self.step_fusion:
                self._alt = dict(p=torch.zeros_like(self.x0), m=torch.zeros_like(self.x0), v=torch.zeros_like(self.x0),
                                 s=torch.empty_like(self.x0))
            if self.mode == "prepared":
                self.prep_scratch = ops.prepared_scratch(self.plans[0], dev)
                self.rows_prepare, self.rows_bwd = ops.prepared_rows(self.plans[0])
                self.s_bufs = [self.s, torch.empty_like(self.x0)]
                self.s_cur = 0
                self.par = 0
                self.rows_in = 0
                self.prepared = False
        self._last = None
        self._next_ready = None        # (blur sigma, crop window) of a forward whose image kernel already ran (image_step)
        for pl in self.plans:
            pl.upload()

    def _set_io(self, dtype):
        """One dtype for all plans, or one per plan (cross-model runs mix fp16 and bf16 models)."""
        per_plan = list(dtype) if isinstance(dtype, (list, tuple)) else [dtype] * len(self.plans)
        if len(per_plan) != len(self.plans):
            raise L.AdvxError("io_dtype: one dtype, or one per plan")
        for d in per_plan:
            ops.io_code(d)
            if d != torch.float32 and self.mode == "step":
                raise L.AdvxError(f"io_dtype={d} is not available in the one-launch step chain")
        self._io_dtype = per_plan[0] if len(set(per_plan)) == 1 else tuple(per_plan)
        if self.mode != "pair":
            for pl, d in zip(self.plans, per_plan):     # emit / collect / prepared_* follow the plan's boundary dtype
                pl.set_io(d)
            self._outs = [None] * len(self.plans)

    @property
    def io_dtype(self):
        return self._io_dtype

    @io_dtype.setter
    def io_dtype(self, dtype):
        self._set_io(dtype)

    # ------------------------------------------------------------------ scalars
    def _opt_scalars(self, apply):
        o = L.OptScalars()
        o.kind, o.apply = self.opt_kind, int(apply)
        t = self.opt_steps + 1
        lr = self.lr
        bias1 = 1.0 - self.beta1 ** t
        bias2 = 1.0 - self.beta2 ** t
        o.lr = lr
        o.decay = 1.0 - lr * self.wd
        o.w1 = 1.0 - self.beta1
        o.beta2 = self.beta2
        o.w2 = 1.0 - self.beta2
        o.bias2_sqrt = bias2 ** 0.5
        o.eps = self.adam_eps
        o.neg_step_size = -(lr / bias1)
        return o

    def _scheduler_step(self):
        # torch.optim.lr_scheduler.StepLR (chained form): multiply at every step_size-th step
        self.opt_steps += 1
        if self.opt_steps % self.step_size == 0:
            self.lr = self.lr * self.gamma

    def imgfit_scale(self):
        # single: (CE + img)/accum (attack_model.py:330); cross: img added once per model and
        # never divided (crossattack_models.py:369).  The DP pre-scale 1/world makes the
        # SUM all-reduce an average.
        return self.scales.imgfit_scale()

    def loss_scale(self, i=0):
        """Factor the caller applies to model i's loss before .backward()."""
        return self.scales.loss_scale(i)

    # ------------------------------------------------------------------ forward
    def forward(self, batches, unit_noises=None, blur_sigma=None, crop=None, use_philox=True):
        if not isinstance(batches, (list, tuple)):
            batches = [batches] * len(self.plans)
        if unit_noises is None:
            unit_noises = [None] * len(self.plans)
        elif not isinstance(unit_noises, (list, tuple)):
            unit_noises = [unit_noises]
        outs = []
        if self.fused:
            if crop is not None:
                raise L.AdvxError("this engine was built for the fused chain: construct with use_crop=True to crop")
            pl = self.plans[0]
            B = batches[0]
            shape = (B * pl.out_shape[0],) + pl.out_shape[1:]
            if (self.mode == "step" and self._out_next is not None and unit_noises[0] is None
                    and self._out_next.shape[0] == B and (use_philox or self._out_kind != "philox")):
                # already emitted by the previous backward_update (same launch as its update)
                out, self._out_next = self._out_next, None
            else:
                ph = None if (unit_noises[0] is not None or not use_philox) else (self.seed, self.iteration)
                out = ops.fused_fwd(pl, self.p, self.x0, self.eps, B, self.stats, self.fused_scratch,
                                    self.s_bufs[self.s_cur], self.v_buf, self.prepared, unit_noise=unit_noises[0],
                                    philox=ph, parity=self.par if self.mode == "step" else 0,
                                    out_dtype=self.io_dtype, step_chain_noise=self.mode == "step")
                self.prepared = True
                self.img_rows = self.rows_fwd
                self._out_next = None
            self.s = self.s_bufs[self.s_cur]
            outs.append(out.view(shape))
            self._last = dict(batches=list(batches))
            return outs
        if self.mode == "prepared":
            if crop is not None:
                raise L.AdvxError("this engine was built for the prepared chain: construct with use_crop=True to crop")
            pl, B, z = self.plans[0], batches[0], unit_noises[0]
            ph = None if (z is not None or not use_philox) else (self.seed, self.iteration)
            keep = (not self.noise_on_padding) and z is None
            buf = None
            if keep:
                if self._outs[0] is None or self._outs[0].shape[0] != B:
                    self._outs[0] = torch.zeros((B, pl.out_numel), dtype=ops._plan_dtype(pl), device=self.p.device)
                buf = self._outs[0]
            out = ops.prepared_fwd(pl, self.p, self.x0, self.eps, B, self.stats, self.prep_scratch, self.workspaces[0],
                                   self.s_bufs[self.s_cur], self.prepared, self.par, unit_noise=z, philox=ph, out=buf,
                                   keep_padding=keep)
            if not self.prepared:
                self.rows_in = self.rows_prepare
                self.prepared = True
            self.s = self.s_bufs[self.s_cur]
            outs.append(out.view((B * pl.out_shape[0],) + pl.out_shape[1:]))
            self._last = dict(batches=list(batches))
            return outs
        blur = (self.blur_kernel, blur_sigma) if self.blur_kernel is not None else None
        # one call: image kernels, the plans' resizes (one launch for all plans that read the image) and one
        # emit per plan; the statistics are reduced inside that chain, sigma is read from the device
        n = len(self.plans)
        given = any(z is not None for z in unit_noises)
        if given and not all(z is not None for z in unit_noises):
            raise L.AdvxError("unit_noises: give the noise of every plan or of none")
        keep = (not self.noise_on_padding) and not given
        bufs = None
        if keep:
            for i, (pl, B) in enumerate(zip(self.plans, batches)):
                if self._outs[i] is None or self._outs[i].shape[0] != B:
                    self._outs[i] = torch.zeros((B, pl.out_numel), dtype=ops._plan_dtype(pl), device=self.p.device)
            bufs = self._outs
        ph = None if (given or not use_philox) else (self.seed, [self.iteration * n + i for i in range(n)])
        ready, self._next_ready = self._next_ready, None
        crop_key = None if crop is None else tuple(int(c) for c in crop)
        image_ready = bool(ready is not None and blur is not None and ready == (float(blur[1]), crop_key))
        if image_ready:
            # the previous backward_update already ran this step's image kernel into the other image buffer
            self.s, self._alt["s"] = self._alt["s"], self.s
        res, arg = ops.forward_multi(self.p, self.x0, self.eps, self.stats, self.img_scratch, self.plans, batches, self.s,
                                   argument=self.argument if crop is not None else None, blur=blur, crop=crop,
                                   unit_noises=unit_noises if given else None, philox=ph, workspaces=self.workspaces,
                                   outs=bufs, keep_padding=keep, image_ready=image_ready)
        outs = [o.view((B * pl.out_shape[0],) + pl.out_shape[1:]) for o, pl, B in zip(res, self.plans, batches)]
        # one plan and a window that composes with its stage 0: the library applied both resizes as one table
        # (include/advx.h "Composed crop") - the backward then goes canvas -> image in one gather
        # (forward_multi hands back no argument exactly then: no second question to the library)
        composed = bool(crop is not None and n == 1 and arg is None)
        self._last = dict(batches=list(batches), blur=blur, crop=crop, composed=composed)
        return outs

    # ----------------------------------------------------------------- backward
    def backward_update(self, grads, next_unit_noise=None, next_batch=None, use_philox=True, next_blur_sigma=None,
                        next_crop=None):
        """grads[i] = d(loss)/d(pixel_values_i) as produced by autograd with the loss already
        multiplied by loss_scale(i).  In the one-launch `step` chain the same kernel emits the
        pixel_values of the NEXT step: `next_unit_noise` (parity mode) / Philox noise and
        `next_batch` (default: same batch) describe that emission.
        next_blur_sigma (and next_crop, for an engine built with use_crop): what the NEXT forward() will be called with.  A
        blur engine on one rank (`self.step_fusion`) then runs that forward's image kernel inside this call's last launch
        (advx_image_step); a next forward() with other arguments simply recomputes.  Same bits either way."""
        if not isinstance(grads, (list, tuple)):
            grads = [grads]
        st = self._last
        if st is None:
            raise L.AdvxError("backward_update called before forward")
        take_step = (self.iteration + 1) % self.accum == 0        # attack_model.py:343
        # accumulate into p.grad across iterations only in the single-model trainer
        first_of_window = (self.iteration % self.accum == 0)
        accumulate = (not self.cross_mode) and (not first_of_window)
        opt = self._opt_scalars(take_step)
        if self.fused:
            pl, B = self.plans[0], st["batches"][0]
            nxt = 1 - self.s_cur
            if self.mode == "step":
                Bn = int(next_batch) if next_batch is not None else B
                if Bn != B:
                    raise L.AdvxError("the one-launch step emits the next batch with the size of the current one")
                out_next = torch.empty((B, pl.out_numel), dtype=torch.float32, device=self.p.device)
                ph = None if (next_unit_noise is not None or not use_philox) else (self.seed, self.iteration + 1)
                ops.fused_step(pl, grads[0], B, self.p, self.x0, self.eps, self.imgfit_scale(), self.mask, self.m, self.v,
                               self.grad, opt, out_next, self.s_bufs[nxt], self.v_buf, self.par, self.img_rows,
                               self.norm_rows, self.stats, self.fused_scratch, unit_noise_next=next_unit_noise, philox=ph)
                self.par = 1 - self.par
                self.img_rows = self.rows_step
                self.norm_rows = self.rows_step
                self._out_next = out_next
                self._out_kind = "given" if next_unit_noise is not None else ("philox" if use_philox else "none")
            elif not self.exchange:
                ops.fused_bwd(pl, grads[0], B, self.p, self.x0, self.eps, self.imgfit_scale(), self.grad, self.stats,
                              self.fused_scratch, mask=self.mask, m=self.m, v=self.v, opt=opt,
                              s_next=self.s_bufs[nxt], v_buf=self.v_buf)
            elif self.peer is not None:
                # backward, peer all-reduce (rank-ordered sums over xGMI) and update: one call,
                # five launches, nothing on the host in between
                ops.fused_bwd_dp(pl, self.peer, grads[0], B, self.p, self.x0, self.eps, self.imgfit_scale(), self.stats,
                                 self.fused_scratch, self.mask, self.m, self.v, opt, self.s_bufs[nxt], self.v_buf)
            else:
                ops.fused_bwd(pl, grads[0], B, self.p, self.x0, self.eps, self.imgfit_scale(), self.grad, self.stats,
                              self.fused_scratch)
                # one exchange per step: the shared image gradient (P_in*4 bytes) over RCCL/xGMI
                dp.allreduce_image_grad_(self.grad, self.pg)
                # one launch: mask, ||g|| partials, optimiser, s/v of the next forward
                ops.fused_update(pl, self.p, self.m, self.v, self.grad, self.mask, self.x0, self.eps, opt,
                                 self.s_bufs[nxt], self.v_buf, self.fused_scratch)
            self.s_cur = nxt          # the next forward's image goes to the other buffer: image() stays valid
        elif self.mode == "prepared":
            pl, B = self.plans[0], st["batches"][0]
            nxt = 1 - self.s_cur
            if not self.exchange:
                ops.prepared_bwd(pl, grads[0], B, self.p, self.x0, self.eps, self.imgfit_scale(), self.mask, self.m, self.v,
                                 self.grad, opt, self.s_bufs[nxt], self.rows_in, self.par, self.stats, self.prep_scratch,
                                 self.workspaces[0])
            elif self.peer is not None:
                ops.prepared_bwd_dp(pl, self.peer, grads[0], B, self.p, self.x0, self.eps, self.imgfit_scale(), self.mask,
                                    self.m, self.v, opt, self.s_bufs[nxt], self.rows_in, self.par, self.stats,
                                    self.prep_scratch, self.workspaces[0])
            else:
                ops.prepared_bwd_grad(pl, grads[0], B, self.p, self.x0, self.eps, self.imgfit_scale(), self.grad,
                                      self.rows_in, self.par, self.stats, self.prep_scratch, self.workspaces[0])
                dp.allreduce_image_grad_(self.grad, self.pg)
                ops.prepared_update(pl, self.p, self.m, self.v, self.grad, self.mask, self.x0, self.eps, opt,
                                    self.s_bufs[nxt], self.par, self.stats, self.prep_scratch, self.workspaces[0])
            self.par = 1 - self.par
            self.rows_in = self.rows_bwd
            self.s_cur = nxt
        else:
            if (len(self.plans) == 1 and st["blur"] is None and not (self.exchange and take_step)
                    and (st["crop"] is None or st.get("composed")) and self._collect_update is not False):
                # one plan, no blur, nothing between the backward and the optimiser: the transposed resize of stage 0
                # (through the composed window's table, if any) inside the optimiser's launch - where the library offers it
                pl, g, B = self.plans[0], grads[0], st["batches"][0]
                done = ops.collect_update(pl, g.reshape(B, pl.out_numel), B, self.p, self.s, self.eps, self.imgfit_scale(),
                                          self.grad, self.mask, self.m, self.v, opt, self.stats, self.img_scratch,
                                          self.upd_scratch, crop=st["crop"], accumulate=accumulate, finalize_norm=False,
                                          workspace=self.workspaces[0], if_supported=True)
                if st["crop"] is None:
                    self._collect_update = done is not None       # without a window the answer never changes
            else:
                done = None
            if done is not None:
                self._norm_pending = True
                if take_step:
                    self._scheduler_step()
                self.iteration += 1
                self._last = None
                return take_step
            if len(self.plans) > 1:
                # batch reductions, then every plan's transposed resize summed in one launch
                ops.collect_multi(self.plans, [g.reshape(B, pl.out_numel) for pl, g, B in zip(self.plans, grads, st["batches"])],
                                  st["batches"], grad_argument=self.garg, workspaces=self.workspaces)
            elif st.get("composed"):
                pl, g, B = self.plans[0], grads[0], st["batches"][0]
                ops.collect_crop(pl, g.reshape(B, pl.out_numel), B, st["crop"], self.img_scratch, grad_s=self.garg,
                                 workspace=self.workspaces[0])
                st = dict(st, crop=None)            # self.garg is the gradient w.r.t. the image already
            else:
                pl, g, B = self.plans[0], grads[0], st["batches"][0]
                ops.collect(pl, g.reshape(B, pl.out_numel), B, grad_argument=self.garg, workspace=self.workspaces[0])
            nxt_crop = None if next_crop is None else tuple(int(c) for c in next_crop)
            if (getattr(self, "step_fusion", False) and take_step and not accumulate and next_blur_sigma is not None
                    and st["blur"] is not None and st["crop"] is None and (nxt_crop is not None) == self.use_crop
                    and (nxt_crop is None or (len(self.plans) == 1 and ops.crop_composes(self.plans[0], self.H, self.W, nxt_crop)))):
                # blur^T, update AND the next step's tanh + blur in one launch (every tile recomputes its halo); the state
                # goes to the other buffers
                a = self._alt
                ops.image_step(self.p, self.m, self.v, a["p"], a["m"], a["v"], self.s, self.garg, self.eps, self.imgfit_scale(),
                               self.grad, self.mask, opt, self.img_scratch, self.upd_scratch, self.x0, a["s"], st["blur"],
                               float(next_blur_sigma), next_crop=nxt_crop, next_plan=self.plans[0] if nxt_crop is not None else None)
                self.p, a["p"] = a["p"], self.p
                self.m, a["m"] = a["m"], self.m
                self.v, a["v"] = a["v"], self.v
                self._next_ready = (float(next_blur_sigma), nxt_crop)      # self.s stays the image of THIS step until then
                self._norm_pending = True
                self._scheduler_step()
                self.iteration += 1
                self._last = None
                return take_step
            if not (self.exchange and take_step):
                # nothing between the image-level backward and the optimiser: one call, the tanh backward
                # (and the crop's transposed resize) inside the optimiser's launch
                ops.image_bwd_update(self.p, self.s, self.garg, self.eps, self.imgfit_scale(), self.grad, self.mask, self.m,
                                     self.v, opt, self.stats, self.img_scratch, self.upd_scratch, blur=st["blur"],
                                     crop=st["crop"], accumulate=accumulate, finalize_norm=False)
                self._norm_pending = True         # ||g|| is reduced when the statistics are read
                if take_step:
                    self._scheduler_step()
                self.iteration += 1
                self._last = None
                return take_step
            ops.image_bwd(self.p, self.s, self.garg, self.eps, self.imgfit_scale(), self.grad, self.img_scratch,
                          blur=st["blur"], crop=st["crop"], accumulate=accumulate)
            if self.exchange and take_step:
                # The reduction is linear, so a gradient-accumulation window is exchanged once,
                # at its end (intermediate grad norms are then rank-local).
                if self.peer is not None:
                    self.peer.all_reduce_(self.grad)
                else:
                    dp.allreduce_image_grad_(self.grad, self.pg)
            ops.update(self.p, self.m, self.v, self.grad, self.mask, opt, self.stats, self.upd_scratch)
        if take_step:
            self._scheduler_step()
        self.iteration += 1
        self._last = None
        return take_step

    # ------------------------------------------------- hipGraph replay of the pair
    def make_schedule(self, n_steps):
        """Per-step scalars of the next `n_steps` optimiser steps in device memory (Philox offsets are the step
        indices, the AdamW / StepLR scalars are derived here in double exactly as backward_update derives them):
        what forward_sched / backward_update_sched read instead of kernel arguments, so that their launches can be
        captured once and replayed (torch.cuda.graph).  Pair chain, one rank, in-kernel noise."""
        if self.mode != "pair" or self.exchange:
            raise L.AdvxError("make_schedule: the replayable form exists for the single-rank pair chain")
        keep = (self.lr, self.opt_steps)
        table = []
        for _ in range(int(n_steps)):
            table.append(self._opt_scalars(True))
            self._scheduler_step()
        self.lr, self.opt_steps = keep
        return ops.make_sched(table, self.iteration, self.p.device)

    def forward_sched(self, batch, sched, out):
        """forward() of the pair into the caller's `out` [batch, out_numel], Philox offset from `sched`."""
        if not self.prepared:
            raise L.AdvxError("forward_sched needs a prepared engine: run one eager forward() first")
        ops.fused_fwd_sched(self.plans[0], self.p, self.x0, self.eps, batch, self.seed, out, self.s_bufs[self.s_cur], self.v_buf,
                            self.stats, self.fused_scratch, sched)
        self.s = self.s_bufs[self.s_cur]
        self._last = dict(batches=[batch])
        return out

    def backward_update_sched(self, grad, sched):
        """backward_update() of the pair with the optimiser scalars of the step from `sched`.  Host-side counters
        (iteration, scheduler) are NOT advanced here - a replay would not advance them either: call
        `advance(steps)` after running or replaying."""
        B = self._last["batches"][0] if self._last else grad.shape[0]
        nxt = 1 - self.s_cur
        ops.fused_bwd_sched(self.plans[0], grad, B, self.p, self.x0, self.eps, self.imgfit_scale(), self.mask, self.m, self.v,
                            self.grad, self.opt_kind, self.s_bufs[nxt], self.v_buf, self.stats, self.fused_scratch, sched)
        self.s_cur = nxt
        self._last = None

    def advance(self, steps):
        """Bring the host-side counters in line with `steps` steps taken through a schedule (eagerly or by replay)."""
        for _ in range(int(steps)):
            self._scheduler_step()
            self.iteration += 1

    # ------------------------------------------------------------------ readout
    def stats_dict(self):
        """Synchronising readout of the device statistics (logging cadence only).

        After backward_update() of step t every entry refers to step t (image statistics of
        s_t, sigma_next = the noise sigma step t+1 will use, ||grad_t||), like the values the
        reference logs at the end of an iteration.  In the fused chain only the pending
        gradient-norm reduction is flushed; the statistics of the already prepared NEXT image
        stay pending until its forward."""
        if self.mode == "step":
            ops.fused_step_flush(self.plans[0], self.par, self.norm_rows, self.stats, self.fused_scratch)
        elif self.mode == "pair":
            ops.fused_flush(self.plans[0], self.stats, self.fused_scratch)
        self._flush_norm()
        v = self.stats.tolist()
        return dict(sigma=v[L.STAT_SIGMA], sigma_next=v[L.STAT_QERR_STD], qerr_mean=v[L.STAT_QERR_MEAN],
                    qerr_l1=v[L.STAT_QERR_L1], img_loss=v[L.STAT_IMGFIT], x_mean=v[L.STAT_X_MEAN],
                    x_std=v[L.STAT_X_STD], grad_norm=v[L.STAT_GRAD_NORM])

    def _flush_norm(self):
        if getattr(self, "_norm_pending", False):
            ops.update_flush(self.p.numel(), self.stats, self.upd_scratch)
            self._norm_pending = False

    def current_lr(self):
        return self.lr

    # ------------------------------------------------------------- checkpoint / resume
    def state_dict(self):
        """Everything needed to continue the run bit-for-bit (the reference saves only the image,
        attack_model.py:414-416; SURVEY 8f row 2 asks for true resume)."""
        if self.fused:
            if self.mode == "step":
                ops.fused_step_flush(self.plans[0], self.par, self.norm_rows, self.stats, self.fused_scratch)
            else:
                ops.fused_flush(self.plans[0], self.stats, self.fused_scratch)
        self._flush_norm()
        # `chain`: the kernel chain that wrote the state.  The one-launch `step` chain addresses the noise generator
        # differently from the others (ADVX_PHILOX_STEP_CHAIN): a run only continues bit for bit on the chain it started on
        return dict(p=self.p.clone(), m=self.m.clone(), v=self.v.clone(), grad=self.grad.clone(), stats=self.stats.clone(),
                    lr=self.lr, opt_steps=self.opt_steps, iteration=self.iteration, seed=self.seed, chain=self.mode)

    def load_state_dict(self, sd):
        if self._last is not None:
            raise L.AdvxError("load_state_dict between forward and backward_update")
        saved_chain = sd.get("chain")                # absent in files written before round 4: nothing to compare
        if saved_chain is not None and saved_chain != self.mode:
            raise L.AdvxError(f"this state was written by the '{saved_chain}' chain and this engine runs the '{self.mode}' chain "
                              "(another batch size under fused_mode='auto', or another fused_mode): the run would go on with "
                              f"another noise stream instead of continuing bit for bit - construct the engine with "
                              f"fused_mode='{saved_chain}' (the trainers do so on --resume_from)")
        for k in ("p", "m", "v", "grad", "stats"):
            getattr(self, k).copy_(sd[k].to(self.p.device))
        self.lr, self.opt_steps, self.iteration = float(sd["lr"]), int(sd["opt_steps"]), int(sd["iteration"])
        self._norm_pending = False                 # the loaded statistics are complete
        self._next_ready = None                    # nothing of the next forward has run on the loaded state
        # the noise seed is NOT taken from the file: under data parallelism only rank 0 writes it, and every
        # rank keeps the stream it was constructed with (seed + 7919 * rank in the trainers)
        if self.fused:
            # nothing prepared, nothing pending: the next forward re-derives s / v from p
            self.prepared = False
            self._out_next = None
            self.img_rows = self.norm_rows = 0
            self.fused_scratch.zero_()
        if self.mode == "prepared":
            self.prepared = False
            self.par = 0
            self.rows_in = 0

    def image(self):
        """x0 + x of the most recent forward (what the reference checkpoints)."""
        return self.s

    def resaved_pixel_values(self, batches):
        """pixel_values of the re-saved image (attack_model.py:368-376): the image of the most
        recent forward after the uint8 round trip, processed and repeated, WITHOUT noise - the
        input of the reference's `loss_resaved` forward.  Log-only path, not pipelined."""
        if not isinstance(batches, (list, tuple)):
            batches = [batches] * len(self.plans)
        q = ops.quantise(self.image())
        outs = []
        for pl, B in zip(self.plans, batches):
            out = ops.emit(pl, q, B)
            outs.append(out.view((B * pl.out_shape[0],) + pl.out_shape[1:]))
        return outs
