"""Single-model universal adversarial-image trainer (reference: src/attack_model.py).

Same entry point (`train(...)`, `main()`), flag names and defaults, the
same per-step order of operations, the same on-disk artefacts
(`optimized_image_iter_{k}.png` + fp32 CHW `.bin`, `mask.pt`, `mask.png`, `config.json`).
What is different underneath:
  * every pixel-space operation between `p` and `pixel_values`, and back, runs in
    libadvx_hip.so through `PixelPGD` (no torch arithmetic, no autograd on that path);
  * no per-step PNG -> disk -> PNG round trip: the quantise-error statistics come from a
    device-side reduction (the round trip equals uint8 truncation because PNG is lossless);
  * prompts are tokenised once and cached; statistics stay on the device and are read at
    logging cadence only; model parameters are frozen;
  * data parallelism: under torch.distributed every rank owns batch_size/world prompts and the
    image gradient is all-reduced once per step over RCCL (the reference has no DP).
"""
import argparse
import json
import math
import os
import random
from datetime import datetime

import numpy as np
import torch
from PIL import Image

from . import dp
from . import prompts as P
from .ops import IO_DTYPES
from .pgd import PixelPGD
from .processors import load_components


# ----------------------------------------------------------------------------- helpers
def setup_device():
    if not torch.cuda.is_available():
        raise RuntimeError("the trainer needs a ROCm device: the pixel path has no CPU fallback")
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))


def create_directory(exp_name, base_path="./runs"):
    path = os.path.join(base_path, exp_name)
    os.makedirs(path, exist_ok=True)
    return path


def save_checkpoint(image: Image.Image, tensor: torch.Tensor, path: str, iteration):
    """attack_model.py:33-36: PNG + raw float32 CHW dump of x0 + x."""
    image.save(os.path.join(path, f"optimized_image_iter_{iteration}.png"))
    tensor.detach().cpu().numpy().astype(np.float32).tofile(os.path.join(path, f"optimized_image_iter_{iteration}.bin"))


def create_mask(mask_type, mask_size, image_shape, device):
    """attack_model.py:66-84."""
    C, H, W = image_shape
    mask = torch.zeros(image_shape, device=device)
    if mask_type == "corner":
        mask[:, :mask_size, :mask_size] = 1.0
    elif mask_type == "bottom_lines":
        mask[:, -mask_size:, :] = 1.0
    elif mask_type == "random_square":
        i = random.randint(0, H - mask_size)
        j = random.randint(0, W - mask_size)
        mask[:, i:i + mask_size, j:j + mask_size] = 1.0
    else:
        mask = torch.ones(image_shape, device=device)
    return mask


def random_resized_crop_params(height, width, scale, ratio):
    """torchvision RandomResizedCrop.get_params (attack_model.py:198-202): draws from torch's
    global CPU generator like the original; restated from the published algorithm."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


# wall-clock bound of a wait of the peer exchange inside the trainers: rank 0 alone writes checkpoints and runs
# the generation probe, and although every rank waits for it at a barrier afterwards, first-step warm-up and
# allocator stalls can skew the ranks by seconds; a rank that is really gone still ends the run (check_replicas)
EXCHANGE_TIMEOUT_S = 300.0


def save_state(engine, exp_path, global_iteration, iteration, name=None, torch_rng=None):
    """True resume state (not in the reference): optimiser moments, schedules, the global RNG
    streams the shared draws come from (identical on all ranks).  Rank-local streams are re-derived on load:
    the noise seed from the rank, the prompt stream from (seed, rank, iteration).
    torch_rng: the generator state to store instead of the current one - a loop that has already drawn the NEXT iteration's
    crop window / blur sigma (draw_image_params ahead of the backward) hands in the state from before that draw, so that a
    resumed run draws the same values at the top of its first iteration."""
    path = os.path.join(exp_path, name or f"state_iter_{global_iteration}.pt")
    torch.save({"engine": {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in engine.state_dict().items()},
                "global_iteration": global_iteration, "iteration": iteration, "py_random": random.getstate(),
                "torch_rng": torch.get_rng_state() if torch_rng is None else torch_rng}, path)
    return path


def draw_image_params(use_gaussian_blur, gblur_sigma, use_local_crop, H, W, crop_scale, crop_ratio, sigma_per_step=False):
    """(blur sigma | None, crop window | None) of ONE iteration, drawn from torch's global generator in the reference's order:
    the cross trainer draws a sigma ~ U(0.1, 2) per step (crossattack_models.py:336-337, Q4) and then the window (:341-343); the
    single trainer's sigma is a constant (attack_model.py:191-194) and only the window is drawn (:307-310).  Both loops call
    this at the top of an iteration - or, when the engine fuses the next step's image kernel into this step's backward
    (PixelPGD.step_fusion), one iteration ahead, right before backward_update: nothing else draws from that generator in
    between, so the sequence of values is the same."""
    sigma = None
    if use_gaussian_blur:
        sigma = torch.empty(1).uniform_(0.1, 2.0).item() if sigma_per_step else float(gblur_sigma)
    crop = random_resized_crop_params(H, W, crop_scale, crop_ratio) if use_local_crop else None
    return sigma, crop


def saved_chain(path):
    """The kernel chain of the run that wrote `path` (None: a file from before round 4, or nothing to resume)."""
    if not path:
        return None
    return torch.load(path, map_location="cpu")["engine"].get("chain")


def load_state(engine, path):
    """-> (global_iteration, next iteration) of the run that wrote `path`."""
    sd = torch.load(path, map_location="cpu")
    engine.load_state_dict(sd["engine"])
    random.setstate(sd["py_random"])
    torch.set_rng_state(sd["torch_rng"])
    return int(sd["global_iteration"]), int(sd["iteration"]) + 1


def reseed_prompt_stream(inputs_processors, seed, rank, iteration):
    """Data parallelism: every rank draws its prompts from its own stream.  Re-seeded per iteration from
    (seed, rank, iteration), so a resumed run continues every rank's stream without having saved it."""
    for ip in inputs_processors:
        ip.rng.seed((int(seed) * 1000003 + int(rank)) * 1000003 + int(iteration))


def assert_replicas(engine, exp_path, rank, iteration, global_iteration):
    """Collective (every rank, same iteration).  If a barrier of the peer exchange gave up or the replicas
    differ, every rank writes its own state beside the checkpoints and raises dp.ReplicaError - the
    trainers' main() turns that into exit status 3 on every rank."""
    reason = dp.check_replicas(engine, engine.pg)
    if reason is None:
        return
    path = save_state(engine, exp_path, global_iteration, iteration, name=f"state_diverged_rank{rank}_iter_{iteration}.pt")
    raise dp.ReplicaError(f"rank {rank}, iteration {iteration}: {reason}; this rank's state is in {path}")


def broadcast_run_name(name):
    """Every rank must write into the SAME run directory: rank 0's time-stamped name wins."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        box = [name]
        torch.distributed.broadcast_object_list(box, src=0)
        name = box[0]
    return name


class JsonlLogger:
    """Default metrics sink (the reference logs ~20 scalars per step to wandb,
    attack_model.py:382-407); wandb is used instead when importable and requested."""

    def __init__(self, path, use_wandb=False, config=None, name=None):
        self.f = open(path, "a")
        self.wandb = None
        if use_wandb:
            try:
                import wandb
                wandb.init(project="image_attack_optimization", name=name, config=config)
                self.wandb = wandb
            except Exception as e:  # pragma: no cover - wandb absent in the build container
                print(f"wandb unavailable ({e}); logging to {path}")

    def log(self, data):
        self.f.write(json.dumps({k: (float(v) if isinstance(v, (int, float)) else v) for k, v in data.items()}) + "\n")
        self.f.flush()
        if self.wandb is not None:
            self.wandb.log(data)

    def close(self):
        self.f.close()
        if self.wandb is not None:
            self.wandb.finish()


# ------------------------------------------------------------------------------- train
def train(exp_name, img_orig, prompt, target_text, model_name, lr, num_iterations, save_steps, batch_size,
          grad_accum_steps, scheduler_step_size, scheduler_gamma, restart_num, mask_type, mask_size, clamp_method,
          epsilon, sigma, start_from_white, target_text_random, DPO_flag=False, refuse_prob=0.1,
          use_gaussian_blur=False, gblur_kernel_size=5, gblur_sigma=7, use_local_crop=False, crop_scale_min=0.6,
          crop_scale_max=1.0, crop_ratio_min=0.75, crop_ratio_max=1.33,
          # --- additions of this framework (defaults reproduce the reference behaviour)
          questions_file=None, test_questions_file=None, answers_file=None, optimizer="adamw", log_every=1,
          use_wandb=False, seed=0, base_path="./runs", components=None, return_engine=False,
          generation_probe=False, resume_from=None, pixel_io="float32", resaved_loss_every=0,
          noise_on_padding=True, suffix_only_ce=False, replica_check_every=None, exchange_transport="auto",
          exchange_timeout_s=EXCHANGE_TIMEOUT_S, unit_noise_fn=None, step_fusion=False):
    """pixel_io: "float32" hands the VLM fp32 pixel_values as the reference does; "model" lets
    the fused pair write them in model.dtype (the cast the vision tower's patch embedding applies
    first anyway) and read the half gradient directly - same numbers, half the traffic.
    replica_check_every (data parallelism): every so many iterations (default: save_steps) all ranks compare
    digests of (p, m, v) and the peer exchange's time-out word; on a mismatch every rank writes its state and
    raises dp.ReplicaError.
    unit_noise_fn (parity tests): callable(iteration, shape) -> N(0, 1) draws on the CPU that replace the in-kernel generator,
    so that a run can be laid beside one of the reference's `train()` (tests/test_gpu_trainer_vs_reference_run.py)."""
    if pixel_io not in ("float32", "model"):
        raise ValueError("pixel_io must be 'float32' or 'model'")
    if clamp_method != "tanh":
        raise NotImplementedError("Clamping method except tanh are not implemented")       # attack_model.py:186
    if DPO_flag:
        raise NotImplementedError("DPO flag is not implemented")                           # attack_model.py:278-279
    if mask_type == "random_square":
        raise NotImplementedError("random_square needs a per-step mask move the reference raises on (:295-296)")
    questions = P.load_pool(questions_file, P.DEFAULT_QUESTIONS)
    test_questions = P.load_pool(test_questions_file, P.DEFAULT_TEST_QUESTIONS)
    if target_text_random:
        target_text = P.load_pool(answers_file, P.DEFAULT_ANSWERS)                         # :147-148
    if prompt != "list":
        questions = [prompt]                                                               # :150-151

    device = setup_device()
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    if batch_size % world != 0:
        raise ValueError("batch_size must be divisible by the number of ranks")
    local_batch = batch_size // world
    exp_path = create_directory(exp_name, base_path)

    load_model_and_processor, AdvInputs, DiffProc = components or load_components(model_name)
    model, processor = load_model_and_processor(model_name, device)
    model.requires_grad_(False)                                                            # Q6: output-neutral
    adv_processor = DiffProc(processor.image_processor, device)

    if isinstance(img_orig, Image.Image):
        original_image = img_orig.convert("RGB")
    elif os.path.exists(img_orig):
        original_image = Image.open(img_orig).convert("RGB")
    elif os.path.exists(os.path.join("./images", img_orig)):
        original_image = Image.open(os.path.join("./images", img_orig)).convert("RGB")
    else:
        raise FileNotFoundError(f"Cannot find {img_orig}")
    x_0 = adv_processor.pil_to_tensor(original_image, resize=False).to(device)
    if start_from_white:
        x_0 = torch.ones_like(x_0)
    _, H, W = x_0.shape

    if mask_type is not None and mask_size is not None:
        mask = create_mask(mask_type, mask_size, x_0.shape, device)
    else:
        mask = (x_0 != 0).float()                                                          # :208
    if rank == 0:
        torch.save(mask.cpu(), os.path.join(exp_path, "mask.pt"))
        Image.fromarray((mask.permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)).save(os.path.join(exp_path, "mask.png"))

    plan = adv_processor.plan_for(H, W)
    # a resumed run stays on the chain it was started on (the one-launch chain draws its noise differently): `auto` would
    # pick by the batch size of THIS call
    chain = saved_chain(resume_from)
    engine = PixelPGD(x_0, [plan], epsilon=epsilon, lr=lr, sigma0=sigma, mask=mask,
                      fused_mode=chain if chain in ("pair", "step", "prepared") else "auto", step_fusion=step_fusion,
                      scheduler_step_size=scheduler_step_size, scheduler_gamma=scheduler_gamma,
                      grad_accum_steps=grad_accum_steps, blur_kernel=gblur_kernel_size if use_gaussian_blur else None,
                      use_crop=use_local_crop, optimizer=optimizer, seed=seed + 7919 * rank,
                      process_group=torch.distributed.group.WORLD if world > 1 else None,
                      noise_on_padding=noise_on_padding, exchange_transport=exchange_transport,
                      exchange_timeout_s=exchange_timeout_s,
                      # small batches (the reference's presets: 1-4) take the one-launch chain; it has a float32 boundary only
                      batch_hint=local_batch if pixel_io == "float32" else None)
    if pixel_io == "model" and engine.mode != "step":
        model_dtype = next(model.parameters()).dtype
        if model_dtype in IO_DTYPES:
            engine.io_dtype = model_dtype

    # shared draws (target text, crop window) come from the global generators, which every
    # rank seeds identically; prompt sampling uses a rank-local stream
    random.seed(seed)
    torch.manual_seed(seed)
    inputs_processor = AdvInputs(questions=questions, test_questions=test_questions, batch_size=local_batch,
                                 original_image=original_image, processor=processor, device=device,
                                 target_text=target_text,
                                 rng=random.Random(seed * 1000003 + rank) if world > 1 else None)
    if hasattr(inputs_processor, "bind_geometry"):
        inputs_processor.bind_geometry(adv_processor, H, W)     # index tensors from the plan, not from an HF image pass
    logger = JsonlLogger(os.path.join(exp_path, "metrics.jsonl"), use_wandb and rank == 0,
                         config=dict(learning_rate=lr, batch_size=batch_size, epsilon=epsilon, sigma=sigma), name=exp_name) \
        if rank == 0 else None

    global_iteration = 0
    accumulated_loss = 0.0
    window_iterations = window_logged = 0       # of the current accumulation window: iterations seen / iterations logged
    refuse_flag = False
    history = []
    ahead = None        # (blur sigma, crop window) of the next iteration when it was drawn ahead
    image_draw = (use_gaussian_blur, gblur_sigma, use_local_crop, H, W, (crop_scale_min, crop_scale_max),
                  (crop_ratio_min, crop_ratio_max), False)
    start_iteration = 0
    if resume_from:
        global_iteration, start_iteration = load_state(engine, resume_from)
    check_every = int(replica_check_every) if replica_check_every else int(save_steps)
    if world > 1:
        torch.distributed.barrier()          # model loading skews the ranks by far more than a step
    for iteration in range(start_iteration, num_iterations):
        if world > 1:
            reseed_prompt_stream([inputs_processor], seed, rank, iteration)
        if target_text_random:
            inputs_processor.set_target_text(random.choice(inputs_processor.target_texts))  # :283-290
        inputs = inputs_processor.get_inputs_train()                                        # :292
        if ahead is not None:
            blur_sigma, crop = ahead               # drawn one iteration ahead, for the fused step (same values, same order)
            ahead = None
        else:
            blur_sigma, crop = draw_image_params(*image_draw)
        given = None
        if unit_noise_fn is not None:
            given = [unit_noise_fn(iteration, (local_batch * plan.out_shape[0],) + tuple(plan.out_shape[1:])).to(device)]
        pixel_values = engine.forward(local_batch, unit_noises=given, blur_sigma=blur_sigma, crop=crop)[0]   # :300-321 (HIP)
        pixel_values.requires_grad_(True)
        inputs["pixel_values"] = pixel_values
        if suffix_only_ce:
            # same loss, the VLM computes the logits of the target positions only (HIP log-softmax + NLL)
            loss = inputs_processor.get_loss_suffix_only(model, inputs)
        else:
            outputs = model(**inputs)                                                       # :324 (PyTorch-ROCm)
            logits = outputs.logits[:, :-1, :]
            loss = inputs_processor.get_loss(logits)                                        # :327
        loss = -loss if refuse_flag else loss
        (loss * engine.loss_scale(0)).backward()                                            # :330-332
        rng_before, nxt = None, {}
        if getattr(engine, "step_fusion", False) and iteration + 1 < num_iterations:
            # the blur chain's backward also runs the NEXT iteration's tanh + blur (one launch, advx_image_step): it needs that
            # iteration's window now.  A checkpoint of this iteration stores the generator as it was before the draw.
            rng_before = torch.get_rng_state()
            ahead = draw_image_params(*image_draw)
            nxt = dict(next_blur_sigma=ahead[0], next_crop=ahead[1])
        stepped = engine.backward_update([pixel_values.grad], **nxt)                        # :335-346, 366-373 (HIP)
        loss_value = loss.detach()
        if stepped:
            global_iteration += 1
        window_iterations += 1
        last = iteration == num_iterations - 1
        if world > 1 and (iteration % check_every == 0 or iteration % save_steps == 0 or last):
            # before anything of this step is written: a lost peer or diverged replicas end the run on EVERY rank
            assert_replicas(engine, exp_path, rank, iteration, global_iteration)
        if rank == 0 and (iteration % log_every == 0 or iteration == num_iterations - 1):
            st = engine.stats_dict()
            ce = float(loss_value)
            total = (ce + st["img_loss"]) / grad_accum_steps
            accumulated_loss += total
            window_logged += 1
            rec = {"iteration": iteration, "global_iteration": global_iteration, "loss": total, "ce_loss": ce,
                   "image_loss": st["img_loss"], "grad norm": st["grad_norm"], "lr": engine.current_lr(),
                   "resave_error_std": st["sigma_next"], "resave_error_mean": st["qerr_mean"],
                   "resave_error_l1": st["qerr_l1"], "adversarial_mean": st["x_mean"], "adversarial_std": st["x_std"],
                   "noise_sigma": st["sigma"], "sigma": sigma,
                   # the reference logs the sample mean / std of the noise tensor it drew (:398-399); the noise here is drawn
                   # inside the kernel, so these are the generator's parameters - what those statistics estimate
                   "noise_mean": 0.0, "noise_std": st["sigma"]}
            if stepped and window_logged == window_iterations:
                # the reference's number (:349-352): the sum over the iterations since the last optimiser step.  Every one of them
                # was logged (always so with the default --log_every 1); a window the log cadence only saw in part has no such
                # key - never a sum mixed from several windows
                rec["accumulated_loss"] = accumulated_loss
            if resaved_loss_every > 0 and iteration % resaved_loss_every == 0:
                # :375-379 - the loss of the image as a PNG of it would be seen (no noise); a whole
                # extra VLM forward, so periodic here instead of every step
                with torch.no_grad():
                    probe_inputs = dict(inputs)
                    probe_inputs["pixel_values"] = engine.resaved_pixel_values(local_batch)[0]
                    rl = inputs_processor.get_loss(model(**probe_inputs).logits[:, :-1, :])
                rec["loss_resaved"] = float(rl)
            history.append(rec)
            logger.log(rec)
        if stepped:
            # the reference resets at every optimiser step, logged or not
            accumulated_loss, window_iterations, window_logged = 0.0, 0, 0
        if rank == 0 and (iteration % save_steps == 0 or iteration == num_iterations - 1):  # :410-416 (Q10 naming)
            img = engine.image()
            pil = adv_processor.tensor2pil(img)
            save_checkpoint(pil, img, exp_path, global_iteration)
            save_state(engine, exp_path, global_iteration, iteration, torch_rng=rng_before)
            if generation_probe:                                                            # :435-445
                from .train_test import run_model_test
                first_row, probe = run_model_test([model], [processor], [inputs_processor], [model_name], test_questions,
                                                  inputs_processor.target_texts[0], exp_path, iteration, pil,
                                                  adv_processors=[adv_processor])
                logger.log(dict(iteration=iteration, **probe))
        if world > 1 and (iteration % save_steps == 0 or last):
            # rank 0 alone wrote the checkpoint / ran the probe (seconds): the others wait HERE, on the host,
            # not inside the next step's exchange kernels with their wall-clock limit
            torch.distributed.barrier()
        if restart_num > 0 and (iteration + 1) % restart_num == 0 and rank == 0:
            print("restart_num has no effect on the optimised tensor in the reference (Q5); ignored")
    if rank == 0:
        img = engine.image()
        save_checkpoint(adv_processor.tensor2pil(img), img, exp_path, "final")              # :475-477
        logger.close()
    if return_engine:
        return engine, history
    return history


def build_parser():
    p = argparse.ArgumentParser(description="Train image attack model (MI355X-native pixel path).")
    p.add_argument("--exp_name", type=str, default="advx-run")
    p.add_argument("--img_orig", type=str, default="gray.png")
    p.add_argument("--prompt", type=str, default="list", help='"list" = sample from the question pool')
    p.add_argument("--target_text", type=str, default="sure, here it is!")
    p.add_argument("--model_name", type=str, default="llava-hf/llava-1.5-7b-hf")
    p.add_argument("--lr", type=float, default=1e-2)
    p.add_argument("--num_iterations", type=int, default=1000)
    p.add_argument("--save_steps", type=int, default=10)
    p.add_argument("--batch_size", type=int, default=4)
    p.add_argument("--grad_accum_steps", type=int, default=1)
    p.add_argument("--scheduler_step_size", type=int, default=100)
    p.add_argument("--scheduler_gamma", type=float, default=1.0)
    p.add_argument("--restart_num", type=int, default=0)
    p.add_argument("--mask_type", type=str, default=None, choices=["corner", "bottom_lines", "random_square"])
    p.add_argument("--mask_size", type=int, default=None)
    p.add_argument("--clamp_method", type=str, default="tanh", choices=["clamp", "tanh", "none"])
    p.add_argument("--start_from_white", action="store_true")
    p.add_argument("--target_text_random", action="store_true")
    p.add_argument("--DPO_flag", action="store_true")
    p.add_argument("--refuse_prob", type=float, default=0.0)
    p.add_argument("--epsilon", type=float, default=0.5)
    p.add_argument("--sigma", type=float, default=0.001)
    p.add_argument("--use_gaussian_blur", action="store_true")
    p.add_argument("--gblur_kernel_size", type=int, default=5)
    p.add_argument("--gblur_sigma", type=float, default=7)
    p.add_argument("--use_local_crop", action="store_true")
    p.add_argument("--crop_scale_min", type=float, default=0.6)
    p.add_argument("--crop_scale_max", type=float, default=1.0)
    p.add_argument("--crop_ratio_min", type=float, default=0.75)
    p.add_argument("--crop_ratio_max", type=float, default=1.33)
    # additions
    p.add_argument("--questions_file", type=str, default=None)
    p.add_argument("--test_questions_file", type=str, default=None)
    p.add_argument("--answers_file", type=str, default=None)
    p.add_argument("--optimizer", type=str, default="adamw", choices=["adamw", "sign"])
    p.add_argument("--log_every", type=int, default=1)
    p.add_argument("--use_wandb", action="store_true")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--generation_probe", action="store_true", help="greedy-generate the test prompts at every save step")
    p.add_argument("--resume_from", type=str, default=None, help="state_iter_*.pt written by a previous run")
    p.add_argument("--resaved_loss_every", type=int, default=0,
                   help="log loss_resaved (the reference's second forward on the re-saved image) every N iterations; 0 = off")
    p.add_argument("--no_noise_on_padding", dest="noise_on_padding", action="store_false",
                   help="keep the constant padding tiles of Mllama / Phi-3.5 exact zeros instead of adding noise to them "
                        "as the reference does; a deviation: the Llama-3.2 vision encoder does attend to its padding tiles, "
                        "tests/test_mllama_padding_visibility.py)")
    p.add_argument("--suffix_only_ce", action="store_true",
                   help="compute the logits of the target positions only (logits_to_keep) and their cross entropy in "
                        "the HIP library: same loss, no [B, S, V] logits tensor")
    p.add_argument("--pixel_io", type=str, default="float32", choices=["float32", "model"],
                   help="dtype of pixel_values at the VLM boundary (model = the VLM's own half dtype)")
    p.add_argument("--step_fusion", action="store_true",
                   help="blur runs on one rank: the backward's last launch also runs the next iteration's tanh + blur "
                        "(advx_image_step; same bits; not faster on MI355X, hence off by default)")
    add_dp_arguments(p)
    return p


def add_dp_arguments(p):
    p.add_argument("--replica_check_every", type=int, default=None,
                   help="data parallelism: compare the replicas' digests and the peer exchange's time-out word every N "
                        "iterations (default: save_steps); a mismatch ends the run with exit status 3 on every rank")
    p.add_argument("--exchange_transport", type=str, default="auto", choices=["auto", "peer", "rccl"],
                   help="transport of the per-step all-reduce of the image gradient")
    p.add_argument("--exchange_timeout_s", type=float, default=EXCHANGE_TIMEOUT_S,
                   help="wall-clock bound of a wait of the peer exchange")


def run_main(train_fn, args):
    """Shared tail of both trainers' main(): process group, ONE run directory for all ranks, exit status 3 when the
    replicas diverged or a peer was lost."""
    import sys
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    name = broadcast_run_name(f"{args.exp_name}_{datetime.now().strftime('%Y%m%d_%H%M%S')}")
    exp_path = create_directory(name)
    if rank == 0:
        with open(os.path.join(exp_path, "config.json"), "w") as f:
            json.dump(vars(args), f, indent=4)
    kw = vars(args).copy()
    kw["exp_name"] = name
    try:
        train_fn(**kw)
    except dp.ReplicaError as e:
        print(f"FATAL: {e}", file=sys.stderr, flush=True)
        sys.exit(3)


def main(argv=None):
    run_main(train, build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
