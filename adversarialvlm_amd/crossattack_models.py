"""Multi-model universal trainer (reference: src/crossattack_models.py).

Per step the shared image x0 + eps*tanh(p) is processed by every model's plugin, each
model contributes  w_i * CE_i + image_fit_loss  (the image-fit term once PER MODEL,
crossattack_models.py:369), the gradients wrt p are SUMMED (:391), masked, and AdamW steps.

Placement
  * one process: all models on the local device, visited serially like the reference
    (which walks cuda:0..cuda:n-1 inside one process, :352-384);
  * torch.distributed with world = n_models * k: rank r runs model r mod n_models on its own
    GPU with batch_size/k prompts; the models run CONCURRENTLY and one RCCL all-reduce(sum)
    of the image gradient per step replaces the reference's peer copies + stack().sum()
    (every rank pre-scales by 1/k so that a model's group averages and the groups add up).
Reference quirks kept: the perturbation scale is `attack_norm` (0.5), not
`--epsilon` (Q3); the blur sigma is redrawn U(0.1, 2) every step (Q4); gradient accumulation
only changes the optimiser cadence because p.grad is rebuilt each iteration; DPO_flag swaps
the target for a refusal string without negating the loss (Q8).
"""
import argparse
import os
import random

import numpy as np
import torch
from PIL import Image

from . import prompts as P
from .attack_model import (EXCHANGE_TIMEOUT_S, JsonlLogger, add_dp_arguments, assert_replicas, create_directory,
                           create_mask, draw_image_params, load_state, reseed_prompt_stream, run_main,
                           save_checkpoint, save_state, setup_device)
from .pgd import PixelPGD
from .processors import load_components


def pil_to_tensor(image: Image.Image, do_convert_rgb: bool = True, resize: bool = False) -> torch.Tensor:
    """crossattack_models.py:106-123."""
    if do_convert_rgb:
        image = image.convert("RGB")
    if resize:
        raise NotImplementedError("Resizing is not universal for models!!!")
    return torch.tensor(np.array(image).astype(np.float32) / 255).permute(2, 0, 1)


def train(exp_name, img_orig, prompt, target_text, model_names, lr, num_iterations, save_steps, batch_size,
          grad_accum_steps, scheduler_step_size, scheduler_gamma, restart_num, mask_type, mask_size, clamp_method,
          epsilon, sigma, start_from_white, target_text_random, DPO_flag=True, refuse_prob=0.1, model_weights=None,
          attack_norm=0.5, use_gaussian_blur=False, gblur_kernel_size=5, use_local_crop=False, crop_scale_min=0.6,
          crop_scale_max=1.0, crop_ratio_min=0.75, crop_ratio_max=1.33,
          questions_file=None, test_questions_file=None, answers_file=None, log_every=1, use_wandb=False, seed=0,
          base_path="./runs", return_engine=False, resaved_loss_every=0, noise_on_padding=True,
          suffix_only_ce=False, pixel_io="float32", components=None, generation_probe=False, resume_from=None,
          replica_check_every=None, exchange_transport="auto", exchange_timeout_s=EXCHANGE_TIMEOUT_S, unit_noise_fn=None,
          step_fusion=False):
    """components: optional {model_name: (load_model_and_processor, AdvInputs, DiffProc)} overriding the registry
    (tests).  generation_probe / resume_from / replica_check_every: as in attack_model.train.
    unit_noise_fn (parity tests): callable(iteration, model index, shape) -> N(0, 1) draws on the CPU that replace the in-kernel
    generator (tests/test_gpu_trainer_vs_reference_run.py lays a run beside one of the reference's `train()`)."""
    if clamp_method != "tanh":
        raise NotImplementedError("Clamping method except tanh are not implemented yet.")
    if mask_type == "random_square":
        raise NotImplementedError("Dynamic random-square mask updating not implemented.")
    questions = P.load_pool(questions_file, P.DEFAULT_QUESTIONS)
    test_questions = P.load_pool(test_questions_file, P.DEFAULT_TEST_QUESTIONS)
    if target_text_random:
        target_text = P.load_pool(answers_file, P.DEFAULT_ANSWERS)
    if prompt != "list":
        questions = [prompt]
    n_models = len(model_names)
    if model_weights is None:
        model_weights = [1.0] * n_models
    elif len(model_weights) != n_models:
        raise ValueError("The length of model_weights must match the number of model_names.")

    device = setup_device()
    dist_on = torch.distributed.is_initialized()
    world = torch.distributed.get_world_size() if dist_on else 1
    rank = torch.distributed.get_rank() if dist_on else 0
    if world > 1:
        if world % n_models != 0:
            raise ValueError("world size must be a multiple of the number of models")
        group_size = world // n_models
        if batch_size % group_size != 0:
            raise ValueError("batch_size must be divisible by the ranks per model")
        my_models = [rank % n_models]
        local_batch = batch_size // group_size
        prescale = 1.0 / group_size
    else:
        my_models = list(range(n_models))
        local_batch = batch_size
        prescale = 1.0
    exp_path = create_directory(exp_name, base_path)

    if isinstance(img_orig, Image.Image):
        original_image = img_orig.convert("RGB")
    elif os.path.exists(img_orig):
        original_image = Image.open(img_orig).convert("RGB")
    elif os.path.exists(os.path.join("./images", img_orig)):
        original_image = Image.open(os.path.join("./images", img_orig)).convert("RGB")
    else:
        raise FileNotFoundError(f"Cannot find {img_orig}")

    models, processors, adv_processors, inputs_processors = [], [], [], []
    for i in my_models:
        load_model_and_processor, AdvInputs, DiffProc = (components or {}).get(model_names[i]) or load_components(model_names[i])
        model, processor = load_model_and_processor(model_names[i], device)
        model.requires_grad_(False)
        models.append(model)
        processors.append(processor)
        adv_processors.append(DiffProc(processor.image_processor, device))
        inputs_processors.append(AdvInputs(questions=questions, test_questions=test_questions, batch_size=local_batch,
                                           original_image=original_image, processor=processor, device=device,
                                           target_text=target_text,
                                           rng=random.Random(seed * 1000003 + rank) if world > 1 else None))
    # the shared draws (target text, refusal coin, blur sigma, crop window) come from the global generators: seed
    # them AFTER the models are loaded - under data parallelism every rank loads a different model, and parameter
    # initialisation may consume the global torch generator by a different amount on each rank
    random.seed(seed)
    torch.manual_seed(seed)
    x_0 = pil_to_tensor(original_image, do_convert_rgb=adv_processors[0].do_convert_rgb).to(device)
    if start_from_white:
        x_0 = torch.ones_like(x_0)
    _, H, W = x_0.shape
    if mask_type is not None and mask_size is not None:
        mask = create_mask(mask_type, mask_size, x_0.shape, device)
    else:
        mask = (x_0 != 0).float()
    if rank == 0:
        torch.save(mask.cpu(), os.path.join(exp_path, "mask.pt"))
        Image.fromarray((mask.permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)).save(os.path.join(exp_path, "mask.png"))

    plans = [ap.plan_for(H, W) for ap in adv_processors]
    for ip, ap in zip(inputs_processors, adv_processors):
        if hasattr(ip, "bind_geometry"):
            ip.bind_geometry(ap, H, W)                           # index tensors from the plan, not from an HF image pass
    engine = PixelPGD(x_0, plans, epsilon=attack_norm, lr=lr, sigma0=0.001, mask=mask,                   # :299 hard-codes 0.001
                      scheduler_step_size=scheduler_step_size, scheduler_gamma=scheduler_gamma,
                      grad_accum_steps=grad_accum_steps, blur_kernel=gblur_kernel_size if use_gaussian_blur else None,
                      use_crop=use_local_crop, model_weights=[model_weights[i] for i in my_models], cross_mode=True,
                      seed=seed + 7919 * rank, allow_fused=(len(plans) == 1),   # one model on this rank: pipelined chains
                      process_group=torch.distributed.group.WORLD if world > 1 else None, grad_prescale=prescale,
                      noise_on_padding=noise_on_padding, exchange_transport=exchange_transport,
                      exchange_timeout_s=exchange_timeout_s, step_fusion=step_fusion)
    if pixel_io == "model":
        # every model receives pixel_values in its own dtype (Qwen2-VL runs bf16, the others fp16)
        from .ops import IO_DTYPES
        dts = [next(m.parameters()).dtype for m in models]
        engine.io_dtype = [d if d in IO_DTYPES else torch.float32 for d in dts]
    elif pixel_io != "float32":
        raise ValueError("pixel_io must be 'float32' or 'model'")
    logger = JsonlLogger(os.path.join(exp_path, "metrics.jsonl"), use_wandb and rank == 0, name=exp_name) if rank == 0 else None

    global_iteration = 0
    history = []
    start_iteration = 0
    if resume_from:
        global_iteration, start_iteration = load_state(engine, resume_from)
    check_every = int(replica_check_every) if replica_check_every else int(save_steps)
    if world > 1:
        torch.distributed.barrier()          # model loading skews the ranks by far more than a step
    ahead = None        # (blur sigma, crop window) of the next iteration when it was drawn ahead
    image_draw = (use_gaussian_blur, None, use_local_crop, H, W, (crop_scale_min, crop_scale_max),
                  (crop_ratio_min, crop_ratio_max), True)
    accumulated_loss = 0.0
    window_iterations = window_logged = 0       # of the current accumulation window: iterations seen / iterations logged
    for iteration in range(start_iteration, num_iterations):
        if world > 1:
            reseed_prompt_stream(inputs_processors, seed, rank, iteration)
        if DPO_flag or target_text_random:                                                  # :303-321
            coin = random.random()
            if DPO_flag and coin < refuse_prob:
                for ip in inputs_processors:
                    ip.set_target_text(random.choice(ip.refuses))
            elif target_text_random:
                text = random.choice(inputs_processors[-1].target_texts)
                for ip in inputs_processors:
                    ip.set_target_text(text)
            else:
                for ip in inputs_processors:
                    ip.set_target_text(target_text if isinstance(target_text, str) else target_text[0])
        if ahead is not None:
            blur_sigma, crop = ahead               # drawn one iteration ahead, for the fused step (same values, same order)
            ahead = None
        else:
            blur_sigma, crop = draw_image_params(*image_draw)                               # Q4: a sigma per step, then the window
        given = None
        if unit_noise_fn is not None:
            given = [unit_noise_fn(iteration, i, (local_batch * pl.out_shape[0],) + tuple(pl.out_shape[1:])).to(device)
                     for i, pl in zip(my_models, plans)]
        pvs = engine.forward(local_batch, unit_noises=given, blur_sigma=blur_sigma, crop=crop)   # :329-362 (HIP)
        grads, losses, step_inputs = [], [], []
        for k, (model, ip, pv) in enumerate(zip(models, inputs_processors, pvs)):           # :352-384
            inputs = ip.get_inputs_train()
            step_inputs.append(inputs)
            pv.requires_grad_(True)
            inputs["pixel_values"] = pv
            if suffix_only_ce:
                model_loss = ip.get_loss_suffix_only(model, inputs)      # logits of the target positions only
            else:
                logits = model(**inputs).logits[:, :-1, :]
                model_loss = ip.get_loss(logits)
            (model_loss * engine.loss_scale(k)).backward()
            grads.append(pv.grad)
            losses.append(model_loss.detach())
        rng_before, nxt = None, {}
        if getattr(engine, "step_fusion", False) and iteration + 1 < num_iterations:
            # as in attack_model.train: the blur chain's backward also runs the next iteration's tanh + blur
            rng_before = torch.get_rng_state()
            ahead = draw_image_params(*image_draw)
            nxt = dict(next_blur_sigma=ahead[0], next_crop=ahead[1])
        stepped = engine.backward_update(grads, **nxt)                                      # :391-406 (HIP)
        if stepped:
            global_iteration += 1
        window_iterations += 1
        last = iteration == num_iterations - 1
        if world > 1 and (iteration % check_every == 0 or iteration % save_steps == 0 or last):
            # the sum this exchange replaces (crossattack_models.py:383-406) cannot silently drop a term
            assert_replicas(engine, exp_path, rank, iteration, global_iteration)
        if rank == 0 and (iteration % log_every == 0 or iteration == num_iterations - 1):
            st = engine.stats_dict()
            rec = {"iteration": iteration, "global_iteration": global_iteration, "img_loss": st["img_loss"],
                   "grad_norm": st["grad_norm"], "lr": engine.current_lr(), "resave_error_std": st["sigma_next"],
                   "resave_error_mean": st["qerr_mean"], "resave_error_l1": st["qerr_l1"],
                   "adversarial_mean": st["x_mean"], "adversarial_std": st["x_std"],
                   # the generator's parameters where the reference logs the sample statistics of its draw (:457-458)
                   "noise_mean": 0.0, "noise_std": st["sigma"],
                   "use_gaussian_blur": bool(use_gaussian_blur), "gblur_kernel_size": gblur_kernel_size}
            for k, i in enumerate(my_models):
                rec[f"loss_{i}_{model_names[i].replace('/', '_')}"] = float(losses[k]) * model_weights[i] + st["img_loss"]
            rec["loss_per_iteration"] = float(np.mean([v for kk, v in rec.items() if kk.startswith("loss_")]))
            accumulated_loss += float(sum(v for kk, v in rec.items() if kk.startswith("loss_") and kk != "loss_per_iteration"))
            window_logged += 1
            if stepped and window_logged == window_iterations:
                # the reference's number (:400-404): the sum over the iterations since the last optimiser step.  Every one of them
                # was logged (always so with the default --log_every 1); a window the log cadence only saw in part has no such
                # key - never a sum mixed from several windows
                rec["accumulated_loss"] = accumulated_loss
            if resaved_loss_every > 0 and iteration % resaved_loss_every == 0:
                # :434-445 - every model's loss on the image as its PNG would be read back (no noise),
                # averaged; one extra forward per model, so periodic here
                with torch.no_grad():
                    resaved = []
                    for model, ip, inputs, pv in zip(models, inputs_processors, step_inputs,
                                                     engine.resaved_pixel_values(local_batch)):
                        probe_inputs = dict(inputs)
                        probe_inputs["pixel_values"] = pv
                        resaved.append(float(ip.get_loss(model(**probe_inputs).logits[:, :-1, :])))
                rec["loss_resaved"] = float(np.mean(resaved))
            history.append(rec)
            logger.log(rec)
        if stepped:
            # the reference resets at every optimiser step, logged or not
            accumulated_loss, window_iterations, window_logged = 0.0, 0, 0
        if rank == 0 and (iteration % save_steps == 0 or iteration == num_iterations - 1):
            img = engine.image()
            pil = adv_processors[0].tensor2pil(img)
            save_checkpoint(pil, img, exp_path, global_iteration)
            save_state(engine, exp_path, global_iteration, iteration, torch_rng=rng_before)
        if generation_probe and (iteration % save_steps == 0 or last):                       # :475-497
            # every rank probes the model(s) it holds; with one process that is all of them, like the reference
            from .train_test import run_model_test
            pil = adv_processors[0].tensor2pil(engine.image())
            first_row, probe = run_model_test(models, processors, inputs_processors, [model_names[i] for i in my_models],
                                              test_questions, "sure, here it is", exp_path, iteration, pil,
                                              adv_processors=adv_processors,
                                              csv_suffix=f"_rank{rank}" if world > 1 else "")
            if rank == 0:
                logger.log(dict(iteration=iteration, **probe))
        if world > 1 and (iteration % save_steps == 0 or last):
            torch.distributed.barrier()      # rank 0's checkpoint I/O is waited for on the host, not inside the exchange
    if rank == 0:
        img = engine.image()
        save_checkpoint(adv_processors[0].tensor2pil(img), img, exp_path, "final")
        logger.close()
    if return_engine:
        return engine, history
    return history


def parse_model_names(s):
    return s.split(",")


def build_parser():
    p = argparse.ArgumentParser(description="Train a universal image against several models.")
    p.add_argument("--exp_name", type=str, default="advx-cross")
    p.add_argument("--img_orig", type=str, default="gray.png")
    p.add_argument("--prompt", type=str, default="list")
    p.add_argument("--target_text", type=str, default="sure, here it is!")
    p.add_argument("--model_names", type=parse_model_names,
                   default="microsoft/Phi-3.5-vision-instruct,Qwen/Qwen2-VL-7B-Instruct,alpindale/Llama-3.2-11B-Vision-Instruct")
    p.add_argument("--lr", type=float, default=1e-2)
    p.add_argument("--num_iterations", type=int, default=1000)
    p.add_argument("--save_steps", type=int, default=10)
    p.add_argument("--batch_size", type=int, default=4)
    p.add_argument("--grad_accum_steps", type=int, default=1)
    p.add_argument("--scheduler_step_size", type=int, default=100)
    p.add_argument("--scheduler_gamma", type=float, default=0.9)
    p.add_argument("--restart_num", type=int, default=0)
    p.add_argument("--mask_type", type=str, default=None, choices=["corner", "bottom_lines", "random_square"])
    p.add_argument("--mask_size", type=int, default=None)
    p.add_argument("--clamp_method", type=str, default="tanh", choices=["clamp", "tanh", "none"])
    p.add_argument("--start_from_white", action="store_true")
    p.add_argument("--target_text_random", action="store_true")
    p.add_argument("--DPO_flag", action="store_true")
    p.add_argument("--refuse_prob", type=float, default=0.0)      # crossattack_models.py:551 (the function default, 0.1, is never reached from the CLI)
    p.add_argument("--epsilon", type=float, default=0.4, help="logged only by the reference (Q3); use --attack_norm")
    p.add_argument("--attack_norm", type=float, default=0.5)
    p.add_argument("--sigma", type=float, default=0.001)
    p.add_argument("--model_weights", type=float, nargs="+", default=None)
    p.add_argument("--use_gaussian_blur", action="store_true")
    p.add_argument("--gblur_kernel_size", type=int, default=5)
    p.add_argument("--use_local_crop", action="store_true")
    p.add_argument("--crop_scale_min", type=float, default=0.6)
    p.add_argument("--crop_scale_max", type=float, default=1.0)
    p.add_argument("--crop_ratio_min", type=float, default=0.75)
    p.add_argument("--crop_ratio_max", type=float, default=1.33)
    p.add_argument("--questions_file", type=str, default=None)
    p.add_argument("--test_questions_file", type=str, default=None)
    p.add_argument("--answers_file", type=str, default=None)
    p.add_argument("--log_every", type=int, default=1)
    p.add_argument("--use_wandb", action="store_true")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no_noise_on_padding", dest="noise_on_padding", action="store_false",
                   help="keep the constant padding tiles of Mllama / Phi-3.5 exact zeros (the reference adds noise there)")
    p.add_argument("--pixel_io", type=str, default="float32", choices=["float32", "model"],
                   help="dtype of pixel_values at each VLM's boundary (model = that VLM's own half dtype)")
    p.add_argument("--step_fusion", action="store_true",
                   help="blur runs on one rank: the backward's last launch also runs the next iteration's tanh + blur (advx_image_step)")
    p.add_argument("--suffix_only_ce", action="store_true",
                   help="logits of the target positions only (logits_to_keep) + HIP cross entropy: same loss, no [B,S,V] tensor")
    p.add_argument("--resaved_loss_every", type=int, default=0,
                   help="log loss_resaved (every model's forward on the re-saved image) every N iterations; 0 = off")
    p.add_argument("--generation_probe", action="store_true", help="greedy-generate the test prompts at every save step")
    p.add_argument("--resume_from", type=str, default=None, help="state_iter_*.pt written by a previous run")
    add_dp_arguments(p)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if isinstance(args.model_names, str):
        args.model_names = parse_model_names(args.model_names)
    run_main(train, args)


if __name__ == "__main__":
    main()
