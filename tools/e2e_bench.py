#!/usr/bin/env python3
"""End-to-end timing (B) of SURVEY.md 8(d): the same PGD loop as bench.py, but with the VLM in it.

A random-init LLaVA-1.5-7B ARCHITECTURE (`LlavaConfig()` defaults = CLIP-L/14-336 + Llama-7B,
fp16, frozen) runs forward + backward-to-pixel_values under PyTorch-ROCm for a 64-prompt batch
(576 image tokens + prompt + target per row), split into micro-batches so the activations fit;
the pixel path on either side is the fused HIP pair.  This number is bounded by ~1e15 FLOP of
dense GEMM per step that this repository does not own; it is reported NEXT TO bench.py's
pixel-path number, never instead of it.

    python tools/e2e_bench.py [--steps 3] [--micro 8] [--model synthetic/llava-1.5-7b]

`--model synthetic/mllama-11b` / `synthetic/qwen2-vl-7b` run the same loop around random-init models of the Llama-3.2-11B-Vision
and Qwen2-VL-7B architectures (BASELINE configs[2] / [3]: the per-rank workload at the real model scale): the prepared chain feeds
`pixel_values [B,1,4,3,560,560]` + aspect-ratio / cross-attention tensors, resp. `[B*n_patches, 1176]` + `image_grid_thw`.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--micro", type=int, default=8)
    ap.add_argument("--model", type=str, default="synthetic/llava-1.5-7b")
    ap.add_argument("--image", type=int, default=336, help="side of the square image being optimised")
    ap.add_argument("--suffix-only-ce", action="store_true", help="logits of the target positions only + HIP cross entropy")
    ap.add_argument("--pixel-io", default="float32", choices=["float32", "model"])
    args = ap.parse_args()
    import threading

    import adversarialvlm_amd.testing  # noqa: F401  (registers the random-init `synthetic/*` architectures)
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.processors import load_components
    # MIOpen searches its convolution kernels the first time it sees a shape: the first step of a vision tower can take minutes
    # (end to end).  A line a minute tells whoever watches that this is not a hang.
    born = time.perf_counter()

    def heartbeat():
        while True:
            time.sleep(60)
            print(f"[e2e] alive, {time.perf_counter() - born:.0f} s", file=sys.stderr, flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()
    dev = torch.device("cuda:0")
    load, AdvInputs, DiffProc = load_components(args.model)
    t0 = time.perf_counter()
    model, proc = load(args.model, dev)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    print(f"[e2e] {args.model} loaded in {t_load:.1f} s, {torch.cuda.memory_allocated() / 2 ** 30:.1f} GiB", file=sys.stderr, flush=True)
    import numpy as np
    from PIL import Image
    size = args.image
    ap_ = DiffProc(proc.image_processor, dev)
    x0 = torch.rand(3, size, size, generator=torch.Generator().manual_seed(0)).to(dev)
    plan = ap_.plan_for(size, size)
    eng = PixelPGD(x0, [plan], seed=1, io_dtype=model.dtype if args.pixel_io == "model" else torch.float32)
    qs = [" ".join(f"w{random.Random(i).randint(0, 9999)}" for _ in range(30)) for i in range(97)]
    image = Image.fromarray((x0.permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8))
    ip = AdvInputs(questions=qs, test_questions=["t"], batch_size=args.batch, original_image=image, processor=proc,
                   device=dev, target_text="a b c d e f g h", rng=random.Random(0))
    if hasattr(ip, "bind_geometry"):
        ip.bind_geometry(ap_, size, size)
    B, mb = args.batch, args.micro
    assert B % mb == 0
    probe = ip.get_inputs_train()["input_ids"]
    rows = model.get_input_embeddings().weight.shape[0]
    assert int(probe.min()) >= 0 and int(probe.max()) < rows, "token ids outside the embedding table"
    lead = plan.out_shape[0]              # rows of pixel_values per sample (Qwen2-VL: patches; else 1)

    pv_shape = (B * lead,) + tuple(plan.out_shape[1:])

    def step():
        inputs = ip.get_inputs_train()
        pv = eng.forward(B)[0]
        grad = torch.empty_like(pv)
        loss_total = 0.0
        tgt_full = ip.target
        for i in range(0, B, mb):
            chunk = pv[i * lead:(i + mb) * lead].detach().requires_grad_(True)
            # every side tensor of the batch has one row per sample
            mb_inputs = {k: v[i:i + mb] for k, v in inputs.items() if k != "pixel_values"}
            mb_inputs["pixel_values"] = chunk.to(model.dtype)
            ip.target = tgt_full[i:i + mb]
            if args.suffix_only_ce:
                loss = ip.get_loss_suffix_only(model, mb_inputs)
            else:
                loss = ip.get_loss(model(**mb_inputs).logits[:, :-1, :].float())
            ip.target = tgt_full
            if os.environ.get("E2E_VERBOSE"):
                torch.cuda.synchronize()
                print(f"[e2e]   micro-batch {i // mb}: forward done, loss {float(loss):.4f}", file=sys.stderr, flush=True)
            (loss * (mb / B) * eng.loss_scale(0)).backward()        # mean over the whole batch
            if os.environ.get("E2E_VERBOSE"):
                torch.cuda.synchronize()
                print(f"[e2e]   micro-batch {i // mb}: backward done", file=sys.stderr, flush=True)
            grad[i * lead:(i + mb) * lead] = chunk.grad
            loss_total += float(loss.detach()) * mb / B
        eng.backward_update([grad])
        return loss_total

    first_step_s = None
    for k in range(args.warmup):
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        if first_step_s is None:
            first_step_s = time.perf_counter() - t1
        print(f"[e2e] warm-up step {k}: {time.perf_counter() - t1:.2f} s, peak {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB",
              file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for k in range(args.steps):
        losses.append(step())
        print(f"[e2e] step {k} issued, {time.perf_counter() - t0:.2f} s so far", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    S = int(ip.get_inputs_train()["input_ids"].shape[1])
    n_params = sum(p.numel() for p in model.parameters())
    flops = 4.0 * n_params * B * S          # fwd 2*P*T + bwd-to-input 2*P*T (weights frozen), attention excluded
    print(json.dumps({"e2e_steps_per_s": round(1.0 / dt, 4), "e2e_prompt_steps_per_s": round(B / dt, 2),
                      "s_per_step": round(dt, 3), "batch": B, "micro_batch": mb, "seq_len": S, "params": n_params,
                      "model": args.model, "dtype": str(model.dtype), "suffix_only_ce": bool(args.suffix_only_ce),
                      "pixel_io": args.pixel_io, "chain": eng.mode, "image": size,
                      "pixel_values_shape": list(pv_shape), "approx_model_tflops": round(flops / dt / 1e12, 1),
                      "losses": [round(v, 4) for v in losses], "load_s": round(t_load, 1),
                      "first_step_s": None if first_step_s is None else round(first_step_s, 2),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == "__main__":
    main()
