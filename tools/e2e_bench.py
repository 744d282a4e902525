#!/usr/bin/env python3
"""End-to-end timing (B) of SURVEY.md 8(d): the same PGD loop as bench.py, but with the VLM in it.

A random-init LLaVA-1.5-7B ARCHITECTURE (`LlavaConfig()` defaults = CLIP-L/14-336 + Llama-7B,
fp16, frozen) runs forward + backward-to-pixel_values under PyTorch-ROCm for a 64-prompt batch
(576 image tokens + prompt + target per row), split into micro-batches so the activations fit;
the pixel path on either side is the fused HIP pair.  This number is bounded by ~1e15 FLOP of
dense GEMM per step that this repository does not own; it is reported NEXT TO bench.py's
pixel-path number, never instead of it.

    python tools/e2e_bench.py [--steps 3] [--micro 8] [--model synthetic/llava-1.5-7b]
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
    ap.add_argument("--suffix-only-ce", action="store_true", help="logits of the target positions only + HIP cross entropy")
    ap.add_argument("--pixel-io", default="float32", choices=["float32", "model"])
    args = ap.parse_args()
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.processors import load_components
    dev = torch.device("cuda:0")
    load, AdvInputs, DiffProc = load_components(args.model)
    t0 = time.perf_counter()
    model, proc = load(args.model, dev)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    size = proc.image_processor.crop_size["height"]
    ap_ = DiffProc(proc.image_processor, dev)
    x0 = torch.rand(3, size, size, generator=torch.Generator().manual_seed(0)).to(dev)
    eng = PixelPGD(x0, [ap_.plan_for(size, size)], seed=1,
                   io_dtype=model.dtype if args.pixel_io == "model" else torch.float32)
    qs = [" ".join(f"w{random.Random(i).randint(0, 9999)}" for _ in range(30)) for i in range(97)]
    ip = AdvInputs(questions=qs, test_questions=["t"], batch_size=args.batch, original_image=None, processor=proc,
                   device=dev, target_text="a b c d e f g h", rng=random.Random(0))
    B, mb = args.batch, args.micro
    assert B % mb == 0
    probe = ip.get_inputs_train()["input_ids"]
    rows = model.get_input_embeddings().weight.shape[0]
    assert int(probe.min()) >= 0 and int(probe.max()) < rows, "token ids outside the embedding table"

    def step():
        inputs = ip.get_inputs_train()
        pv = eng.forward(B)[0]
        grad = torch.empty_like(pv)
        loss_total = 0.0
        tgt_full = ip.target
        for i in range(0, B, mb):
            chunk = pv[i:i + mb].detach().requires_grad_(True)
            mb_inputs = dict(input_ids=inputs["input_ids"][i:i + mb], attention_mask=inputs["attention_mask"][i:i + mb],
                             pixel_values=chunk.to(model.dtype))
            ip.target = tgt_full[i:i + mb]
            if args.suffix_only_ce:
                loss = ip.get_loss_suffix_only(model, mb_inputs)
            else:
                loss = ip.get_loss(model(**mb_inputs).logits[:, :-1, :].float())
            ip.target = tgt_full
            (loss * (mb / B) * eng.loss_scale(0)).backward()        # mean over the whole batch
            grad[i:i + mb] = chunk.grad
            loss_total += float(loss.detach()) * mb / B
        eng.backward_update([grad])
        return loss_total

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    S = int(ip.get_inputs_train()["input_ids"].shape[1])
    n_params = sum(p.numel() for p in model.parameters())
    flops = 4.0 * n_params * B * S          # fwd 2*P*T + bwd-to-input 2*P*T (weights frozen), attention excluded
    print(json.dumps({"e2e_steps_per_s": round(1.0 / dt, 4), "e2e_prompt_steps_per_s": round(B / dt, 2),
                      "s_per_step": round(dt, 3), "batch": B, "micro_batch": mb, "seq_len": S, "params": n_params,
                      "model": args.model, "dtype": str(model.dtype), "suffix_only_ce": bool(args.suffix_only_ce),
                      "pixel_io": args.pixel_io, "approx_model_tflops": round(flops / dt / 1e12, 1),
                      "losses": [round(v, 4) for v in losses], "load_s": round(t_load, 1),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == "__main__":
    main()
