#!/usr/bin/env python3
"""Pair (two launches per step) against the one-launch step chain at SMALL prompt batches.

The reference's own presets run batch 1-4 on one GPU (scripts/attacks/attack_clamp_tanh_llava.sh:30-32), and an 8-way
strong-scaled 64-prompt step leaves 8 prompts per rank: there the pair's two kernels move 1-11 MB each and a step is
bounded by launches, not bytes.  LLaVA 336 x 336, in-kernel noise, AdamW, resident buffers (nothing to keep cold at
these sizes); wall time per step over N steps between two synchronisations = what the host can issue AND the device
can run; `device` = the same steps between two HIP events.

    python tools/small_batch_bench.py [--steps 2000]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adversarialvlm_amd.pgd import PixelPGD  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def run(mode, B, steps, dev, H=336, W=336):
    x0 = torch.rand(3, H, W, generator=torch.Generator().manual_seed(0)).to(dev)
    eng = PixelPGD(x0, [Plan.llava(H, W)], lr=1e-2, seed=3, fused_mode=mode)
    g = (torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1)) * 0.01).to(dev)

    def step():
        eng.forward(B)
        eng.backward_update([g])
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(steps):
        step()
    b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e6
    return wall, a.elapsed_time(b) / steps * 1e3, eng.p.double().sum().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    print(f"{'B':>3} | {'pair wall':>10} {'device':>8} | {'step wall':>10} {'device':>8} | step/pair (wall)")
    for B in (1, 2, 4, 8, 16, 32, 64):
        pw, pd, _ = run("pair", B, a.steps, dev)
        sw, sd, _ = run("step", B, a.steps, dev)
        rows.append(dict(batch=B, pair_wall_us=round(pw, 2), pair_device_us=round(pd, 2), step_wall_us=round(sw, 2),
                         step_device_us=round(sd, 2)))
        print(f"{B:>3} | {pw:>10.2f} {pd:>8.2f} | {sw:>10.2f} {sd:>8.2f} | {sw / pw:.3f}", flush=True)
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
