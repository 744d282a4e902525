#!/usr/bin/env python3
"""What a wider tap window costs the image-sized gather kernels (VERDICT r02 item 4 i: compose the crop window's
resize with the plan's stage-0 resize into one table per axis, one launch instead of two).

A separable resize evaluated per output pixel costs rows x columns loads; composing two resizes ADDS their tap counts per
axis (window 2-3 taps + plan 5 taps -> 6-8), so the gather grows with the SQUARE: 25 + 9 loads in two launches become
49-64 in one.  This tool times the plan's forward resize (canvas 336 x 336 fixed, three channels per thread,
k_stage0_fwd_multi / k_stage_fwd_t) and its transposed resize for LLaVA plans whose source sizes give 3, 5, 7 and 9 forward taps per
axis, and the transposed resize of up-sampling plans (8-10 transposed taps per axis, what a composed backward would gather).
Batch 1, no noise: the emit / reduction beside the resize is a constant few microseconds.

    python tools/taps_cost.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adversarialvlm_amd import ops  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    print(f"{'source -> 336':>16} {'fwd taps/axis':>14} {'T taps/axis':>12} | {'emit B=1 (resize + emit)':>26} | {'collect B=1 (reduce + resize^T)':>32}")
    base = {}
    for src in (336, 404, 512, 672, 680, 1008, 1020):
        plan = Plan.llava(src, src, 336, 336)
        plan.upload()
        _, _, w = plan.taps(0, 0)
        _, tc, _ = plan.taps(0, 0, transposed=True)
        img = torch.rand(3, src, src, device=dev)
        g = torch.randn(1, plan.out_numel, device=dev)
        ws = torch.empty(plan.workspace_floats, device=dev)
        garg = torch.empty_like(img)
        te = timeit(lambda: ops.emit(plan, img, 1, workspace=ws))
        tc_us = timeit(lambda: ops.collect(plan, g, 1, grad_argument=garg, workspace=ws))
        base[src] = (te, tc_us)
        print(f"{src:>9} -> 336 {w.shape[1]:>14} {int(tc.max()):>12} | {te:>23.2f} us | {tc_us:>29.2f} us")
    print("up-sampling plans (the transposed resize of a 512 x 512 SOURCE gathers T x T canvas elements per pixel):")
    for out in (512, 1024, 1536, 2048):
        plan = Plan.llava(512, 512, out, out)
        plan.upload()
        _, tc, _ = plan.taps(0, 0, transposed=True)
        g = torch.randn(1, plan.out_numel, device=dev)
        ws = torch.empty(plan.workspace_floats, device=dev)
        garg = torch.empty(3, 512, 512, device=dev)
        t = timeit(lambda: ops.collect(plan, g, 1, grad_argument=garg, workspace=ws))
        print(f"   512 -> {out:>5}: transposed taps/axis {int(tc.max()):>3}   collect B=1 {t:8.2f} us")


if __name__ == "__main__":
    main()
