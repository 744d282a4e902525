#!/usr/bin/env python3
"""Same-run evidence for "rocprofv3 reports the long float32 emits slower than they run unprofiled" (VERDICT r02 item 6).

Mllama 336 prepared chain, B = 64 (the 963 MB emit).  Every forward() - one k_emit launch - is bracketed by its own pair
of HIP events on the launch stream; the script prints their mean.  Run it bare and under
`rocprofv3 --kernel-trace --stats -- python3 tools/emit_pair.py`: the profiled run then holds, for the SAME launches, the
event-bracketed time and rocprofv3's own k_emit duration (its kernel_stats.csv), and the bare run the unprofiled time.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adversarialvlm_amd.pgd import PixelPGD  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    H = W = 336
    B = 64
    plan = Plan.mllama(H, W)
    eng = PixelPGD(torch.rand(3, H, W, device=dev), [plan], fused_mode="prepared")
    g = torch.randn(B, plan.out_numel, device=dev)
    for _ in range(10):
        eng.forward(B)
        eng.backward_update([g])
    torch.cuda.synchronize()
    pairs = []
    t0 = time.perf_counter()
    for _ in range(50):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.forward(B)              # prepared chain: exactly one launch, k_emit
        b.record()
        pairs.append((a, b))
        eng.backward_update([g])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    ev = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
    print(f"mllama 336 prepared, B = 64: k_emit between its own HIP events: mean {sum(ev) / len(ev):.1f} us, median {ev[len(ev) // 2]:.1f} us "
          f"(50 launches); whole step {wall:.1f} us wall", flush=True)


if __name__ == "__main__":
    main()
