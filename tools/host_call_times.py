#!/usr/bin/env python3
"""Host time of PixelPGD.forward / backward_update per call (development tool): perf_counter around each call, no synchronisation
inside the loop, median over the steps after warm-up.  At the reference's own batch sizes (1-4 prompts) the plan-based chains run at
the HOST's pace; this shows which call the host spends its time in.
    ADVX_CHAIN_BATCH=1 python tools/host_call_times.py llava-crop generic"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adversarialvlm_amd import ops  # noqa: E402
from adversarialvlm_amd.pgd import PixelPGD  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "llava-crop"
    B = int(os.environ.get("ADVX_CHAIN_BATCH", "1"))
    dev = torch.device("cuda:0")
    H = W = 512
    crop = (40, 30, 400, 420) if "crop" in which else None
    kw = dict(use_crop=crop is not None, **(dict(blur_kernel=9) if "blur" in which else {}))
    plan = Plan.llava(H, W)
    eng = PixelPGD(torch.rand(3, H, W, device=dev), [plan], allow_fused=False, **kw)
    g = [torch.randn(B, plan.out_numel, device=dev)]
    sig = 7.0 if "blur" in which else None
    tf, tb = [], []
    for it in range(400):
        if it % 100 == 0:
            torch.cuda.synchronize()          # keep the launch queue from filling: host time only
        t0 = time.perf_counter()
        eng.forward(B, blur_sigma=sig, crop=crop)
        t1 = time.perf_counter()
        eng.backward_update(g)
        t2 = time.perf_counter()
        if it >= 50:
            tf.append(t1 - t0)
            tb.append(t2 - t1)
    torch.cuda.synchronize()
    print(f"{which} B={B}: forward {statistics.median(tf) * 1e6:.1f} us, backward_update {statistics.median(tb) * 1e6:.1f} us per call (host, median)")
    if crop is not None:
        t0 = time.perf_counter()
        for _ in range(2000):
            ops.crop_composes(plan, H, W, crop)
        t1 = time.perf_counter()
        for _ in range(2000):
            ops.crop_compose_rows(plan, H, W, crop)
        t2 = time.perf_counter()
        print(f"   ops.crop_composes {(t1 - t0) / 2000 * 1e6:.2f} us, ops.crop_compose_rows (compose_exact's walk) {(t2 - t1) / 2000 * 1e6:.2f} us per call")


if __name__ == "__main__":
    main()
