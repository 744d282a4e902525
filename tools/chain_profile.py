#!/usr/bin/env python3
"""Run one configuration of the plan-based chains for a profiler (development tool):
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o name -- python3 tools/chain_profile.py mllama prepared
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adversarialvlm_amd.pgd import PixelPGD  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "mllama"
    chain = sys.argv[2] if len(sys.argv) > 2 else "prepared"
    dev = torch.device("cuda:0")
    H = W = 336 if which == "mllama" else 512
    plans = {"mllama": lambda: [Plan.mllama(H, W)], "phi3": lambda: [Plan.phi3(H, W)], "qwen2vl": lambda: [Plan.qwen2vl(H, W)],
             "llava": lambda: [Plan.llava(H, W)]}[which]()
    B = 64
    x0 = torch.rand(3, H, W, device=dev)
    eng = PixelPGD(x0, plans, allow_fused=(chain == "prepared"))
    assert eng.mode == chain, eng.mode
    gs = [torch.randn(B, pl.out_numel, device=dev) for pl in plans]
    for _ in range(60):
        eng.forward(B)
        eng.backward_update(gs)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
