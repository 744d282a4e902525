#!/usr/bin/env python3
"""Single-model crop chains (the reference's --use_local_crop presets: attack_clamp_tanh_{llava,llama,phi3,qwen2vl*}.sh) at full
size, B = 64: the composed window o plan tables (default) against the two-launch form (ADVX_TUNE_SEPARATE_CROP).  Development tool.

    python tools/crop_chain_bench.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adversarialvlm_amd import ops  # noqa: E402
from adversarialvlm_amd.pgd import PixelPGD  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def run(mk_plan, H, W, B, blur, crop, steps=300):
    dev = torch.device("cuda:0")
    plan = mk_plan()
    eng = PixelPGD(torch.rand(3, H, W, device=dev), [plan], blur_kernel=blur, use_crop=True, allow_fused=False)
    g = torch.randn(B, plan.out_numel, device=dev)

    def step():
        eng.forward(B, blur_sigma=7.0 if blur else None, crop=crop)
        eng.backward_update([g])
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6, ops.crop_composes(plan, H, W, crop)


def main():
    cases = [("llava 512->336", lambda: Plan.llava(512, 512), 512, (40, 30, 400, 420)),
             ("mllama 512 (4 x 560 tiles)", lambda: Plan.mllama(512, 512), 512, (40, 30, 400, 420)),
             ("qwen2vl 512", lambda: Plan.qwen2vl(512, 512), 512, (40, 30, 400, 420)),
             ("phi3 512", lambda: Plan.phi3(512, 512), 512, (40, 30, 400, 420)),
             ("llava 336 (identity resize)", lambda: Plan.llava(336, 336), 336, (26, 20, 262, 276)),
             ("mllama 336 (one 560 tile of four)", lambda: Plan.mllama(336, 336), 336, (26, 20, 262, 276))]
    print(f"{'chain (B = 64, crop window 0.78 x 0.82 of the image)':44s} {'blur':>5s} | {'composed':>10s} | {'two launches':>12s}")
    for name, mk, S, win in cases:
        for blur in (None, 9):
            res = []
            for rep in range(2):                       # alternate the two forms: box noise shows as disagreement between the repeats
                with ops.compose_crop_everywhere():
                    a, composes = run(mk, S, S, 64, blur, win)
                with ops.separate_crop():
                    b, _ = run(mk, S, S, 64, blur, win)
                res.append((a, b))
            print(f"{name:44s} {str(blur):>5s} | {res[0][0]:6.1f} / {res[1][0]:6.1f} us | {res[0][1]:6.1f} / {res[1][1]:6.1f} us   "
                  f"{'' if composes else '(does not compose)'}", flush=True)


if __name__ == "__main__":
    main()
