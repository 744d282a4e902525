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
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _tune import apply_env_tuning
    apply_env_tuning()                                      # ADVX_TUNE="code=value,..." (tools/_tune.py)
    which = sys.argv[1] if len(sys.argv) > 1 else "mllama"
    chain = sys.argv[2] if len(sys.argv) > 2 else "prepared"
    dev = torch.device("cuda:0")
    H = W = 336 if which == "mllama" else 512
    crop = None
    if which in ("llava-crop", "llava-blur-crop"):      # what half of the reference's attack scripts run (--use_local_crop)
        H = W = 512
        plans, B = [Plan.llava(H, W)], 64
        kw = dict(use_crop=True, **(dict(blur_kernel=9) if "blur" in which else {}))
        crop = (40, 30, 400, 420)
    elif which == "llava336-crop":                      # native resolution + window, no blur
        H = W = 336
        plans, B = [Plan.llava(H, W)], 64
        kw = dict(use_crop=True)
        crop = (20, 30, 280, 300)
    elif which == "llava336-blur5-crop":                # native resolution + blur 5 + window (tools/generic_bench.py's row)
        H = W = 336
        plans, B = [Plan.llava(H, W)], 64
        kw = dict(use_crop=True, blur_kernel=5)
        crop = (20, 30, 280, 300)
    elif which in ("mllama-crop", "qwen2vl-crop", "phi3-crop"):      # a window the plan's stage 0 does not compose with: two launches each way
        H = W = 336 if which == "mllama-crop" else 512
        mk = {"mllama-crop": Plan.mllama, "qwen2vl-crop": Plan.qwen2vl, "phi3-crop": Plan.phi3}[which]
        plans, B = [mk(H, W)], 64
        kw = dict(use_crop=True)
        crop = (20, 30, H - 56, H - 36)
    elif which == "cross-noblur":                # four of the reference's five cross-model scripts run without blur
        H = W = 336
        plans, B, kw = [Plan.phi3(H, W), Plan.qwen2vl(H, W), Plan.mllama(H, W)], 16, dict(cross_mode=True)
    elif which == "cross":                       # BASELINE configs 4/5 in the shape tools/generic_bench.py times
        H = W = 336
        plans, B, kw = [Plan.phi3(H, W), Plan.qwen2vl(H, W), Plan.mllama(H, W)], 16, dict(blur_kernel=5, cross_mode=True)
    else:
        plans = {"mllama": lambda: [Plan.mllama(H, W)], "phi3": lambda: [Plan.phi3(H, W)], "qwen2vl": lambda: [Plan.qwen2vl(H, W)],
                 "llava": lambda: [Plan.llava(H, W)]}[which]()
        B, kw = 64, {}
    if os.environ.get("ADVX_CHAIN_BATCH"):          # the reference's own presets run 1-4 prompts per step
        B = int(os.environ["ADVX_CHAIN_BATCH"])
    x0 = torch.rand(3, H, W, device=dev)
    eng = PixelPGD(x0, plans, allow_fused=(chain == "prepared"), step_fusion=not os.environ.get("ADVX_NO_ANNOUNCE"), **kw)
    assert eng.mode == chain, eng.mode
    gs = [torch.randn(B, pl.out_numel, device=dev) for pl in plans]
    sig = 7.0 if "blur_kernel" in kw else None
    import time
    for it in range(60):
        if it == 10:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        eng.forward(B, blur_sigma=sig, crop=crop)
        # the trainers announce the next step's blur sigma / window: a blur chain then takes advx_image_step (ADVX_NO_ANNOUNCE=1: rounds 1-3)
        eng.backward_update(gs, **(dict(next_blur_sigma=sig, next_crop=crop) if (sig and not os.environ.get("ADVX_NO_ANNOUNCE")) else {}))
    torch.cuda.synchronize()
    print(f"{which} {chain}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us/step (wall, 50 steps)")


if __name__ == "__main__":
    main()
