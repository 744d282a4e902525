#!/usr/bin/env python3
"""Step time of the generic kernel chain for the non-identity configurations (development tool)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adversarialvlm_amd.pgd import PixelPGD  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def run(name, plans, H, W, B, blur=None, crop=None, steps=50, cross=False, pad_noise=True, prepared=False, io=torch.float32,
        announce=True):
    dev = torch.device("cuda:0")
    x0 = torch.rand(3, H, W, device=dev)
    eng = PixelPGD(x0, plans, blur_kernel=blur, use_crop=crop is not None, cross_mode=cross, allow_fused=prepared,
                   fused_mode="prepared" if prepared else "auto", noise_on_padding=pad_noise, io_dtype=io, step_fusion=announce)
    name = f"{name} [{eng.mode}{'' if io == torch.float32 else ', ' + str(io).split('.')[-1]}]"
    gs = [torch.randn(B, pl.out_numel, device=dev).to(io) for pl in plans]

    # announce: tell backward_update the next step's blur sigma / window (the trainers do): a blur chain on one rank then runs
    # the next step's image kernel inside the backward's last launch (advx_image_step)
    nxt = dict(next_blur_sigma=7.0, next_crop=crop) if (blur and announce) else {}
    if blur:
        name += " [step fusion]" if (nxt and getattr(eng, "step_fusion", False)) else " [no step fusion]"

    def step():
        eng.forward(B, blur_sigma=7.0 if blur else None, crop=crop)
        eng.backward_update(gs, **nxt)
    for _ in range(5):
        step()
    # best of three timed blocks: a HIP process stalls once for ~40 ms on the host around its ~1600th launch (seen at the same
    # step in every run, whatever is launched; round 4) - inside a 50-step block that reads as 1.1 ms per step
    dt = float("inf")
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = min(dt, (time.perf_counter() - t0) / steps)
    # Two byte counts.  "tensor": what the reference's tensor costs - B*P_out written (padding tiles included: the
    # reference adds noise to them and Llama-3.2's vision encoder does see them, tests/test_mllama_padding_visibility.py)
    # plus the gradient of the elements an image reaches.  "algorithmic" (SURVEY 8(d)): the live elements only, both ways -
    # the figure the roofline fraction is quoted against; for Mllama it counts one tile of four (495 MB at B = 64).
    live = [pl.live_range()[1] - pl.live_range()[0] for pl in plans]
    wr = sum(pl.out_numel if pad_noise else lv for pl, lv in zip(plans, live))
    eb = 4 if io == torch.float32 else 2
    tensor_bytes = eb * B * (wr + sum(live)) + 4 * 10 * 3 * H * W
    algo_bytes = eb * B * 2 * sum(live) + 4 * 10 * 3 * H * W
    print(f"{name:52s} B={B:3d}  {dt * 1e6:9.1f} us/step  {1 / dt:9.1f} steps/s  algorithmic {algo_bytes / 1e6:8.1f} MB = "
          f"{algo_bytes / dt / 8e12:4.2f} of HBM peak; tensor {tensor_bytes / 1e6:8.1f} MB = {tensor_bytes / dt / 8e12:4.2f}", flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _tune import apply_env_tuning
    apply_env_tuning()                                      # ADVX_TUNE="code=value,..." (tools/_tune.py)
    if os.environ.get("ADVX_BWD_XCD") is not None:          # A/B of the readers' XCD-aware block map (ADVX_TUNE_BWD_XCD)
        from adversarialvlm_amd import _lib
        _lib.check(_lib.load().advx_set_tuning(7, int(os.environ["ADVX_BWD_XCD"])), "advx_set_tuning")
    if os.environ.get("ADVX_XCD_MAP") is not None:          # ... and of the writers' (ADVX_TUNE_XCD_MAP)
        from adversarialvlm_amd import _lib
        _lib.check(_lib.load().advx_set_tuning(6, int(os.environ["ADVX_XCD_MAP"])), "advx_set_tuning")
    run("llava 336 identity (generic)", [Plan.llava(336, 336)], 336, 336, 64)
    run("llava 512->336", [Plan.llava(512, 512)], 512, 512, 64)
    run("llava 512->336", [Plan.llava(512, 512)], 512, 512, 64, prepared=True)
    run("llava 512->336 blur9 crop", [Plan.llava(512, 512)], 512, 512, 64, blur=9, crop=(20, 30, 400, 420), announce=False)
    run("llava 512->336 blur9 crop", [Plan.llava(512, 512)], 512, 512, 64, blur=9, crop=(20, 30, 400, 420))
    run("llava 512->336 blur9", [Plan.llava(512, 512)], 512, 512, 64, blur=9, announce=False)
    run("llava 512->336 blur9", [Plan.llava(512, 512)], 512, 512, 64, blur=9)
    run("llava 336 blur5 crop", [Plan.llava(336, 336)], 336, 336, 64, blur=5, crop=(20, 30, 280, 300), announce=False)
    run("llava 336 blur5 crop", [Plan.llava(336, 336)], 336, 336, 64, blur=5, crop=(20, 30, 280, 300))
    run("mllama 336 (4x560 tiles)", [Plan.mllama(336, 336)], 336, 336, 64)
    run("mllama 336, padding kept zero", [Plan.mllama(336, 336)], 336, 336, 64, pad_noise=False)
    run("mllama 336 (4x560 tiles)", [Plan.mllama(336, 336)], 336, 336, 64, prepared=True)
    run("mllama 336, padding kept zero", [Plan.mllama(336, 336)], 336, 336, 64, pad_noise=False, prepared=True)
    run("mllama 336 B=32", [Plan.mllama(336, 336)], 336, 336, 32)
    run("phi3 512", [Plan.phi3(512, 512)], 512, 512, 64)
    run("phi3 512, padding kept zero", [Plan.phi3(512, 512)], 512, 512, 64, pad_noise=False)
    run("phi3 512", [Plan.phi3(512, 512)], 512, 512, 64, prepared=True)
    run("phi3 512, padding kept zero", [Plan.phi3(512, 512)], 512, 512, 64, pad_noise=False, prepared=True)
    run("qwen2vl 512", [Plan.qwen2vl(512, 512)], 512, 512, 64)
    run("qwen2vl 512", [Plan.qwen2vl(512, 512)], 512, 512, 64, prepared=True)
    run("mllama 336 (4x560 tiles)", [Plan.mllama(336, 336)], 336, 336, 64, prepared=True, io=torch.float16)
    run("mllama 336, padding kept zero", [Plan.mllama(336, 336)], 336, 336, 64, pad_noise=False, prepared=True, io=torch.float16)
    run("phi3 512", [Plan.phi3(512, 512)], 512, 512, 64, prepared=True, io=torch.float16)
    run("qwen2vl 512", [Plan.qwen2vl(512, 512)], 512, 512, 64, prepared=True, io=torch.bfloat16)
    run("llava 512->336", [Plan.llava(512, 512)], 512, 512, 64, prepared=True, io=torch.float16)
    run("cross phi3+qwen+mllama 336 blur5", [Plan.phi3(336, 336), Plan.qwen2vl(336, 336), Plan.mllama(336, 336)], 336, 336,
        16, blur=5, cross=True, announce=False)
    run("cross phi3+qwen+mllama 336 blur5", [Plan.phi3(336, 336), Plan.qwen2vl(336, 336), Plan.mllama(336, 336)], 336, 336,
        16, blur=5, cross=True)
    run("cross, padding kept zero", [Plan.phi3(336, 336), Plan.qwen2vl(336, 336), Plan.mllama(336, 336)], 336, 336,
        16, blur=5, cross=True, pad_noise=False)
