#!/usr/bin/env python3
"""Where do the HIP path and the torch-CPU oracle disagree on `0` versus the smallest fp32 denormal (1.4e-45)?

VERDICT r02 item 3: three sign-optimiser fuzz cases (seed 777001: 153, 883, 2242) ended with sign(0) on one side
and sign(1.4e-45) on the other - a whole lr step at one pixel.  This tool (GPU box) prints the evidence:
  1. arithmetic: does either side flush?  denormal products / sums on the CPU (ATen), on the GPU under torch, and in
     this library's blur kernels (an impulse image through advx_blur_fwd / advx_blur_bwd);
  2. the blur's separable form against the oracle's 2-D product kernel on an impulse, at the denormal level;
  3. the reported case replayed (`--case`): for every step, the pixels where exactly one of the two gradients is zero,
     with their operands.

    python tools/diag_denormal.py [--seed 777001 --case 883]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

TINY = 1.1754944e-38         # smallest normal fp32


def arithmetic(dev):
    print("== 1. denormal arithmetic")
    a, b = torch.tensor([1e-30, 3e-23, 1.5e-45], dtype=torch.float32), torch.tensor([1e-10, 4e-23, 0.6], dtype=torch.float32)
    print("   CPU  (ATen)  products:", (a * b).tolist(), "| flush_denormal flag can be set:", torch.set_flush_denormal(False) is not None)
    print("   GPU  (torch) products:", (a.to(dev) * b.to(dev)).cpu().tolist())
    print("   CPU exp(-100), exp(-103):", torch.exp(torch.tensor([-100.0, -103.0])).tolist(),
          " GPU:", torch.exp(torch.tensor([-100.0, -103.0], device=dev)).cpu().tolist())
    # conv2d on the CPU with denormal weights (the oracle's blur is a depthwise conv2d)
    import torch.nn.functional as F
    w = torch.full((1, 1, 1, 1), 1e-40)
    print("   CPU conv2d(1.0, weight 1e-40):", float(F.conv2d(torch.ones(1, 1, 1, 1), w)),
          "| conv2d(1e-3, weight 1e-40):", float(F.conv2d(torch.full((1, 1, 1, 1), 1e-3), w)))


def impulse(dev):
    from adversarialvlm_amd import ops
    from oracle import pixel_ops as P
    print("== 2. impulse through the blur, separable (HIP) vs 2-D product kernel (oracle)")
    for k, sigma in ((15, 0.48), (9, 0.3), (5, 0.15), (15, 0.7)):
        x = torch.zeros(3, 33, 33)
        x[:, 16, 16] = 1.0
        ref = P.gaussian_blur(x, k, sigma)
        got = ops.blur_fwd(x.to(dev), k, sigma).cpu()
        g1 = P.gaussian_kernel1d(k, sigma)
        den_ref = int(((ref != 0) & (ref.abs() < TINY)).sum())
        den_got = int(((got != 0) & (got.abs() < TINY)).sum())
        only_ref = int(((ref != 0) & (got == 0)).sum())
        only_got = int(((got != 0) & (ref == 0)).sum())
        worst = float((got - ref).abs().max())
        print(f"   k={k} sigma={sigma}: 1-D weights min {float(g1.min()):.3e}; denormal outputs oracle {den_ref} / HIP {den_got}; "
              f"nonzero only in oracle {only_ref}, only in HIP {only_got}; max |diff| {worst:.3e}")
        if only_ref or only_got:
            idx = torch.nonzero(((ref != 0) & (got == 0)) | ((got != 0) & (ref == 0)))[:4]
            for c, y, xx in idx.tolist():
                i, j = y - 16 + k // 2, xx - 16 + k // 2
                wi, wj = float(g1[i]), float(g1[j])
                print(f"      ({c},{y},{xx}): oracle {float(ref[c, y, xx]):.3e}  HIP {float(got[c, y, xx]):.3e}   "
                      f"w_i {wi:.3e} w_j {wj:.3e}  fl(w_i*w_j) {float(g1[i] * g1[j]):.3e}  exact {wi * wj:.3e}")


def replay(dev, seed, case):
    import fuzz_pgd
    from oracle.pgd import PGDOracle
    from adversarialvlm_amd.pgd import PixelPGD
    rng = np.random.default_rng(seed)
    for k in range(case + 1):
        desc, procs, batches, steps, mask, kw = fuzz_pgd.draw_case(rng)
    x0 = torch.rand(3, desc["H"], desc["W"], generator=torch.Generator().manual_seed(seed * 7919 + case)) * 1.1 - 0.05
    print(f"== 3. seed {seed} case {case}: {desc}")
    plans = [p[2] for p in procs]
    ora = PGDOracle(x0, [p[1] for p in procs], lr=kw["lr"], mask=mask, grad_accum_steps=kw.get("accum", 1),
                    blur_kernel=kw.get("blur_kernel"), model_weights=kw.get("weights"), cross_mode=kw.get("cross", False),
                    optimizer=kw["optimizer"], scheduler_gamma=kw["gamma"], scheduler_step_size=kw["step_size"])
    eng = PixelPGD(x0.to(dev), plans, lr=kw["lr"], mask=None if mask is None else mask.to(dev), grad_accum_steps=kw.get("accum", 1),
                   blur_kernel=kw.get("blur_kernel"), model_weights=kw.get("weights"), cross_mode=kw.get("cross", False),
                   optimizer=kw["optimizer"], allow_fused=kw.get("fused", True), scheduler_gamma=kw["gamma"],
                   scheduler_step_size=kw["step_size"])
    gen = torch.Generator().manual_seed(11)
    shapes = [(B * pl.out_shape[0],) + pl.out_shape[1:] for pl, B in zip(plans, batches)]
    all_z = [[torch.randn(s, generator=gen) for s in shapes] for _ in range(steps + 1)]
    for t in range(steps):
        crop = kw["crop_fn"](t) if "crop_fn" in kw else None
        bs = kw["blur_sigma_fn"](t) if "blur_sigma_fn" in kw else None
        gs = [torch.randn(s, generator=gen) * 0.01 for s in shapes]
        ora.forward(batches, all_z[t], blur_sigma=bs, crop=crop)
        eng.forward(batches, [z.to(dev) for z in all_z[t]], blur_sigma=bs, crop=crop)
        ref = ora.backward_update(gs)
        eng.backward_update([g.to(dev) * eng.loss_scale(i) for i, g in enumerate(gs)])
        ge, gr = eng.grad.cpu(), ref["grad"]
        one_zero = ((ge == 0) != (gr == 0))
        den_e = int(((ge != 0) & (ge.abs() < TINY)).sum())
        den_r = int(((gr != 0) & (gr.abs() < TINY)).sum())
        print(f"   step {t}: blur sigma {bs}, crop {crop}: max|g| {float(gr.abs().max()):.3e}; denormal entries HIP {den_e} / oracle {den_r}; "
              f"pixels where exactly one side is 0: {int(one_zero.sum())}; p differs by > lr/2 at "
              f"{int(((eng.p.cpu() - ora.p.detach()).abs() > 0.5 * kw['lr']).sum())} pixels")
        for c, y, x in torch.nonzero(one_zero)[:6].tolist():
            print(f"      ({c},{y},{x}): HIP {float(ge[c, y, x]):.3e}  oracle {float(gr[c, y, x]):.3e}")
        with torch.no_grad():            # keep the two runs together, as the trajectory test does
            ora.p.copy_(eng.p.cpu())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=777001)
    ap.add_argument("--case", type=int, default=883)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    arithmetic(dev)
    impulse(dev)
    replay(dev, a.seed, a.case)


if __name__ == "__main__":
    main()
