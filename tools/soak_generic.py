#!/usr/bin/env python3
"""Soak of the plan-based generic chain (development tool): N steps with a new random crop window every step (and, optionally, blur
with a new sigma), run TWICE from the same seeds - the two runs must leave identical bits (p, m, v, statistics), everything finite.
Exercises what a long training run exercises and single steps do not: the per-step table builds into one scratch, the records that
decide whether a backward may reuse them, compose_exact's memo, the one-launch collect + update.
    python tools/soak_generic.py [--steps 3000] [--blur 9] [--size 512] [--batch 2]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from adversarialvlm_amd.pgd import PixelPGD  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def run(steps, size, batch, blur, maker, seed):
    dev = torch.device("cuda:0")
    H = W = size
    plan = {"llava": Plan.llava, "qwen2vl": Plan.qwen2vl, "mllama": Plan.mllama, "phi3": Plan.phi3}[maker](H, W)
    gen = torch.Generator().manual_seed(seed)
    x0 = torch.rand(3, H, W, generator=gen).to(dev)
    eng = PixelPGD(x0, [plan], lr=3e-3, seed=seed, allow_fused=False, use_crop=True, **(dict(blur_kernel=blur) if blur else {}))
    rng = np.random.default_rng(seed)
    gs = [(torch.randn(batch, plan.out_numel, generator=gen) * 0.02).to(dev) for _ in range(4)]
    for t in range(steps):
        # torchvision's RandomResizedCrop draws: area 8 % .. 100 %, log-uniform aspect 3/4 .. 4/3 (attack_model.py:191-202)
        area = H * W * rng.uniform(0.08, 1.0)
        ar = np.exp(rng.uniform(np.log(3 / 4), np.log(4 / 3)))
        w, h = int(round(np.sqrt(area * ar))), int(round(np.sqrt(area / ar)))
        w, h = min(max(w, 4), W), min(max(h, 4), H)
        crop = (int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1)), h, w)
        eng.forward(batch, blur_sigma=float(rng.uniform(0.1, 2.0)) if blur else None, crop=crop)
        eng.backward_update([gs[t % 4]])
    st = eng.stats_dict()
    torch.cuda.synchronize()
    return eng.p.clone(), eng.m.clone(), eng.v.clone(), st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--blur", type=int, default=0)
    ap.add_argument("--maker", default="llava")
    a = ap.parse_args()
    t0 = time.time()
    one = run(a.steps, a.size, a.batch, a.blur, a.maker, 11)
    two = run(a.steps, a.size, a.batch, a.blur, a.maker, 11)
    same = all(torch.equal(x, y) for x, y in zip(one[:3], two[:3])) and one[3] == two[3]
    finite = all(bool(torch.isfinite(x).all()) for x in one[:3]) and all(np.isfinite(v) for v in one[3].values())
    print(f"soak {a.maker} {a.size}x{a.size} batch {a.batch} blur {a.blur}: {a.steps} steps twice in {time.time() - t0:.0f} s; "
          f"identical bits: {same}; finite: {finite}; |p| max {float(one[0].abs().max()):.3f}; stats {one[3]}", flush=True)
    return 0 if (same and finite) else 1


if __name__ == "__main__":
    sys.exit(main())
