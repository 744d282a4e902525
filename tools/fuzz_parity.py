#!/usr/bin/env python3
"""Randomised differential test of advx_emit / advx_collect against the CPU oracle (development tool;
the seeded subset that runs in CI is tests/test_gpu_processors.py::test_random_geometries).

    python tools/fuzz_parity.py [--cases 200] [--seed 0] [--budget-s 240]

Draws image sizes (including extreme aspect ratios, sizes below one tile, odd sizes) and processor
parameters, runs forward + backward on the GPU through the C ABI and on the CPU through
oracle/processors.py, and reports every case whose pixel_values / image gradient differ by more than
the parity bar or whose integer metadata differs at all.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle.processors import LlavaOracle, MllamaOracle, Phi3Oracle, Qwen2VLOracle  # noqa: E402

FWD_TOL, BWD_TOL = 5e-6, 5e-5


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def draw_case(rng):
    kind = rng.choice(["llava", "mllama", "phi3", "qwen"])
    shape = rng.choice(["square", "wide", "tall", "tiny", "any"])
    if shape == "square":
        H = W = int(rng.integers(8, 700))
    elif shape == "wide":
        H, W = int(rng.integers(8, 120)), int(rng.integers(300, 1100))
    elif shape == "tall":
        H, W = int(rng.integers(300, 1100)), int(rng.integers(8, 120))
    elif shape == "tiny":
        H, W = int(rng.integers(4, 40)), int(rng.integers(4, 40))
    else:
        H, W = int(rng.integers(8, 900)), int(rng.integers(8, 900))
    if kind == "llava":
        args = dict(crop_h=int(rng.choice([16, 33, 64, 224, 336])), crop_w=int(rng.choice([16, 33, 64, 224, 336])))
    elif kind == "mllama":
        args = dict(tile=int(rng.choice([16, 28, 64, 224, 560])), max_tiles=int(rng.integers(1, 5)))
    elif kind == "phi3":
        args = dict(num_crops=int(rng.choice([1, 2, 4, 6, 9, 16])))
        H, W = max(H, 8), max(W, 8)
    else:
        lo = int(rng.choice([4, 16, 64])) * 28 * 28
        hi = lo * int(rng.choice([1, 2, 8, 40]))
        args = dict(min_pixels=lo, max_pixels=hi)
    return kind, H, W, args


def run_case(kind, H, W, args, dev, seed):
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(seed)
    img = torch.rand(3, H, W, generator=gen) * 1.2 - 0.1
    ora = {"llava": LlavaOracle, "mllama": MllamaOracle, "phi3": Phi3Oracle, "qwen": Qwen2VLOracle}[kind](**args)
    x = img.clone().requires_grad_(True)
    try:
        res = ora.process(x)
    except Exception as e:                       # the oracle (like the reference) rejects the geometry
        try:
            getattr(Plan, {"qwen": "qwen2vl"}.get(kind, kind))(H, W, **args)
        except Exception:
            return "both-reject", 0.0, 0.0
        return f"oracle rejects ({type(e).__name__}: {e}) but the plan is accepted", 0.0, 0.0
    ref = res["pixel_values"]
    up = torch.randn(ref.shape, generator=gen)
    ref.backward(up)
    plan = getattr(Plan, {"qwen": "qwen2vl"}.get(kind, kind))(H, W, **args)
    xg = img.to(dev).requires_grad_(True)
    pv = ops.ProcessFunction.apply(xg, plan)
    if tuple(pv.shape) != tuple(ref.shape):
        return f"shape {tuple(pv.shape)} vs {tuple(ref.shape)}", 0.0, 0.0
    pv.backward(up.to(dev).view(pv.shape))
    meta = []
    if kind in ("mllama",) and int(plan.info.num_tiles) != int(res["num_tiles"]):
        meta.append(f"num_tiles {plan.info.num_tiles} vs {res['num_tiles']}")
    if kind == "qwen" and int(plan.info.num_tiles) != int(res["num_tiles"][0]):
        meta.append(f"num_tiles {plan.info.num_tiles} vs {res['num_tiles']}")
    if kind == "phi3":
        if [[plan.info.image_h, plan.info.image_w]] != res["image_sizes"]:
            meta.append(f"image_sizes {[plan.info.image_h, plan.info.image_w]} vs {res['image_sizes']}")
        if [plan.info.num_img_tokens] != res["num_img_tokens"]:
            meta.append(f"num_img_tokens {plan.info.num_img_tokens} vs {res['num_img_tokens']}")
    ef = rel_err(pv.detach().cpu(), ref.detach())
    eb = rel_err(xg.grad.cpu(), x.grad) if float(x.grad.abs().max()) > 0 else float(xg.grad.abs().max())
    if meta:
        return "; ".join(meta), ef, eb
    if not (ef < FWD_TOL and eb < BWD_TOL):
        return "numeric", ef, eb
    return "ok", ef, eb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--budget-s", type=float, default=240.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    t0, bad, done, worst = time.time(), 0, 0, (0.0, 0.0)
    for k in range(a.cases):
        if time.time() - t0 > a.budget_s:
            break
        kind, H, W, args = draw_case(rng)
        try:
            verdict, ef, eb = run_case(kind, H, W, args, dev, a.seed * 100003 + k)
        except Exception as e:
            verdict, ef, eb = f"EXCEPTION {type(e).__name__}: {e}", 0.0, 0.0
        done += 1
        worst = (max(worst[0], ef), max(worst[1], eb))
        if verdict not in ("ok", "both-reject"):
            bad += 1
            print(f"FAIL case {k}: {kind} {H}x{W} {args}: {verdict} (fwd {ef:.2e}, bwd {eb:.2e})", flush=True)
        elif k % 20 == 0:
            print(f"case {k}: {kind} {H}x{W} {args}: {verdict} (fwd {ef:.2e}, bwd {eb:.2e})", flush=True)
    print(f"{done} cases, {bad} failures, worst fwd {worst[0]:.2e} bwd {worst[1]:.2e}, {time.time() - t0:.0f} s", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
