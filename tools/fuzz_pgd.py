#!/usr/bin/env python3
"""Randomised differential test of whole PGD trajectories: PixelPGD (HIP, through the C ABI) against
oracle/pgd.py on identical inputs (development tool; a seeded subset runs as
tests/test_gpu_pgd.py::test_random_trajectories).

    python tools/fuzz_pgd.py [--cases 100] [--seed 0] [--budget-s 240]

Draws the image size, one processor or a weighted cross-model set, prompt batches, blur, random-resized
crop windows, localized masks, gradient accumulation, optimiser, scheduler and the kernel chain, runs 3-5
steps on both sides with the same noise / blur sigma / crop windows, and applies the parity bar of the
trajectory tests (p, grad, sigma, statistics within 1e-4 relative).
"""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import pixel_ops as P  # noqa: E402
from oracle.processors import LlavaOracle, MllamaOracle, Phi3Oracle, Qwen2VLOracle  # noqa: E402


def one_processor(rng, H, W, allow_phi3=True):
    from adversarialvlm_amd.plan import Plan
    kind = rng.choice(["llava", "llava-ident", "mllama", "qwen"] + (["phi3"] if allow_phi3 else []))
    if kind == "llava":
        ch, cw = int(rng.choice([16, 33, 48, 64])), int(rng.choice([16, 33, 48, 64]))
        return kind, LlavaOracle(ch, cw), Plan.llava(H, W, ch, cw)
    if kind == "llava-ident":
        return kind, LlavaOracle(H, W), Plan.llava(H, W, H, W)
    if kind == "mllama":
        tile, mt = int(rng.choice([16, 28, 32, 64])), int(rng.integers(1, 5))
        return kind, MllamaOracle(tile=tile, max_tiles=mt), Plan.mllama(H, W, tile=tile, max_tiles=mt)
    if kind == "qwen":
        lo = int(rng.choice([4, 16])) * 28 * 28
        hi = lo * int(rng.choice([1, 4, 16]))
        return kind, Qwen2VLOracle(min_pixels=lo, max_pixels=hi), Plan.qwen2vl(H, W, min_pixels=lo, max_pixels=hi)
    nc = int(rng.choice([1, 4, 6]))
    return kind, Phi3Oracle(num_crops=nc), Plan.phi3(H, W, num_crops=nc)


LARGE = False      # --large: images of 250 k positions and more, also very narrow / very flat ones (the kernels with three
                   # channels per thread, the (chunk, row) partitions and their scratch sizes; the default sizes never get there)


def draw_case(rng):
    H, W = int(rng.integers(12, 140)), int(rng.integers(12, 140))
    if rng.random() < 0.2:
        H = W = int(rng.choice([32, 64, 112]))               # sizes the pair / one-launch chains accept
    if LARGE:
        shape = rng.choice(["square", "narrow", "flat", "odd"])
        if shape == "square":
            H = W = int(rng.choice([512, 520, 600, 672]))
        elif shape == "narrow":
            W = int(rng.integers(20, 90)); H = int(rng.integers(255000 // W, 300000 // W))
        elif shape == "flat":
            H = int(rng.integers(20, 90)); W = int(rng.integers(255000 // H, 300000 // H))
        else:
            H = int(rng.integers(400, 700)); W = int(rng.integers(252000 // H + 1, 320000 // H + 2))
    cross = rng.random() < 0.25
    n_models = int(rng.integers(2, 4)) if cross else 1
    procs = []
    while len(procs) < n_models:
        try:
            procs.append(one_processor(rng, H, W, allow_phi3=(len(procs) == 0)))
        except Exception:
            # a geometry the plan rejects (e.g. Qwen2-VL's smart_resize leaves no 2x2 patch block);
            # fuzz_parity.py checks that the oracle rejects the same ones - draw another processor
            continue
    steps = int(rng.integers(3, 6))
    if LARGE:
        steps = 3                                   # the oracle takes seconds per step at these sizes
    kw = dict(optimizer=str(rng.choice(["adamw", "adamw", "sign"])), lr=float(rng.choice([1e-2, 3e-3, 1e-3])),
              gamma=float(rng.choice([1.0, 0.5, 0.9])), step_size=int(rng.integers(1, 4)))
    desc = dict(H=H, W=W, models=[p[0] for p in procs], steps=steps, **kw)
    if cross:
        kw["weights"] = [float(w) for w in rng.uniform(0.2, 2.0, size=n_models)]
        kw["cross"] = True
    blur = None if rng.random() < 0.55 else int(rng.choice([3, 5, 9, 15]))
    if blur is not None:
        sig = [float(s) for s in rng.uniform(0.1, 2.0, size=steps)]
        kw["blur_kernel"] = blur
        kw["blur_sigma_fn"] = lambda t, sig=sig: sig[t]
        desc["blur"] = (blur, [round(s, 3) for s in sig])
    if rng.random() < 0.3:
        crops = []
        for _ in range(steps):
            h, w = int(rng.integers(max(4, H // 3), H + 1)), int(rng.integers(max(4, W // 3), W + 1))
            crops.append((int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1)), h, w))
        kw["crop_fn"] = lambda t, crops=crops: crops[t]
        kw["fused"] = False                                   # cropping engines are built for the generic chain
        desc["crops"] = crops
    mask = None
    if rng.random() < 0.35:
        kind = str(rng.choice(["corner", "bottom_lines"]))
        size = int(rng.integers(2, min(H, W)))
        mask = P.create_mask(kind, size, (3, H, W))
        desc["mask"] = (kind, size)
    if not cross and rng.random() < 0.25:
        kw["accum"] = int(rng.integers(2, 4))
        desc["accum"] = kw["accum"]
    if rng.random() < 0.15:
        kw["fused"] = False
    desc["fused"] = kw.get("fused", True)
    if (desc["fused"] and not cross and procs[0][0] == "llava-ident" and blur is None and "accum" not in kw
            and procs[0][2].fused_supported()):
        # the identity-resize LLaVA plan may take any of the fused chains
        mode = str(rng.choice(["auto", "pair", "step", "step-noise-ahead", "prepared"]))
        kw["fused_mode"] = mode.split("-")[0]
        if mode == "step-noise-ahead":
            kw["noise_ahead"] = True
        desc["chain"] = mode
    batches = [int(rng.integers(1, 3 if LARGE else 7)) for _ in range(n_models)]
    desc["batches"] = batches
    return desc, procs, batches, steps, mask, kw


def oracle_sensitivity(x0, procs, batches, steps, mask, kw, perturb):
    """How far the ORACLE's own trajectory moves when the upstream gradients it is fed change by
    `perturb` (relative to their largest entry) - the inputs of `_trajectory` replayed on the CPU.
    Returns the relative L2 distances {p, sigma, qerr_mean, x_std, s} between the two oracle runs."""
    from conftest import rel_err
    from oracle.pgd import PGDOracle
    plans = [p[2] for p in procs]
    shapes = [(B * pl.out_shape[0],) + pl.out_shape[1:] for pl, B in zip(plans, batches)]
    out = {}
    runs = []
    for eps in (0.0, perturb):
        ora = PGDOracle(x0, [p[1] for p in procs], lr=kw["lr"], mask=mask, grad_accum_steps=kw.get("accum", 1),
                        blur_kernel=kw.get("blur_kernel"), model_weights=kw.get("weights"), cross_mode=kw.get("cross", False),
                        optimizer=kw["optimizer"], scheduler_gamma=kw["gamma"], scheduler_step_size=kw["step_size"])
        gen = torch.Generator().manual_seed(11)                     # the input stream of _trajectory
        pert = torch.Generator().manual_seed(12)
        all_z = [[torch.randn(s, generator=gen) for s in shapes] for _ in range(steps + 1)]
        trace = []
        for t in range(steps):
            gs = [torch.randn(s, generator=gen) * 0.01 for s in shapes]
            if eps:
                gs = [g + eps * float(g.abs().max()) * torch.randn(g.shape, generator=pert) for g in gs]
            ora.forward(batches, all_z[t], blur_sigma=kw["blur_sigma_fn"](t) if "blur_sigma_fn" in kw else None,
                        crop=kw["crop_fn"](t) if "crop_fn" in kw else None)
            ref = ora.backward_update(gs)
            trace.append(dict(p=ora.p.detach().clone(), s=ref["s"].clone(), sigma=ref["sigma_next"],
                              qerr_mean=ref["qerr_mean"], x_std=ref["x_std"]))
        runs.append(trace)
    for a, b in zip(*runs):
        for k in ("p", "s"):
            out[k] = max(out.get(k, 0.0), rel_err(b[k], a[k]))
        for k in ("sigma", "qerr_mean", "x_std"):
            out[k] = max(out.get(k, 0.0), abs(b[k] - a[k]) / max(abs(a[k]), 1e-12))
    return out


def run_case(T, dev, rng, seed):
    """-> (verdict, desc, worst): verdict is "ok", "ill-conditioned" or a failure text.

    Pixels of p at which the optimiser is discontinuous (|g| of the order of adam_eps: edge tap of a crop window,
    border of a mask) are handled inside `_trajectory` (`_check_p`: the gradient must agree elementwise at the pixel
    and have been tiny there; every accepted pixel is printed).  "ill-conditioned" remains for the statistics: everything
    that is a smooth function of the inputs agrees (the gradient, its norm, the image-fit loss: 1e-7 level) and the
    quantities that moved past the bar (sigma, qerr_mean: one pixel's uint8 truncation flipping in the quantiser) move
    just as far between two runs of the ORACLE whose upstream gradients differ by 3e-7 - the size of the rounding
    differences between the two implementations (`oracle_sensitivity`)."""
    desc, procs, batches, steps, mask, kw = draw_case(rng)
    x0 = torch.rand(3, desc["H"], desc["W"], generator=torch.Generator().manual_seed(seed)) * 1.1 - 0.05
    try:
        worst = T._trajectory(dev, x0, [p[1] for p in procs], [p[2] for p in procs], batches, steps, mask=mask,
                               max_ill=max(T.FUZZ_MAX_ILL, x0.numel() // 10000 if LARGE else 0), **kw)   # tests/test_gpu_fullsize.py's rule for large images
        return "ok", desc, worst
    except AssertionError as e:
        arg = e.args[0] if e.args else None
        if isinstance(arg, tuple) and len(arg) == 3 and isinstance(arg[2], dict):
            worst = arg[2]
            # pixel_values of step t+1 carry the difference of p_t on, so they only have to stay under the bar
            smooth = all(worst.get(k, 0.0) < 2e-6 for k in ("grad", "grad_norm", "imgfit"))
            if smooth:
                sens = oracle_sensitivity(x0, procs, batches, steps, mask, kw, perturb=3e-7)
                over = {k: v for k, v in worst.items() if v >= T.TOL}
                if all(sens.get(k, 0.0) >= 0.2 * v for k, v in over.items()):
                    return "ill-conditioned", desc, dict(worst, **{f"oracle_sensitivity_{k}": sens.get(k, 0.0) for k in over})
            return f"PARITY {arg[0]} {arg[1]:.3e} {worst}", desc, worst
        return f"PARITY {str(e)[:300]}", desc, {}
    except Exception as e:
        return f"EXCEPTION {type(e).__name__}: {e}\n{traceback.format_exc(limit=3)}", desc, {}


def run_relations(dev, rng, seed):
    """Engine-against-engine relations on a random case (no oracle involved):
      * half boundary: pixel_values are the fp32 ones rounded once; a half gradient gives the update its
        widened copy gives - bit for bit;
      * padding tiles kept zero: same values on every element an image reaches, same p - bit for bit;
      * with one plan and no blur / crop / accumulation: the prepared chain against the generic one
        (same arithmetic, different summation order of the statistics) - p to 1e-6.
    -> (verdict, desc)"""
    from adversarialvlm_amd.pgd import PixelPGD
    desc, procs, batches, steps, mask, kw = draw_case(rng)
    if "crop_fn" in kw:
        return "skipped", desc
    H, W = desc["H"], desc["W"]
    x0 = (torch.rand(3, H, W, generator=torch.Generator().manual_seed(seed)) * 1.1 - 0.05).to(dev)
    plans_of = lambda: [type(p[2])(p[2].kind, H, W, [p[2].desc.a0, p[2].desc.a1, p[2].desc.a2, p[2].desc.a3, p[2].desc.a4])
                        for p in procs]
    half = [torch.float16 if rng.random() < 0.5 else torch.bfloat16 for _ in procs]
    if any(pl.out_numel % 4 for pl in plans_of()):
        half = None
    common = dict(lr=kw["lr"], mask=None if mask is None else mask.to(dev), grad_accum_steps=kw.get("accum", 1),
                  blur_kernel=kw.get("blur_kernel"), model_weights=kw.get("weights"), cross_mode=kw.get("cross", False),
                  optimizer=kw["optimizer"], scheduler_gamma=kw["gamma"], scheduler_step_size=kw["step_size"], seed=5,
                  allow_fused=kw.get("fused", True))
    engines = {"f32": PixelPGD(x0, plans_of(), **common), "keep": PixelPGD(x0, plans_of(), noise_on_padding=False, **common)}
    if engines["f32"].mode == "pair":
        del engines["keep"]                       # the identity LLaVA plan has no padding
    if half is not None and engines["f32"].mode != "step":
        engines["half"] = PixelPGD(x0, plans_of(), io_dtype=half if len(half) > 1 else half[0], **common)
    if engines["f32"].mode == "prepared":
        engines["generic"] = PixelPGD(x0, plans_of(), **dict(common, allow_fused=False))
    desc["relations"] = sorted(engines)
    gen = torch.Generator().manual_seed(seed + 1)
    try:
        for t in range(steps):
            bs = kw["blur_sigma_fn"](t) if "blur_sigma_fn" in kw else None
            outs = {k: e.forward(batches, blur_sigma=bs) for k, e in engines.items()}
            grads = [(torch.randn(o.shape, generator=gen) * 0.02).to(dev) for o in outs["f32"]]
            ref = outs["f32"]
            if "half" in outs:
                for a, b, d in zip(ref, outs["half"], half):
                    assert b.dtype == d and torch.equal(a.to(d), b), "half pixel_values are not the rounded fp32 ones"
            if "keep" in outs:
                for a, b, pl in zip(ref, outs["keep"], engines["keep"].plans):
                    lo, hi = pl.live_range()
                    fa, fb = a.reshape(a.shape[0] // pl.out_shape[0], -1), b.reshape(a.shape[0] // pl.out_shape[0], -1)
                    assert torch.equal(fa[:, lo:hi], fb[:, lo:hi]), "kept-padding run differs where an image is"
                    assert not fb[:, :lo].any() and not fb[:, hi:].any(), "padding is not zero"
            if "generic" in outs:
                for a, b in zip(ref, outs["generic"]):
                    assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max())), "prepared vs generic pixel_values"
            scale = [engines["f32"].loss_scale(i) for i in range(len(grads))]
            if "half" in engines:
                rounded = [(g * s).to(d) for g, s, d in zip(grads, scale, half)]
                engines["half"].backward_update(rounded)
                feed = [r.float() for r in rounded]
            else:
                feed = [g * s for g, s in zip(grads, scale)]
            for k, e in engines.items():
                if k != "half":
                    e.backward_update(feed)
            p = engines["f32"].p
            if "half" in engines:
                assert torch.equal(p, engines["half"].p), f"half boundary: p differs at step {t}"
            if "keep" in engines:
                assert torch.equal(p, engines["keep"].p), f"kept padding: p differs at step {t}"
            if "generic" in engines:
                d = float((p - engines["generic"].p).norm() / max(float(p.norm()), 1e-30))
                assert d < 1e-5, f"prepared vs generic p: {d:.2e} at step {t}"
    except AssertionError as e:
        return f"RELATION {e}", desc
    except Exception as e:
        return f"EXCEPTION {type(e).__name__}: {e}\n{traceback.format_exc(limit=4)}", desc
    return "ok", desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--relations", action="store_true", help="engine-against-engine relations instead of the oracle")
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--budget-s", type=float, default=240.0)
    ap.add_argument("--only", type=str, default="", help="comma-separated case numbers: draw every case of the seed's "
                    "stream (the draws are what positions it) but run only these - to replay a reported failure (trajectory mode)")
    ap.add_argument("--large", action="store_true", help="images of 250 k positions and more, extreme aspect ratios included")
    a = ap.parse_args()
    global LARGE
    LARGE = bool(a.large)
    only = {int(x) for x in a.only.split(",") if x.strip()}
    import test_gpu_pgd as T
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    t0, bad, soft, done, seen = time.time(), 0, 0, 0, {}
    for k in range(a.cases):
        if time.time() - t0 > a.budget_s:
            break
        if only and k not in only:
            draw_case(rng)          # advance the stream exactly as a run of this case would
            continue
        if a.relations:
            verdict, desc = run_relations(dev, rng, a.seed * 7919 + k)
            worst = {"-": 0.0}
        else:
            verdict, desc, worst = run_case(T, dev, rng, a.seed * 7919 + k)
        done += 1
        if verdict == "skipped":
            continue
        tag = desc.get("chain", "cross" if len(desc["models"]) > 1 else ("fused-auto" if desc["fused"] else "generic"))
        seen[tag] = seen.get(tag, 0) + 1
        if verdict == "ill-conditioned":
            soft += 1
            print(f"ill-conditioned case {k}: {desc}: { {n: f'{v:.1e}' for n, v in worst.items() if v > 1e-5} }", flush=True)
        elif verdict != "ok":
            bad += 1
            print(f"FAIL case {k}: {desc}: {verdict}", flush=True)
        elif k % 10 == 0 or LARGE:
            print(f"case {k}: {desc}: ok, worst {max(worst.values()):.2e} ({max(worst, key=worst.get)})", flush=True)
    print(f"{done} cases, {bad} failures, {soft} ill-conditioned (see run_case), {time.time() - t0:.0f} s; chains {seen}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
