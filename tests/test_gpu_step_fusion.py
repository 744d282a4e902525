"""GPU tier: step-to-step fusion of the blur chains (advx_image_step / advx_forward_multi_ready, round 4).

The backward of step t also runs the first image kernel of step t+1 (eps*tanh of the updated p, blur, s = x0 + blur,
statistics partials, forward rows of the next crop window's composed tables) by halo recompute.  The contract is "bit for bit
the two calls it replaces" - checked at the C-ABI level (ops), at the engine level over several steps, and through the
single trainer's checkpoints (a resumed run continues bit for bit although the crop window is drawn one iteration ahead).
Reference lines: attack_model.py:300-312 (next iteration) moved behind :335-346 (this iteration)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _opt(L, kind, t, lr=1e-2):
    o = L.OptScalars()
    o.kind, o.apply = kind, 1
    b1, b2 = 0.9, 0.999
    o.lr, o.decay, o.w1, o.beta2, o.w2 = lr, 1 - lr * 0.01, 1 - b1, b2, 1 - b2
    o.bias2_sqrt = (1 - b2 ** (t + 1)) ** 0.5
    o.eps = 1e-8
    o.neg_step_size = -(lr / (1 - b1 ** (t + 1)))
    return o


GEOMS = [  # (H, W, kernel size, crop window of the NEXT step or None)
    (64, 50, 5, None), (46, 46, 9, None), (97, 130, 9, (5, 9, 80, 100)), (96, 160, 3, (0, 0, 96, 160)),
    (121, 67, 7, None), (336, 336, 5, (30, 20, 280, 300)), (512, 512, 9, (20, 30, 400, 420)), (200, 300, 9, None),
    (43, 200, 7, None),       # exactly 32 + 3r + 2 rows (r = 3)
]


@pytest.mark.parametrize("H,W,k,next_crop", GEOMS)
@pytest.mark.parametrize("kind", ["adamw", "sign"])
def test_image_step_equals_the_two_calls_it_replaces(H, W, k, next_crop, kind):
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    assert ops.image_step_supported(H, W, k)
    gen = torch.Generator().manual_seed(H * 1000 + W + k)
    x0 = torch.rand(3, H, W, generator=gen).to(DEV)
    p0 = (torch.randn(3, H, W, generator=gen) * 0.6).to(DEV)
    m0 = (torch.randn(3, H, W, generator=gen) * 1e-3).to(DEV)
    v0 = (torch.rand(3, H, W, generator=gen) * 1e-5).to(DEV)
    mask = (torch.rand(3, H, W, generator=gen) > 0.2).float().to(DEV)
    g_img = (torch.randn(3, H, W, generator=gen) * 1e-3).to(DEV)
    plan = Plan.llava(H, W, max(24, 2 * H // 3), max(24, 2 * W // 3))
    if next_crop is not None:
        assert ops.crop_composes(plan, H, W, next_crop)
    o = _opt(L, L.OPT_ADAMW if kind == "adamw" else L.OPT_SIGN, 3)
    sig_t, sig_n = 1.7, 0.9
    B = 2

    def fresh():
        stats = torch.zeros(L.STATS_N, device=DEV)
        stats[L.STAT_QERR_STD] = 0.01
        return dict(p=p0.clone(), m=m0.clone(), v=v0.clone(), grad=torch.zeros_like(p0), stats=stats,
                    scr=ops.image_scratch(H, W, k, DEV), upd=ops.update_scratch(p0.numel(), DEV), s=torch.empty_like(p0),
                    ws=torch.empty(plan.workspace_floats, device=DEV))

    # step t's image (what both forms read as `s`)
    a, b = fresh(), fresh()
    for st in (a, b):
        ops.image_fwd(st["p"], x0, 0.5, st["stats"], st["scr"], blur=(k, sig_t), s=st["s"])
    assert torch.equal(a["s"], b["s"])
    # (A) the two calls: backward + update in place, then the next step's forward_multi
    ops.image_bwd_update(a["p"], a["s"], g_img, 0.5, 1.0, a["grad"], mask, a["m"], a["v"], o, a["stats"], a["scr"], a["upd"],
                         blur=(k, sig_t), finalize_norm=True)
    s_next_a = torch.empty_like(p0)
    outs_a, _ = ops.forward_multi(a["p"], x0, 0.5, a["stats"], a["scr"], [plan], [B], s_next_a, blur=(k, sig_n), crop=next_crop,
                                  philox=(7, [11]), workspaces=[a["ws"]])
    # (B) one launch for both, then the ready forward
    p2, m2, v2, s_next_b = torch.full_like(p0, 9.0), torch.full_like(p0, 9.0), torch.full_like(p0, 9.0), torch.full_like(p0, 9.0)
    ops.image_step(b["p"], b["m"], b["v"], p2, m2, v2, b["s"], g_img, 0.5, 1.0, b["grad"], mask, o, b["scr"], b["upd"], x0,
                   s_next_b, (k, sig_t), sig_n, next_crop=next_crop, next_plan=plan)
    ops.update_flush(p0.numel(), b["stats"], b["upd"])
    assert torch.equal(b["p"], p0) and torch.equal(b["m"], m0) and torch.equal(b["v"], v0)        # the step's state is read only
    outs_b, _ = ops.forward_multi(p2, x0, 0.5, b["stats"], b["scr"], [plan], [B], s_next_b, blur=(k, sig_n), crop=next_crop,
                                  philox=(7, [11]), workspaces=[b["ws"]], image_ready=True)
    torch.cuda.synchronize()
    assert torch.equal(p2, a["p"]), float((p2 - a["p"]).abs().max())
    if kind == "adamw":
        assert torch.equal(m2, a["m"]) and torch.equal(v2, a["v"])
    assert torch.equal(b["grad"], a["grad"])
    assert torch.equal(s_next_b, s_next_a), float((s_next_b - s_next_a).abs().max())
    assert torch.equal(outs_b[0], outs_a[0])
    assert torch.equal(b["stats"], a["stats"]), (b["stats"].tolist(), a["stats"].tolist())
    # ... and the backward of the next step finds the composed tables the ready forward finished
    if next_crop is not None:
        g_out = torch.randn(B, plan.out_numel, generator=gen).to(DEV)
        ga = ops.collect_crop(plan, g_out, B, next_crop, a["scr"], grad_s=torch.empty_like(p0), workspace=a["ws"])
        gb = ops.collect_crop(plan, g_out, B, next_crop, b["scr"], grad_s=torch.empty_like(p0), workspace=b["ws"])
        assert torch.equal(ga, gb)


def test_image_step_refuses_what_it_does_not_cover():
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    assert not ops.image_step_supported(42, 200, 7)          # one row short of 32 + 3r + 2 (r = 3)
    assert not ops.image_step_supported(336, 336, 11)        # kernel sizes 3..9
    assert not ops.image_step_supported(336, 336, 4) and not ops.image_step_supported(336, 336, 0)
    with ops.generic_kernels():
        assert not ops.image_step_supported(336, 336, 5)
    H = W = 64
    x = torch.rand(3, H, W, device=DEV)
    z = torch.zeros_like(x)
    o = _opt(L, L.OPT_ADAMW, 0)
    scr, upd = ops.image_scratch(H, W, 5, DEV), ops.update_scratch(x.numel(), DEV)
    with pytest.raises(L.AdvxError, match="other buffers"):
        ops.image_step(z, z.clone(), z.clone(), z, z.clone(), z.clone(), x, x, 0.5, 1.0, z.clone(), x, o, scr, upd, x, z.clone(),
                       (5, 1.0), 1.0)
    o.apply = 0
    with pytest.raises(L.AdvxError, match="always steps"):
        ops.image_step(z, z.clone(), z.clone(), z.clone(), z.clone(), z.clone(), x, x, 0.5, 1.0, z.clone(), x, o, scr, upd, x,
                       z.clone(), (5, 1.0), 1.0)


@pytest.mark.parametrize("H,W,k,use_crop,nplans", [(96, 80, 9, True, 1), (70, 70, 5, False, 1), (336, 336, 5, False, 3),
                                                   (512, 512, 9, True, 1)])
def test_engine_with_and_without_the_fused_step_is_bit_identical(H, W, k, use_crop, nplans):
    """PixelPGD told the next step's blur sigma / window (one launch for backward + next image kernel) against PixelPGD left
    to recompute: pixel_values, image, p, m, v, gradient and every statistic equal over six steps - including a step whose
    announced window is NOT the one the next forward asks for (the engine then simply recomputes)."""
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(3)
    x0 = torch.rand(3, H, W, generator=gen).to(DEV)
    mask = (torch.rand(3, H, W, generator=gen) > 0.1).float().to(DEV)

    def plans():
        if nplans == 1:
            return [Plan.llava(H, W, 56, 56) if H < 300 else Plan.llava(H, W)]
        return [Plan.phi3(H, W), Plan.qwen2vl(H, W), Plan.mllama(H, W)]
    B = 2
    wins = [(5, 4, H - 20, W - 16), (0, 3, H - 7, W - 9), (8, 8, H - 16, W - 24), (2, 2, H - 30, W - 4), (10, 1, H - 12, W - 10),
            (4, 6, H - 8, W - 12), (1, 1, H - 2, W - 2)]
    sigmas = [1.3, 0.4, 1.9, 0.7, 1.1, 0.25, 1.5]
    engs = [PixelPGD(x0, plans(), lr=2e-2, mask=mask, blur_kernel=k, use_crop=use_crop, cross_mode=nplans > 1, seed=5,
                     allow_fused=False, step_fusion=True) for _ in range(2)]
    assert all(e.step_fusion for e in engs)
    outs = [[], []]
    for t in range(6):
        crop = wins[t] if use_crop else None
        gs = None
        for which, eng in enumerate(engs):
            pvs = eng.forward(B, blur_sigma=sigmas[t], crop=crop)
            if gs is None:
                gs = [torch.randn(pv.shape, generator=gen).to(DEV) * 1e-3 for pv in pvs]
            outs[which].append([pv.clone() for pv in pvs] + [eng.image().clone()])
            if which == 0:
                eng.backward_update(gs)
            else:
                # step 3 announces a window / sigma the next forward will not use: the engine must notice and recompute
                nc = (wins[t + 1] if t != 3 else wins[6]) if use_crop else None
                ns = sigmas[t + 1] if (t != 3 or use_crop) else 0.333
                eng.backward_update(gs, next_blur_sigma=ns, next_crop=nc)
        a, b = engs
        assert torch.equal(a.p, b.p) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v) and torch.equal(a.grad, b.grad), t
        assert torch.equal(a.image(), b.image())
        assert a.stats_dict() == b.stats_dict(), (t, a.stats_dict(), b.stats_dict())
    for xa, xb in zip(*outs):
        for u, w in zip(xa, xb):
            assert torch.equal(u, w)
    assert engs[1]._next_ready is not None and engs[0]._next_ready is None


def _gray(tmp, size):
    from PIL import Image
    path = os.path.join(tmp, f"gray{size}.png")
    Image.fromarray(np.full((size, size, 3), 128, dtype=np.uint8)).save(path)
    return path


@pytest.mark.parametrize("use_crop", [False, True])
def test_trainer_resume_with_blur_draws_the_same_windows(tmp_path, use_crop):
    """The single trainer with blur 9 (+ a crop window per iteration) takes the fused step, drawing each window one iteration
    ahead; its checkpoints store the generator from before that draw, so 4 iterations + resume + 3 more equal 7 in one go,
    bit for bit, and both equal a run with the fusion off (the default)."""
    from adversarialvlm_amd import attack_model, ops
    tmp = str(tmp_path)
    img = _gray(tmp, 72)

    def kw(name, iters, **extra):
        d = dict(exp_name=name, img_orig=img, prompt="list", target_text="sure here it is", model_name="synthetic/tiny-llava",
                 lr=1e-2, num_iterations=iters, save_steps=3, batch_size=3, grad_accum_steps=1, scheduler_step_size=2,
                 scheduler_gamma=0.8, restart_num=0, mask_type=None, mask_size=None, clamp_method="tanh", epsilon=0.5, sigma=1e-3,
                 start_from_white=False, target_text_random=False, base_path=tmp, seed=3, use_gaussian_blur=True,
                 gblur_kernel_size=9, gblur_sigma=2.0, use_local_crop=use_crop, step_fusion=True)
        d.update(extra)
        return d
    eng, hist = attack_model.train(**kw("full", 7, return_engine=True))
    assert eng.step_fusion and eng.mode == "generic"
    attack_model.train(**kw("part", 4))
    attack_model.train(**kw("rest", 7, resume_from=os.path.join(tmp, "part", "state_iter_4.pt")))
    plain_eng, plain = attack_model.train(**kw("plain", 7, return_engine=True, step_fusion=False))
    assert not plain_eng.step_fusion

    def final(name):
        return np.fromfile(os.path.join(tmp, name, "optimized_image_iter_final.bin"), dtype=np.float32)
    assert np.array_equal(final("full"), final("rest"))
    assert np.array_equal(final("full"), final("plain"))
    # what the owned path logs is equal to the bit; the VLM's own loss (torch GEMMs and reductions on the GPU) only to rounding
    for key in ("image_loss", "grad norm", "resave_error_std", "resave_error_mean", "adversarial_mean", "adversarial_std", "lr"):
        assert [h[key] for h in hist] == [h[key] for h in plain], key
    assert [h["ce_loss"] for h in hist] == pytest.approx([h["ce_loss"] for h in plain], rel=1e-5)
