"""GPU tier: the specialised / merged kernels against the general ones, bit for bit.

csrc/advx_blur.h restates k_blur, k_crop_bwd, blur_fold, k_tanh_bwd and k_bwd_update with the blur radius as a
template parameter and merges the image-level backward into one launch.  Element for element the arithmetic is
the same, so every result must be IDENTICAL to what the general kernels give (`ops.generic_kernels()` forces
them); parity of the general kernels with the oracle is what the other GPU tests establish."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.mark.parametrize("H,W", [(64, 64), (7, 9), (33, 95), (130, 31), (515, 70), (336, 336)])
@pytest.mark.parametrize("k", [3, 5, 9, 15])
def test_blur_forward_specialised_equals_general(H, W, k):
    from adversarialvlm_amd import ops
    if k // 2 > min(H, W) - 1:
        pytest.skip("radius does not fit the reflect padding")
    x = torch.randn(3, H, W, generator=torch.Generator().manual_seed(H * 1000 + W + k)).to(DEV)
    for sigma in (0.3, 1.7, 10.0):
        fast = ops.blur_fwd(x, k, sigma)
        with ops.generic_kernels():
            ref = ops.blur_fwd(x, k, sigma)
        assert torch.equal(fast, ref), (H, W, k, sigma, float((fast - ref).abs().max()))


def _run(H, W, k, crop, steps, generic, accumulate=1, optimizer="adamw", masked=True, seed=0):
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(seed)
    x0 = torch.rand(3, H, W, generator=gen)
    mask = (torch.rand(3, H, W, generator=gen) > 0.3).float() if masked else None
    oh, ow = max(8, H * 2 // 3), max(8, W * 2 // 3)
    B = 3
    zs = [torch.randn(B, 3, oh, ow, generator=gen) for _ in range(steps)]
    gs = [torch.randn(B, 3, oh, ow, generator=gen) * 0.05 for _ in range(steps)]
    windows = [(1, 2, H - 3, W - 4), (0, 0, H, W), (H // 5, W // 7, H - H // 4, W - W // 3)]
    sigmas = [0.4, 2.5, 9.0]

    def go():
        eng = PixelPGD(x0.to(DEV), [Plan.llava(H, W, oh, ow)], lr=1e-2, blur_kernel=k, use_crop=crop, mask=mask,
                       grad_accum_steps=accumulate, optimizer=optimizer, allow_fused=False, scheduler_step_size=2,
                       scheduler_gamma=0.7)
        outs = []
        for t in range(steps):
            pv = eng.forward(B, [zs[t].to(DEV)], blur_sigma=sigmas[t % 3], crop=windows[t % 3] if crop else None)[0]
            eng.backward_update([gs[t].to(DEV)])
            outs.append((pv.clone(), eng.p.clone(), eng.grad.clone(), eng.m.clone(), eng.v.clone(), eng.image().clone()))
        return outs, eng.stats_dict()

    if generic:
        with ops.generic_kernels():
            return go()
    # the merged kernels with the crop window resized on its own (two launches each way): what is bit-identical to the
    # general kernels; the composed window o plan table is compared with this form in tests/test_gpu_compose.py
    with ops.separate_crop():
        return go()


@pytest.mark.parametrize("H,W,k", [(64, 64, 5), (40, 52, 9), (20, 24, 5), (97, 130, 9), (70, 33, 15), (130, 45, 3)])
@pytest.mark.parametrize("crop", [False, True])
def test_merged_image_backward_equals_the_separate_kernels(H, W, k, crop):
    """forward blur (radius-templated) and the merged backward (crop^T on load, blur^T, fold, tanh', mask,
    optimiser in one launch) over three steps with changing sigma and window: every tensor identical."""
    fast, st_f = _run(H, W, k, crop, 3, generic=False)
    ref, st_r = _run(H, W, k, crop, 3, generic=True)
    for t, (a, b) in enumerate(zip(fast, ref)):
        for name, u, v in zip(("pixel_values", "p", "grad", "m", "v", "image"), a, b):
            assert torch.equal(u, v), (t, name, float((u - v).abs().max()))
    for key in st_f:
        assert st_f[key] == pytest.approx(st_r[key], rel=1e-6, abs=1e-12), key


@pytest.mark.parametrize("accumulate,optimizer,masked", [(2, "adamw", True), (1, "sign", False), (3, "sign", True)])
def test_merged_image_backward_accumulation_and_sign(accumulate, optimizer, masked):
    fast, st_f = _run(66, 80, 9, True, 6, generic=False, accumulate=accumulate, optimizer=optimizer, masked=masked, seed=4)
    ref, st_r = _run(66, 80, 9, True, 6, generic=True, accumulate=accumulate, optimizer=optimizer, masked=masked, seed=4)
    for t, (a, b) in enumerate(zip(fast, ref)):
        for name, u, v in zip(("pixel_values", "p", "grad", "m", "v", "image"), a, b):
            assert torch.equal(u, v), (t, name, float((u - v).abs().max()))
    assert st_f["grad_norm"] == pytest.approx(st_r["grad_norm"], rel=1e-6)


@pytest.mark.parametrize("crop", [False, True])
def test_merged_gradient_only_backward(crop):
    """advx_image_bwd (the data-parallel form: unmasked gradient, the all-reduce follows) through the same merged
    kernel: identical to k_crop_bwd + k_blur<1,2> + k_tanh_bwd<true>, with and without accumulation."""
    from adversarialvlm_amd import ops
    H, W, k = 75, 100, 9
    gen = torch.Generator().manual_seed(11)
    p = (torch.randn(3, H, W, generator=gen) * 0.5).to(DEV)
    x0 = torch.rand(3, H, W, generator=gen).to(DEV)
    garg = torch.randn(3, H, W, generator=gen).to(DEV)
    window = (5, 9, 60, 80) if crop else None
    stats = torch.zeros(16, device=DEV)
    res = []
    for generic in (False, True):
        def go():
            scratch = ops.image_scratch(H, W, k, DEV)
            s, _ = ops.image_fwd(p, x0, 0.5, stats.clone(), scratch, blur=(k, 1.3), crop=window)
            g = torch.full_like(p, 0.25)
            ops.image_bwd(p, s, garg, 0.5, 0.7, g, scratch, blur=(k, 1.3), crop=window, accumulate=True)
            g2 = torch.empty_like(p)
            ops.image_bwd(p, s, garg, 0.5, 0.7, g2, scratch, blur=(k, 1.3), crop=window, accumulate=False)
            return s, g, g2
        if generic:
            with ops.generic_kernels():
                res.append(go())
        else:
            res.append(go())
    for u, v in zip(*res):
        assert torch.equal(u, v), float((u - v).abs().max())


# ------------------------------------------------------------------ windowed resizes (advx_resize.h)
def _plans():
    from adversarialvlm_amd.plan import Plan
    return {
        "llava512": lambda: Plan.llava(512, 512),
        "llava_odd": lambda: Plan.llava(97, 130, 56, 72),
        "llava_up": lambda: Plan.llava(40, 52, 64, 64),
        "mllama336": lambda: Plan.mllama(336, 336),
        "mllama_wide": lambda: Plan.mllama(300, 1000, tile=64),
        "phi3_512": lambda: Plan.phi3(512, 512),
        "phi3_tall": lambda: Plan.phi3(700, 300),
        "qwen512": lambda: Plan.qwen2vl(512, 512),
        "qwen_small": lambda: Plan.qwen2vl(60, 90, min_pixels=28 * 28, max_pixels=28 * 28 * 16),
    }


@pytest.mark.parametrize("name", sorted(_plans()))
def test_windowed_forward_resizes_equal_general(name):
    """process() of every plugin through k_stage_fwd_t / k_stage0_fwd_multi_t against the run-time-loop kernels."""
    from adversarialvlm_amd import ops
    plan = _plans()[name]()
    img = torch.rand(3, plan.in_h, plan.in_w, generator=torch.Generator().manual_seed(5)).to(DEV)
    fast = ops.emit(plan, img, 2)
    fast_m = ops.emit_multi([plan], img, [2])[0]
    with ops.generic_kernels():
        ref = ops.emit(plan, img, 2)
    assert torch.equal(fast, ref), float((fast - ref).abs().max())
    assert torch.equal(fast_m, ref)


@pytest.mark.parametrize("window", [(3, 5, 50, 70), (0, 0, 64, 96), (10, 20, 17, 23), (30, 1, 34, 95)])
def test_windowed_crop_equals_general(window):
    from adversarialvlm_amd import ops
    gen = torch.Generator().manual_seed(2)
    img = torch.rand(3, 64, 96, generator=gen).to(DEV)
    g = torch.randn(3, 64, 96, generator=gen).to(DEV)
    fast = (ops.crop_resize_fwd(img, window), ops.crop_resize_bwd(g, window))
    with ops.generic_kernels():
        ref = (ops.crop_resize_fwd(img, window), ops.crop_resize_bwd(g, window))
    assert torch.equal(fast[0], ref[0]) and torch.equal(fast[1], ref[1])


@pytest.mark.parametrize("H,W,window", [(512, 512, (40, 30, 400, 420)), (500, 530, (0, 0, 500, 530)), (600, 450, (100, 50, 160, 170))])
def test_three_channel_gathers_equal_general(H, W, window):
    """From 250 k positions up the crop adjoint and the transposed resize handle the three channels of a position in
    one thread (k_crop_bwd_rows3, k_stage_bwd3): same bits as the general kernels."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(H + W)
    g = torch.randn(3, H, W, generator=gen).to(DEV)
    plan = Plan.llava(H, W)
    gout = (torch.randn(2, plan.out_numel, generator=gen) * 0.1).to(DEV)
    acc0 = torch.randn(3, H, W, generator=gen).to(DEV)

    def go():
        pv = ops.emit(plan, g, 2)
        a = ops.collect(plan, gout.view_as(pv), 2)
        b = ops.collect(plan, gout.view_as(pv), 2, grad_argument=acc0.clone(), accumulate=True)
        return ops.crop_resize_bwd(g, window), a, b

    fast = go()
    with ops.generic_kernels():
        ref = go()
    for u, v in zip(fast, ref):
        assert torch.equal(u, v), float((u - v).abs().max())


@pytest.mark.parametrize("name", ["qwen512", "phi3_512", "mllama_big"])
def test_three_channel_gathers_other_layouts(name):
    """k_stage_bwd3 behind a patch layout with two temporal copies of the sums (Qwen2-VL), behind a second stage whose
    gradient is added into the sums (Phi-3.5) and behind tiles (Mllama), at sizes that take it: the general kernels' bits."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    mk = {"qwen512": lambda: Plan.qwen2vl(512, 512), "phi3_512": lambda: Plan.phi3(512, 512),
          "mllama_big": lambda: Plan.mllama(600, 520)}[name]
    gen = torch.Generator().manual_seed(3)
    plan0 = mk()
    img = torch.rand(3, plan0.in_h, plan0.in_w, generator=gen).to(DEV)
    gout = (torch.randn(2, plan0.out_numel, generator=gen) * 0.1).to(DEV)

    def go():
        plan = mk()
        pv = ops.emit(plan, img, 2)
        return pv, ops.collect(plan, gout.view_as(pv), 2)

    fast = go()
    with ops.generic_kernels():
        ref = go()
    for u, v in zip(fast, ref):
        assert torch.equal(u, v), (name, float((u - v).abs().max()))


@pytest.mark.parametrize("name", ["llava512", "llava_odd", "mllama_wide", "phi3_tall", "qwen_small", "phi3_512", "qwen512"])
def test_prepared_chain_windowed_equals_general(name):
    """Three steps of the prepared chain (k_plan_head through the windowed kernel, ||g|| reduced by its block 0)."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    gen = torch.Generator().manual_seed(8)
    plan0 = _plans()[name]()
    x0 = torch.rand(3, plan0.in_h, plan0.in_w, generator=gen)
    B = 2
    gs = [torch.randn(B, plan0.out_numel, generator=gen) * 0.05 for _ in range(3)]

    def go():
        eng = PixelPGD(x0.to(DEV), [_plans()[name]()], lr=1e-2, fused_mode="prepared")
        outs = []
        for t in range(3):
            pv = eng.forward(B)[0]
            eng.backward_update([gs[t].to(DEV).view_as(pv)])
            outs.append((pv.clone(), eng.p.clone()))
        return outs, eng.stats_dict()

    fast, st_f = go()
    with ops.generic_kernels():
        ref, st_r = go()
    for (a, b), (c, d) in zip(fast, ref):
        assert torch.equal(a, c) and torch.equal(b, d)
    for key in st_f:
        assert st_f[key] == pytest.approx(st_r[key], rel=1e-6, abs=1e-12), key


def test_cross_chain_windowed_equals_general():
    """Three plans over one image with blur (the multi-plan launches) against the general kernels."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(21)
    H, W = 120, 150
    x0 = torch.rand(3, H, W, generator=gen)
    mk = lambda: [Plan.phi3(H, W), Plan.qwen2vl(H, W, min_pixels=28 * 28, max_pixels=28 * 28 * 36), Plan.mllama(H, W, tile=48)]
    plans = mk()
    Bs = [2, 3, 2]
    gs = [[torch.randn(b, pl.out_numel, generator=gen) * 0.05 for pl, b in zip(plans, Bs)] for _ in range(3)]

    def go():
        eng = PixelPGD(x0.to(DEV), mk(), lr=1e-2, blur_kernel=5, cross_mode=True, model_weights=[0.2, 0.8, 1.6])
        outs = []
        for t in range(3):
            pvs = eng.forward(Bs, blur_sigma=0.3 + 0.6 * t)
            eng.backward_update([g.to(DEV).view_as(pv) for g, pv in zip(gs[t], pvs)])
            outs.append([pv.clone() for pv in pvs] + [eng.p.clone(), eng.grad.clone()])
        return outs

    fast = go()
    with ops.generic_kernels():
        ref = go()
    for a, b in zip(fast, ref):
        for u, v in zip(a, b):
            assert torch.equal(u, v), float((u - v).abs().max())


def _trim_plans():
    from adversarialvlm_amd.plan import Plan
    return {
        # equal sizes: bicubic rows (0, 1, 0, 0) - the reference's cross preset runs Phi-3.5 and Qwen2-VL at 336 x 336
        "qwen336": lambda: Plan.qwen2vl(336, 336),
        "phi3_336": lambda: Plan.phi3(336, 336),
        "qwen_equal_small": lambda: Plan.qwen2vl(56, 84, min_pixels=28 * 28, max_pixels=28 * 28 * 16),
        "mllama336": lambda: Plan.mllama(336, 336),
        "llava_up2": lambda: Plan.llava(32, 48, 64, 96),           # exact 2x: bilinear-family rows with zero ends
        "llava512": lambda: Plan.llava(512, 512),
        "phi3_tall": lambda: Plan.phi3(700, 300),
    }


@pytest.mark.parametrize("name", sorted(_trim_plans()))
def test_trimmed_tap_rows_equal_full_rows(name):
    """The device tables drop zero-weight taps at the ends of a row (advx_plan_upload); a plan uploaded under
    ops.full_tap_rows() keeps ATen's rows.  Forward, gradient and three prepared steps: identical bits."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    gen = torch.Generator().manual_seed(13)
    mk = _trim_plans()[name]
    plan0 = mk()
    x0 = torch.rand(3, plan0.in_h, plan0.in_w, generator=gen)
    img = x0.to(DEV)
    B = 2
    gs = [torch.randn(B, plan0.out_numel, generator=gen) * 0.05 for _ in range(3)]

    def go():
        plan = mk()
        res = [ops.emit(plan, img, B)]
        res.append(ops.collect(plan, gs[0].to(DEV).view_as(res[0]), B))
        eng = PixelPGD(x0.to(DEV), [mk()], lr=1e-2, fused_mode="prepared")
        for t in range(3):
            pv = eng.forward(B)[0]
            eng.backward_update([gs[t].to(DEV).view_as(pv)])
            res += [pv.clone(), eng.p.clone()]
        eng = PixelPGD(x0.to(DEV), [mk()], lr=1e-2, blur_kernel=3, allow_fused=False)
        for t in range(2):
            pv = eng.forward(B, blur_sigma=0.8)[0]
            eng.backward_update([gs[t].to(DEV).view_as(pv)])
            res += [pv.clone(), eng.p.clone(), eng.grad.clone()]
        return res

    fast = go()
    with ops.full_tap_rows():
        ref = go()
    assert len(fast) == len(ref)
    for u, v in zip(fast, ref):
        assert torch.equal(u, v), (name, float((u - v).abs().max()))


@pytest.mark.parametrize("keep", [False, True])
def test_cross_chain_with_crop_mixed_boundaries_equals_general(keep):
    """Three plans with different boundary dtypes over one image, blur AND a crop window per step (the window's
    transposed tables ride in the merged emit; the reductions cannot merge: different load variants), padding kept
    zero or not: the general kernels' bits."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(33)
    H, W = 120, 152
    x0 = torch.rand(3, H, W, generator=gen)
    mk = lambda: [Plan.phi3(H, W), Plan.qwen2vl(H, W, min_pixels=28 * 28, max_pixels=28 * 28 * 36), Plan.mllama(H, W, tile=48)]
    dts = [torch.float16, torch.float32, torch.bfloat16]
    plans = mk()
    Bs = [2, 3, 2]
    gs = [[torch.randn(b, pl.out_numel, generator=gen) * 0.05 for pl, b in zip(plans, Bs)] for _ in range(3)]
    windows = [(3, 5, 100, 120), (0, 0, H, W), (20, 30, 64, 70)]

    def go():
        eng = PixelPGD(x0.to(DEV), mk(), lr=1e-2, blur_kernel=5, use_crop=True, cross_mode=True, model_weights=[0.2, 0.8, 1.6],
                       io_dtype=dts, noise_on_padding=not keep, seed=9)
        outs = []
        for t in range(3):
            pvs = eng.forward(Bs, blur_sigma=0.4 + 0.5 * t, crop=windows[t])
            eng.backward_update([g.to(DEV).to(pv.dtype).view_as(pv) for g, pv in zip(gs[t], pvs)])
            outs.append([pv.clone() for pv in pvs] + [eng.p.clone(), eng.grad.clone()])
        return outs

    fast = go()
    with ops.generic_kernels():
        ref = go()
    for a, b in zip(fast, ref):
        for u, v in zip(a, b):
            assert u.dtype == v.dtype and torch.equal(u, v), float((u.float() - v.float()).abs().max())


def test_merged_reductions_streamed_loads():
    """k_batch_reduce_multi with the non-temporal load variant: every plan's gradient is beyond the Infinity Cache
    (Phi-3.5 and Qwen2-VL at 512 x 512, 64 prompts each: 465 MB and 390 MB read).  Same image gradient, bit for bit, as
    one k_batch_reduce per plan."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    mk = lambda: [Plan.phi3(512, 512), Plan.qwen2vl(512, 512)]
    plans = mk()
    B = 64
    gen = torch.Generator(device=DEV).manual_seed(17)
    grads = [torch.randn(B, pl.out_numel, generator=gen, device=DEV) * 0.01 for pl in plans]

    def go():
        return ops.collect_multi(mk(), grads, [B, B])

    fast = go()
    with ops.generic_kernels():
        ref = go()
    assert torch.equal(fast, ref), float((fast - ref).abs().max())


def test_large_image_takes_the_general_backward():
    """Beyond 2048 tiles (here 33 x 33 x 3) the merged backward has no room for its ||g|| partials and the host falls
    back to k_blur<1,2> + k_bwd_update<1>; the forward stays radius-templated.  Same bits either way."""
    fast, st_f = _run(1040, 1040, 5, False, 2, generic=False, masked=False)
    ref, st_r = _run(1040, 1040, 5, False, 2, generic=True, masked=False)
    for a, b in zip(fast, ref):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    assert st_f["grad_norm"] == pytest.approx(st_r["grad_norm"], rel=1e-6)


def test_unknown_tuning_switch_is_an_error():
    from adversarialvlm_amd import _lib as L
    assert L.load().advx_set_tuning(99, 1) != 0 and b"unknown switch" in L.load().advx_last_error()
