"""GPU tier: advx_collect_update against the two calls it replaces (advx_collect / advx_collect_crop, then advx_image_bwd_update
without blur), tensor for tensor, bit for bit - AdamW and the sign step, first and later iterations of an accumulation window,
with and without applying the step, every processor family, with and without a composing crop window.  ||g|| comes from another
partition of the image (doubles): relative 1e-6.  Where the one-launch form is not offered (long transposed rows, a window that does
not compose) the call must say so and launch nothing."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _opt(L, kind, t, apply=1, lr=1e-2):
    o = L.OptScalars()
    o.kind, o.apply = kind, apply
    b1, b2 = 0.9, 0.999
    o.lr, o.decay, o.w1, o.beta2, o.w2 = lr, 1 - lr * 0.01, 1 - b1, b2, 1 - b2
    o.bias2_sqrt = (1 - b2 ** (t + 1)) ** 0.5
    o.eps = 1e-8
    o.neg_step_size = -(lr / (1 - b1 ** (t + 1)))
    return o


def _plan(maker, H, W):
    from adversarialvlm_amd.plan import Plan
    q = dict(min_pixels=28 * 28, max_pixels=28 * 28 * 1280)
    return {"llava": lambda: Plan.llava(H, W), "qwen2vl": lambda: Plan.qwen2vl(H, W, **q), "phi3": lambda: Plan.phi3(H, W),
            "mllama": lambda: Plan.mllama(H, W)}[maker]()


CASES = [  # (maker, H, W, crop window)
    ("llava", 336, 336, (20, 30, 280, 300)), ("mllama", 336, 336, None), ("llava", 97, 130, (3, 5, 80, 101)), ("qwen2vl", 200, 300, None),
    ("llava", 512, 512, None), ("llava", 512, 512, (40, 30, 400, 420)), ("llava", 500, 640, None),
    ("llava", 672, 672, (0, 0, 672, 672)), ("qwen2vl", 512, 512, None), ("qwen2vl", 512, 512, (16, 24, 470, 450)),
    ("phi3", 512, 512, None), ("phi3", 600, 520, (10, 20, 560, 480)), ("mllama", 600, 600, None),
    ("mllama", 520, 700, (8, 8, 500, 640)), ("llava", 1030, 770, None), ("llava", 1500, 1400, (100, 50, 1200, 1300)),
    ("llava", 1365, 340, None),          # 3 chunks x 683 row pairs = 2049 workgroups unless the last partial group is counted
]


@pytest.mark.parametrize("maker,H,W,crop", CASES)
@pytest.mark.parametrize("kind,accumulate,apply,B", [("adamw", 0, 1, 2), ("adamw", 1, 1, 1), ("sign", 0, 1, 3), ("adamw", 1, 0, 2)])
def test_collect_update_equals_the_two_calls(maker, H, W, crop, kind, accumulate, apply, B):
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    plan = _plan(maker, H, W)
    if not ops.collect_update_supported(plan, H, W, crop):
        pytest.skip("the library does not offer the one-launch form for this geometry (covered by the refusal test)")
    # B = 1, 2: a plain float32 plan's prompts are summed inside the gather (ADVX_TUNE_DIRECT_BATCH); 3: batch reduction first
    gen = torch.Generator().manual_seed(H * 1000 + W + (crop[2] if crop else 0))
    x0 = torch.rand(3, H, W, generator=gen).to(DEV)
    p0 = (torch.randn(3, H, W, generator=gen) * 0.6).to(DEV)
    m0 = (torch.randn(3, H, W, generator=gen) * 1e-3).to(DEV)
    v0 = (torch.rand(3, H, W, generator=gen) * 1e-5).to(DEV)
    g0 = (torch.randn(3, H, W, generator=gen) * 1e-3).to(DEV)
    mask = (torch.rand(3, H, W, generator=gen) > 0.2).float().to(DEV)
    go = (torch.randn(B, plan.out_numel, generator=gen) * 0.05).to(DEV)
    eps = 0.3
    s = (x0 + eps * torch.tanh(p0)).contiguous()
    opt = _opt(L, L.OPT_ADAMW if kind == "adamw" else L.OPT_SIGN, 3, apply)

    def run(fused):
        p, m, v, grad = p0.clone(), m0.clone(), v0.clone(), g0.clone()
        stats = torch.zeros(L.STATS_N, dtype=torch.float32, device=DEV)
        img_scratch = ops.image_scratch(H, W, 0, DEV)
        upd_scratch = ops.update_scratch(p.numel(), DEV)
        ws = torch.zeros(plan.workspace_floats, dtype=torch.float32, device=DEV)
        if fused:
            ops.collect_update(plan, go, B, p, s, eps, 0.7, grad, mask, m, v, opt, stats, img_scratch, upd_scratch, crop=crop,
                               accumulate=bool(accumulate), workspace=ws)
        else:
            garg = torch.empty_like(x0)
            if crop is not None:
                ops.collect_crop(plan, go, B, crop, img_scratch, grad_s=garg, workspace=ws)
            else:
                ops.collect(plan, go, B, grad_argument=garg, workspace=ws)
            ops.image_bwd_update(p, s, garg, eps, 0.7, grad, mask, m, v, opt, stats, img_scratch, upd_scratch,
                                 accumulate=bool(accumulate))
        torch.cuda.synchronize()
        return p, m, v, grad, float(stats[L.STAT_GRAD_NORM])

    a, b = run(True), run(False)
    for k, (x, y) in enumerate(zip(a[:4], b[:4])):
        assert torch.equal(x, y), (k, float((x - y).abs().max()))
    assert a[4] == pytest.approx(b[4], rel=1e-6)
    if not apply:
        assert torch.equal(a[0], p0) and torch.equal(a[1], m0) and torch.equal(a[2], v0)


def test_every_family_is_offered_the_one_launch_form_at_full_size():
    """Every family, with the image as it stands or through a composing window, takes the one-launch form."""
    from adversarialvlm_amd import ops
    for maker, H, W, crop in CASES:
        plan = _plan(maker, H, W)
        composes = crop is None or ops.crop_composes(plan, H, W, crop)      # a window that does not compose is resized on its own
        assert ops.collect_update_supported(plan, H, W, crop) == composes, (maker, H, W, crop)


@pytest.mark.parametrize("maker,H,W,crop", [
    ("llava", 100, 120, None),                        # up-sampling: a pixel of the image reaches more than six canvas rows
    ("qwen2vl", 512, 512, (16, 24, 470, 450)),        # a window that does not compose with the plan's stage 0
    ("phi3", 600, 520, (10, 20, 560, 480)),
    ("llava", 24, 30, None),                          # fewer than 1000 positions
])
def test_collect_update_refuses_what_it_does_not_cover_and_launches_nothing(maker, H, W, crop):
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    plan = _plan(maker, H, W)
    assert not ops.collect_update_supported(plan, H, W, crop)
    gen = torch.Generator().manual_seed(1)
    p = torch.randn(3, H, W, generator=gen).to(DEV)
    keep = p.clone()
    z = torch.zeros_like(p)
    grad, m, v = z.clone(), z.clone(), z.clone()
    go = torch.randn(1, plan.out_numel, generator=gen).to(DEV)
    stats = torch.zeros(L.STATS_N, dtype=torch.float32, device=DEV)
    args = (plan, go, 1, p, z, 0.3, 0.7, grad, torch.ones_like(p), m, v, _opt(L, L.OPT_ADAMW, 0), stats,
            ops.image_scratch(H, W, 0, DEV), ops.update_scratch(p.numel(), DEV))
    assert ops.collect_update(*args, crop=crop, if_supported=True) is None
    with pytest.raises(L.AdvxError):
        ops.collect_update(*args, crop=crop)
    torch.cuda.synchronize()
    assert torch.equal(p, keep) and not bool(grad.any()) and not bool(m.any())


def test_the_switch_turns_the_offer_off():
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    plan = _plan("llava", 512, 512)
    lib = L.load()
    try:
        L.check(lib.advx_set_tuning(L.TUNE_COLLECT_UPDATE, 0), "advx_set_tuning")
        assert not ops.collect_update_supported(plan, 512, 512)
    finally:
        L.check(lib.advx_set_tuning(0, 0), "advx_set_tuning")
    assert ops.collect_update_supported(plan, 512, 512)
