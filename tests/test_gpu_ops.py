"""GPU tier: every single op of the C ABI against the CPU oracle on the same seeded inputs.
Tolerance: fp32 within 1e-4 relative (north star); in practice ~1e-6."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import pixel_ops as P

pytestmark = pytest.mark.gpu
TOL = 1e-4      # north-star bound
TIGHT = 5e-6    # what the kernels are expected to reach


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from adversarialvlm_amd import ops as o
    return o


def test_tanh_fwd_bwd(ops, dev):
    torch.manual_seed(0)
    p = (torch.randn(3, 37, 53) * 2).requires_grad_(True)
    x = P.tanh_reparam(p, 0.5)
    g = torch.randn_like(x)
    x.backward(g)
    assert rel_err(ops.tanh_fwd(p.detach().to(dev), 0.5).cpu(), x.detach()) < TIGHT
    assert rel_err(ops.tanh_bwd(p.detach().to(dev), g.to(dev), 0.5).cpu(), p.grad) < TIGHT


def test_tanh_golden(ops, dev):
    g = load_golden("closed_form.npz")
    x = ops.tanh_fwd(torch.tensor(g["fit_p"]).to(dev), float(g["fit_eps"]))
    assert rel_err(x.cpu(), g["fit_x"]) < TIGHT


@pytest.mark.parametrize("H,W,k,sigma", [(40, 56, 5, 7.0), (33, 70, 9, 10.0), (64, 64, 5, 0.37), (20, 20, 31, 4.0),
                                          (336, 336, 9, 10.0), (8, 100, 3, 1.5)])
def test_blur_fwd_bwd(ops, dev, H, W, k, sigma):
    torch.manual_seed(1)
    x = torch.randn(3, H, W, requires_grad=True)
    y = P.gaussian_blur(x, k, sigma)
    g = torch.randn_like(y)
    y.backward(g)
    assert rel_err(ops.blur_fwd(x.detach().to(dev), k, sigma).cpu(), y.detach()) < TIGHT
    assert rel_err(ops.blur_bwd(g.to(dev), k, sigma).cpu(), x.grad) < TIGHT


def test_blur_rejects_bad_kernel(ops, dev):
    from adversarialvlm_amd._lib import AdvxError
    x = torch.zeros(3, 8, 8, device=dev)
    with pytest.raises(AdvxError):
        ops.blur_fwd(x, 4, 1.0)       # even
    with pytest.raises(AdvxError):
        ops.blur_fwd(x, 33, 1.0)      # too large
    with pytest.raises(AdvxError):
        ops.blur_fwd(torch.zeros(3, 3, 3, device=dev), 9, 1.0)   # radius does not fit reflect pad


@pytest.mark.parametrize("H,W,crop", [(64, 80, (3, 5, 40, 60)), (336, 336, (20, 11, 280, 300)), (50, 50, (0, 0, 50, 50)),
                                        (48, 64, (10, 20, 12, 9))])
def test_crop_resize_fwd_bwd(ops, dev, H, W, crop):
    torch.manual_seed(2)
    s = torch.rand(3, H, W, requires_grad=True)
    y = P.resized_crop(s, *crop, (H, W))
    g = torch.randn_like(y)
    y.backward(g)
    assert rel_err(ops.crop_resize_fwd(s.detach().to(dev), crop).cpu(), y.detach()) < TIGHT
    assert rel_err(ops.crop_resize_bwd(g.to(dev), crop).cpu(), s.grad) < TIGHT


def test_crop_rejects_window_outside(ops, dev):
    from adversarialvlm_amd._lib import AdvxError
    with pytest.raises(AdvxError):
        ops.crop_resize_fwd(torch.zeros(3, 32, 32, device=dev), (10, 10, 30, 30))


@pytest.mark.parametrize("B,n", [(1, 1024), (4, 4096), (7, 1000), (64, 3 * 64 * 64), (5, 10101), (3, 2)])
def test_batch_reduce(ops, dev, B, n):
    torch.manual_seed(3)
    g = torch.randn(B, n)
    out = ops.batch_reduce(g.to(dev)).cpu()
    assert rel_err(out, g.double().sum(0)) < TIGHT


def test_batch_reduce_is_deterministic(ops, dev):
    g = torch.randn(64, 3 * 112 * 112, device=dev)
    a = ops.batch_reduce(g)
    b = ops.batch_reduce(g)
    assert torch.equal(a, b)


def test_philox_normal_statistics(ops, dev):
    from scipy import stats as ss
    n = 1 << 20
    z = ops.philox_normal(n, seed=1234, offset=5, device=dev).cpu().numpy()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3
    assert ss.kstest(z[:200000], "norm").pvalue > 1e-3
    assert abs(ss.skew(z)) < 1e-2 and abs(ss.kurtosis(z)) < 2e-2
    z2 = ops.philox_normal(n, seed=1234, offset=5, device=dev).cpu().numpy()
    assert np.array_equal(z, z2)                       # counter based: reproducible
    z3 = ops.philox_normal(n, seed=1234, offset=6, device=dev).cpu().numpy()
    assert abs(np.corrcoef(z, z3)[0, 1]) < 5e-3        # different offset: independent stream


@pytest.mark.parametrize("n", [4096, 1001])
def test_philox_normal_matches_oracle(ops, dev, n):
    """The device generator is Philox4x32-10 (oracle/philox.py, pinned by the Random123 known answers)
    with counter (q, 0, offset) and Box-Muller on the hardware log2/sqrt/sin/cos: compared at their
    tolerance, far below anything a different counter, key or round count could pass."""
    from oracle import philox
    seed, offset = 0x1234567890ABCDEF, (7 << 32) | 9
    z = ops.philox_normal(n, seed=seed, offset=offset, device=dev).cpu().double().numpy()
    ref = philox.unit_noise(1, n, seed, offset)[0]
    assert np.abs(z - ref).max() < 2e-5


def test_adamw_trajectory_golden(ops, dev):
    """torch.optim.AdamW + StepLR 6-step trajectory captured in closed_form.npz."""
    from adversarialvlm_amd import _lib as L
    g = load_golden("closed_form.npz")
    G = torch.tensor(g["adamw_g"])
    n = G[0].numel()
    p = torch.zeros(n, device=dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    mask = torch.ones(n, device=dev)
    stats = torch.zeros(L.STATS_N, device=dev)
    scratch = ops.update_scratch(n, dev)
    lr, step_size, gamma = float(g["adamw_lr0"]), int(g["adamw_step_size"]), float(g["adamw_gamma"])
    for t in range(G.shape[0]):
        assert lr == pytest.approx(float(g["adamw_lr"][t]), rel=1e-12)
        o = L.OptScalars()
        o.kind, o.apply = L.OPT_ADAMW, 1
        b1, b2 = 0.9, 0.999
        o.lr, o.decay, o.w1, o.beta2, o.w2 = lr, 1 - lr * 0.01, 1 - b1, b2, 1 - b2
        o.bias2_sqrt = (1 - b2 ** (t + 1)) ** 0.5
        o.eps = 1e-8
        o.neg_step_size = -(lr / (1 - b1 ** (t + 1)))
        grad = G[t].flatten().to(dev).clone()
        ops.update(p, m, v, grad, mask, o, stats, scratch)
        assert rel_err(p.cpu(), g["adamw_p"][t].flatten()) < TIGHT
        assert rel_err(m.cpu(), g["adamw_m"][t].flatten()) < TIGHT
        assert rel_err(v.cpu(), g["adamw_v"][t].flatten()) < TIGHT
        assert float(stats[L.STAT_GRAD_NORM]) == pytest.approx(float(G[t].norm()), rel=1e-5)
        if (t + 1) % step_size == 0:
            lr = lr * gamma


def test_image_fwd_stats_and_quantiser(ops, dev):
    """image_fit_loss, quantise-error statistics and x mean/std against the oracle,
    including the 256 lattice points that must survive the truncating quantiser."""
    from adversarialvlm_amd import _lib as L
    g = load_golden("closed_form.npz")
    s_target = torch.tensor(g["q_s"])            # contains k/255 exactly
    # choose p = 0 so that s = x0: the statistics must reproduce the fixture
    x0 = s_target.clone()
    p = torch.zeros_like(x0)
    stats = torch.zeros(L.STATS_N, device=dev)
    stats[L.STAT_QERR_STD] = 0.25
    scratch = ops.image_scratch(16, 16, 0, dev)
    s, _ = ops.image_fwd(p.to(dev), x0.to(dev), 0.5, stats, scratch)
    assert torch.equal(s.cpu(), x0)
    st = stats.cpu()
    assert float(st[L.STAT_SIGMA]) == 0.25                     # rotated from the previous QERR_STD
    assert float(st[L.STAT_QERR_STD]) == pytest.approx(float(g["q_std"]), rel=1e-5)
    assert float(st[L.STAT_QERR_MEAN]) == pytest.approx(float(g["q_mean"]), rel=1e-5)
    assert float(st[L.STAT_QERR_L1]) == pytest.approx(float(g["q_l1"]), rel=1e-5)
    assert float(st[L.STAT_IMGFIT]) == pytest.approx(float(P.image_fit_loss(x0, torch.zeros_like(x0))), rel=1e-5)


def test_image_fit_grad_golden(ops, dev):
    from adversarialvlm_amd import _lib as L
    g = load_golden("closed_form.npz")
    p = torch.tensor(g["fit_p"]).to(dev)
    x0 = torch.tensor(g["fit_x0"]).to(dev)
    _, H, W = p.shape
    stats = torch.zeros(L.STATS_N, device=dev)
    scratch = ops.image_scratch(H, W, 0, dev)
    s, _ = ops.image_fwd(p, x0, float(g["fit_eps"]), stats, scratch)
    assert float(stats[L.STAT_IMGFIT]) == pytest.approx(float(g["fit_loss"]), rel=1e-5)
    grad = torch.empty_like(p)
    ops.image_bwd(p, s, torch.zeros_like(p), float(g["fit_eps"]), 1.0, grad, scratch)
    assert rel_err(grad.cpu(), g["fit_p_grad"]) < TIGHT


def test_quantise_is_uint8_truncation():
    """advx_quantise = tensor2pil -> PNG -> pil_to_tensor (attack_model.py:368-371), bit-exact,
    including the 256 lattice points and values outside [0, 1]."""
    from adversarialvlm_amd import ops
    from oracle import pixel_ops as P
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    s = torch.cat([torch.rand(3 * 40 * 40, generator=gen) * 1.4 - 0.2, torch.arange(256, dtype=torch.float32) / 255,
                   torch.tensor([-1.0, 0.0, 1.0, 2.0, 0.999999, 1e-9])])
    got = ops.quantise(s.to(dev)).cpu()
    assert torch.equal(got, P.quantise(s))


@pytest.mark.parametrize("blur,crop", [(None, None), ((5, 1.7), None), (None, (6, 9, 50, 61)), ((9, 0.8), (6, 9, 50, 61))])
@pytest.mark.parametrize("kind,apply,accumulate", [("adamw", 1, False), ("adamw", 0, True), ("sign", 1, True)])
def test_image_bwd_update_is_image_bwd_then_update(ops, dev, blur, crop, kind, apply, accumulate):
    """advx_image_bwd_update (tanh backward - and the crop's transposed resize - inside the optimiser's
    launch) against advx_image_bwd followed by advx_update: identical bits in grad, p, m, v, ||g||."""
    from adversarialvlm_amd import _lib as L
    H, W = 64, 80
    gen = torch.Generator().manual_seed(17)
    x0 = torch.rand(3, H, W, generator=gen).to(dev)
    mask = (torch.rand(3, H, W, generator=gen) > 0.3).float().to(dev)
    garg = (torch.randn(3, H, W, generator=gen) * 0.05).to(dev)
    o = L.OptScalars()
    o.kind, o.apply = (L.OPT_ADAMW if kind == "adamw" else L.OPT_SIGN), apply
    o.lr, o.decay, o.w1, o.beta2, o.w2 = 1e-2, 1 - 1e-4, 0.1, 0.999, 0.001
    o.bias2_sqrt, o.eps, o.neg_step_size = (1 - 0.999 ** 3) ** 0.5, 1e-8, -(1e-2 / (1 - 0.9 ** 3))
    runs = []
    for fused in (False, True):
        g2 = torch.Generator().manual_seed(18)
        p = (torch.randn(3, H, W, generator=g2) * 0.3).to(dev)
        m = (torch.randn(3, H, W, generator=g2) * 0.01).to(dev)
        v = (torch.rand(3, H, W, generator=g2) * 1e-4).to(dev)
        grad = (torch.randn(3, H, W, generator=g2) * 0.01).to(dev)
        stats = torch.zeros(L.STATS_N, device=dev)
        iscr = ops.image_scratch(H, W, blur[0] if blur else 0, dev)
        uscr = ops.update_scratch(p.numel(), dev)
        s, arg = ops.image_fwd(p, x0, 0.5, stats, iscr, blur=blur, crop=crop,
                               argument=torch.empty_like(x0) if crop is not None else None)
        if fused:
            ops.image_bwd_update(p, s, garg, 0.5, 0.7, grad, mask, m, v, o, stats, iscr, uscr, blur=blur, crop=crop,
                                 accumulate=accumulate)
        else:
            ops.image_bwd(p, s, garg, 0.5, 0.7, grad, iscr, blur=blur, crop=crop, accumulate=accumulate)
            ops.update(p, m, v, grad, mask, o, stats, uscr)
        runs.append((grad.clone(), p.clone(), m.clone(), v.clone(), stats.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    assert float(runs[0][4][L.STAT_GRAD_NORM]) > 0
