"""GPU tier: advx_emit / advx_collect (the four differentiable processors) against the CPU
oracle and against the fixtures captured from the imported reference classes."""
import numpy as np
import pytest
import torch

from conftest import load_golden, lcg_tensor, rel_err
from oracle.processors import LlavaOracle, MllamaOracle, Phi3Oracle, Qwen2VLOracle

pytestmark = pytest.mark.gpu
TIGHT = 5e-6


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _run(plan, img, up, dev):
    from adversarialvlm_amd import ops
    x = img.detach().to(dev).requires_grad_(True)
    pv = ops.ProcessFunction.apply(x, plan)
    pv.backward(up.to(dev).view(pv.shape))
    return pv.detach().cpu(), x.grad.cpu()


@pytest.mark.parametrize("name", ["down", "mixed", "ident", "up"])
def test_llava_reference_capture(dev, name):
    from adversarialvlm_amd.plan import Plan
    g = load_golden("llava_reference.npz")
    ch, cw = (int(v) for v in g[f"{name}_crop"])
    img = torch.tensor(g[f"{name}_image"])
    plan = Plan.llava(img.shape[1], img.shape[2], ch, cw)
    pv, grad = _run(plan, img, lcg_tensor((1, 3, ch, cw), int(g[f"{name}_salt"])), dev)
    assert tuple(pv.shape) == g[f"{name}_pixel_values"].shape
    assert rel_err(pv, g[f"{name}_pixel_values"]) < TIGHT
    assert rel_err(grad, g[f"{name}_image_grad"]) < TIGHT
    if name == "ident":   # identity resize is a bit-exact passthrough before normalisation
        ref = LlavaOracle(ch, cw).process(img)["pixel_values"]
        assert torch.equal(pv, ref)


def test_llava_full_size_512_to_336(dev):
    from adversarialvlm_amd.plan import Plan
    g = load_golden("llava_reference.npz")
    img = lcg_tensor((3, 512, 512), int(g["full_salt_image"])) + 0.5
    plan = Plan.llava(512, 512)
    pv, grad = _run(plan, img, lcg_tensor((1, 3, 336, 336), int(g["full_salt_up"])), dev)
    assert abs(float(pv.double().sum()) - float(g["full_pv_sum"])) < 1e-5 * abs(float(g["full_pv_sum"]))
    assert rel_err(pv.flatten()[g["full_pv_idx"]], g["full_pv_val"]) < TIGHT
    assert rel_err(grad.flatten()[g["full_grad_idx"]], g["full_grad_val"]) < TIGHT


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_qwen_reference_capture(dev, name):
    from adversarialvlm_amd.plan import Plan
    g = load_golden("qwen2vl_reference.npz")
    minp, maxp = (int(v) for v in g[f"{name}_minmax"])
    img = torch.tensor(g[f"{name}_image"])
    plan = Plan.qwen2vl(img.shape[1], img.shape[2], min_pixels=minp, max_pixels=maxp)
    assert plan.info.num_tiles == int(g[f"{name}_num_tiles"][0])
    assert (plan.stage(0).res_h, plan.stage(0).res_w) == tuple(int(v) for v in g[f"{name}_optimal_size"])
    ref = g[f"{name}_pixel_values"]
    pv, grad = _run(plan, img, lcg_tensor(ref.shape, int(g[f"{name}_salt"])), dev)
    assert tuple(pv.shape) == ref.shape
    assert rel_err(pv, ref) < TIGHT
    assert rel_err(grad, g[f"{name}_image_grad"]) < TIGHT


@pytest.mark.parametrize("name", ["wide", "tall", "square"])
def test_phi3_reference_capture(dev, name):
    from adversarialvlm_amd.plan import Plan
    g = load_golden("phi3_reference.npz")
    img = torch.tensor(g[f"{name}_image"])
    plan = Plan.phi3(img.shape[1], img.shape[2])
    assert [[plan.info.image_h, plan.info.image_w]] == g[f"{name}_image_sizes"].tolist()
    assert [plan.info.num_img_tokens] == g[f"{name}_num_img_tokens"].tolist()
    pv, grad = _run(plan, img, lcg_tensor((1, 7, 3, 336, 336), int(g[f"{name}_salt"])), dev)
    flat = pv.reshape(7, -1).double()
    np.testing.assert_allclose(flat.sum(1).numpy(), g[f"{name}_tile_sum"], rtol=2e-6, atol=2e-3)
    np.testing.assert_allclose((flat ** 2).sum(1).numpy(), g[f"{name}_tile_sumsq"], rtol=2e-6, atol=2e-3)
    assert rel_err(pv.flatten()[g[f"{name}_pv_idx"]], g[f"{name}_pv_val"]) < TIGHT
    assert rel_err(grad, g[f"{name}_image_grad"]) < 2e-5
    n_real = plan.info.num_tiles
    assert torch.count_nonzero(pv[0, n_real:]) == 0            # padding tiles are exact zeros


@pytest.mark.parametrize("name", ["one", "wide3", "tall4", "two_by_two", "wide2", "tall2", "up"])
def test_mllama_reference_capture(dev, name):
    """HIP path against captures from the reference's own DifferentiableMllamaImageProcessor."""
    from adversarialvlm_amd.plan import Plan
    g = load_golden("mllama_reference.npz")
    img = torch.tensor(g[f"{name}_image"])
    plan = Plan.mllama(img.shape[1], img.shape[2], tile=int(g[f"{name}_tile"]), max_tiles=4)
    assert plan.info.num_tiles == int(g[f"{name}_num_tiles"])
    ref = g[f"{name}_pixel_values"]
    pv, grad = _run(plan, img, lcg_tensor(ref.shape, int(g[f"{name}_salt"])), dev)
    assert rel_err(pv, ref) < TIGHT
    assert rel_err(grad, g[f"{name}_image_grad"]) < TIGHT
    assert torch.count_nonzero(pv[0, 0, plan.info.num_tiles:]) == 0


def test_mllama_full_size_512_to_560_tiles(dev):
    from adversarialvlm_amd.plan import Plan
    g = load_golden("mllama_reference.npz")
    img = lcg_tensor((3, 512, 512), int(g["full_salt_image"])) + 0.5
    plan = Plan.mllama(512, 512)
    shape = tuple(int(v) for v in g["full_shape"])
    assert plan.info.num_tiles == int(g["full_num_tiles"])
    pv, grad = _run(plan, img, lcg_tensor(shape, int(g["full_salt_up"])), dev)
    assert abs(float(pv.double().sum()) - float(g["full_pv_sum"])) < 1e-5 * abs(float(g["full_pv_sum"]))
    assert rel_err(pv.flatten()[g["full_pv_idx"]], g["full_pv_val"]) < TIGHT
    assert rel_err(grad.flatten()[g["full_grad_idx"]], g["full_grad_val"]) < TIGHT


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_mllama_restated_fixture(dev, name):
    from adversarialvlm_amd.plan import Plan
    g = load_golden("mllama_restated.npz")
    img = torch.tensor(g[f"{name}_image"])
    tile = int(g[f"{name}_tile"])
    plan = Plan.mllama(img.shape[1], img.shape[2], tile=tile, max_tiles=4)
    assert plan.info.num_tiles == int(g[f"{name}_num_tiles"])
    ref = g[f"{name}_pixel_values"]
    pv, grad = _run(plan, img, lcg_tensor(ref.shape, int(g[f"{name}_salt"])), dev)
    assert rel_err(pv, ref) < TIGHT
    assert rel_err(grad, g[f"{name}_image_grad"]) < TIGHT
    assert torch.count_nonzero(pv[0, 0, plan.info.num_tiles:]) == 0


@pytest.mark.parametrize("kind,H,W", [("llava", 336, 336), ("llava", 512, 512), ("mllama", 336, 336), ("mllama", 700, 420),
                                       ("phi3", 336, 336), ("phi3", 300, 500), ("qwen", 336, 336), ("qwen", 512, 512)])
def test_full_size_against_oracle(dev, kind, H, W):
    """BASELINE-size geometries live against the oracle (torch CPU), forward and backward."""
    from adversarialvlm_amd.plan import Plan
    torch.manual_seed(5)
    img = torch.rand(3, H, W)
    plan, ora = {"llava": (Plan.llava, LlavaOracle), "mllama": (Plan.mllama, MllamaOracle),
                 "phi3": (Plan.phi3, Phi3Oracle), "qwen": (Plan.qwen2vl, Qwen2VLOracle)}[kind]
    plan = plan(H, W)
    x = img.clone().requires_grad_(True)
    ref = ora().process(x)["pixel_values"]
    up = torch.randn(ref.shape)
    ref.backward(up)
    pv, grad = _run(plan, img, up, dev)
    assert tuple(pv.shape) == tuple(ref.shape)
    assert rel_err(pv, ref.detach()) < TIGHT
    assert rel_err(grad, x.grad) < 2e-5


def test_emit_broadcast_noise_and_collect_sum(dev):
    """repeat(B) + randn*sigma (attack_model.py:316-321) with the noise supplied, and the
    batch-sum backward, on the Mllama layout (tiles + zero tiles)."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    torch.manual_seed(6)
    B, H, W, tile = 5, 50, 90, 32
    img = torch.rand(3, H, W, requires_grad=True)
    z = torch.randn(B, 1, 4, 3, tile, tile)
    sigma = 0.0123
    ref1 = MllamaOracle(tile=tile).process(img)["pixel_values"]
    ref = ref1.repeat(B, 1, 1, 1, 1, 1) + z * sigma
    up = torch.randn_like(ref)
    ref.backward(up)
    plan = Plan.mllama(H, W, tile=tile)
    sig = torch.tensor([sigma], device=dev)
    out = ops.emit(plan, img.detach().to(dev), B, sigma_dev=sig, unit_noise=z.to(dev))
    assert rel_err(out.cpu().view(ref.shape), ref.detach()) < TIGHT
    g = ops.collect(plan, up.to(dev).view(B, -1), B)
    assert rel_err(g.cpu(), img.grad) < TIGHT
    # accumulate flag adds instead of overwriting
    g2 = ops.collect(plan, up.to(dev).view(B, -1), B, grad_argument=g.clone(), accumulate=True)
    assert rel_err(g2.cpu(), 2 * img.grad) < TIGHT


def test_emit_philox_noise_statistics(dev):
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    plan = Plan.llava(64, 64, 64, 64)
    img = torch.rand(3, 64, 64, device=dev)
    sig = torch.tensor([0.05], device=dev)
    clean = ops.emit(plan, img, 16)
    noisy = ops.emit(plan, img, 16, sigma_dev=sig, philox=(7, 3))
    d = (noisy - clean).cpu().numpy() / 0.05
    assert abs(d.mean()) < 1e-2 and abs(d.std() - 1.0) < 1e-2
    assert abs(np.corrcoef(d[0], d[1])[0, 1]) < 3e-2        # batch rows get independent draws
    again = ops.emit(plan, img, 16, sigma_dev=sig, philox=(7, 3))
    assert torch.equal(noisy, again)                          # counter-based: reproducible


@pytest.mark.parametrize("kind", ["llava", "qwen2vl", "mllama"])
def test_emit_philox_noise_matches_oracle(dev, kind):
    """Element i of batch row b gets lane i%4 of Philox block (i//4, b, offset) -- in every layout, the
    flat index of the emitted sample addresses the stream (oracle/philox.py::unit_noise)."""
    from oracle import philox
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    plan = {"llava": lambda: Plan.llava(48, 40, 32, 32),
            "qwen2vl": lambda: Plan.qwen2vl(60, 90, min_pixels=28 * 28 * 4, max_pixels=28 * 28 * 64),
            "mllama": lambda: Plan.mllama(64, 64, tile=32)}[kind]()
    img = torch.rand(3, plan.in_h, plan.in_w, device=dev)
    B, seed, offset = 5, 99, (1 << 33) + 4
    sig = torch.tensor([0.25], device=dev)
    clean = ops.emit(plan, img, B).double()
    noisy = ops.emit(plan, img, B, sigma_dev=sig, philox=(seed, offset)).double()
    z = ((noisy - clean) / 0.25).cpu().numpy().reshape(B, -1)
    ref = philox.unit_noise(B, plan.out_numel, seed, offset)
    # fp32 rounding of clean + 0.25 z (values of a few units) on top of the generator's own tolerance
    assert np.abs(z - ref).max() < 1e-5 + 4 * 2.0 ** -22 * float(clean.abs().max() + 2)


def test_width_one_resize_follows_the_gpu_kernel(dev):
    """594x24 into one 16x16 Mllama tile resizes to 16x1.  torch's GPU antialias kernel (what the
    reference runs), the tap tables and the HIP path agree; ATen's CPU kernel does not for a width-1
    image, which is why oracle/processors.py::_aa widens the column (found by tools/fuzz_parity.py)."""
    import torch.nn.functional as F
    from oracle.processors import _aa
    from adversarialvlm_amd.plan import Plan
    img = torch.rand(3, 594, 24, generator=torch.Generator().manual_seed(0))
    on_gpu = F.interpolate(img.to(dev)[None], size=[16, 1], mode="bilinear", align_corners=False, antialias=True)[0].cpu()
    assert rel_err(_aa(img, 16, 1), on_gpu) < TIGHT
    plan = Plan.mllama(594, 24, tile=16, max_tiles=1)
    x = img.clone().requires_grad_(True)
    ref = MllamaOracle(tile=16, max_tiles=1).process(x)["pixel_values"]
    up = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    ref.backward(up)
    pv, grad = _run(plan, img, up, dev)
    assert rel_err(pv, ref.detach()) < TIGHT and rel_err(grad, x.grad) < 2e-5


def test_random_geometries(dev):
    """Seeded subset of tools/fuzz_parity.py: extreme aspect ratios, images below one tile, odd sizes
    and processor parameters, forward + backward + integer metadata against the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    rng = np.random.default_rng(2024)
    bad = []
    for k in range(60):
        kind, H, W, args = fuzz_parity.draw_case(rng)
        verdict, ef, eb = fuzz_parity.run_case(kind, H, W, args, dev, 7000 + k)
        if verdict not in ("ok", "both-reject"):
            bad.append((kind, H, W, args, verdict, ef, eb))
    assert not bad, bad


@pytest.mark.parametrize("H,W,half", [(60, 90, False), (112, 112, False), (45, 31, False), (112, 112, True)])
def test_multi_plan_calls_equal_the_single_plan_calls(dev, H, W, half):
    """advx_emit_multi / advx_collect_multi (cross-model runs: the plans' image resizes in one launch,
    their image gradients summed in one kernel) against n advx_emit_ex / accumulating advx_collect
    calls: identical bits, with noise, per-plan boundary dtypes and padding kept zero."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd._lib import AdvxError
    from adversarialvlm_amd.plan import Plan
    mk = lambda: [Plan.phi3(H, W), Plan.qwen2vl(H, W, min_pixels=28 * 28 * 4, max_pixels=28 * 28 * 64),
                  Plan.mllama(H, W, tile=32), Plan.llava(H, W, 48, 48)]
    single, multi = mk(), mk()
    if half:
        for group in (single, multi):
            group[0].set_io(torch.float16)
            group[1].set_io(torch.bfloat16)
    gen = torch.Generator().manual_seed(12)
    img = torch.rand(3, H, W, generator=gen).to(dev)
    sig = torch.tensor([0.05], device=dev)
    batches = [2, 3, 1, 4]
    offs = [40, 41, 42, 43]
    for keep in (False, True):
        bufs_a = [torch.zeros(B, pl.out_numel, dtype=ops._plan_dtype(pl), device=dev) for pl, B in zip(single, batches)] if keep else [None] * 4
        bufs_b = [b.clone() for b in bufs_a] if keep else None
        ws_a = [torch.empty(pl.workspace_floats, device=dev) for pl in single]
        ws_b = [torch.empty(pl.workspace_floats, device=dev) for pl in multi]
        ref = [ops.emit(pl, img, B, sigma_dev=sig, philox=(5, o), workspace=w, out=b, keep_padding=keep)
               for pl, B, o, w, b in zip(single, batches, offs, ws_a, bufs_a)]
        got = ops.emit_multi(multi, img, batches, sigma_dev=sig, philox=(5, offs), workspaces=ws_b, outs=bufs_b, keep_padding=keep)
        for a, b in zip(ref, got):
            assert a.dtype == b.dtype and torch.equal(a, b)
        ups = [(torch.randn(o.shape, generator=gen) * 0.1).to(dev).to(o.dtype) for o in ref]
        g_ref = torch.empty(3, H, W, device=dev)
        for i, (pl, u, B, w) in enumerate(zip(single, ups, batches, ws_a)):
            ops.collect(pl, u, B, grad_argument=g_ref, accumulate=(i > 0), workspace=w)
        g_got = ops.collect_multi(multi, ups, batches, workspaces=ws_b)
        assert torch.equal(g_ref, g_got)
        # accumulate: adds onto what is there
        g2 = ops.collect_multi(multi, ups, batches, grad_argument=g_got.clone(), accumulate=True, workspaces=ws_b)
        assert rel_err(g2.cpu(), 2 * g_ref.cpu()) < TIGHT
    # one plan is the plain call; argument checks
    one = ops.emit_multi(multi[:1], img, [2], workspaces=ws_b[:1])
    assert torch.equal(one[0], ops.emit(single[0], img, 2, workspace=ws_a[0]))
    with pytest.raises(AdvxError):
        ops.emit_multi(multi[:2], img, [1, 1], workspaces=[ws_b[0], ws_b[0]])            # shared workspace
    with pytest.raises(AdvxError):
        ops.emit_multi(multi + multi[:1], img, batches + [1], workspaces=ws_b + ws_a[:1])   # more than four plans
    with pytest.raises(AdvxError):
        ops.emit_multi([multi[0], Plan.llava(H + 1, W, 48, 48)], img, [1, 1])              # different images


def test_layout_index_map_matches_device(dev):
    """advx_plan_out_index (host) vs what the device wrote: integer layout bit-exact."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    for plan in (Plan.qwen2vl(60, 90, min_pixels=28 * 28 * 4, max_pixels=28 * 28 * 64), Plan.mllama(90, 50, tile=32)):
        st = plan.stage(0)
        # an image whose canvas values are all distinct: use mean 0 / std 1 via a fresh plan
        p2 = type(plan)(plan.kind, plan.in_h, plan.in_w, (plan.desc.a0, plan.desc.a1, plan.desc.a2, plan.desc.a3, plan.desc.a4),
                        mean=(0, 0, 0), std=(1, 1, 1))
        img = torch.rand(3, plan.in_h, plan.in_w, device=dev)
        out = ops.emit(p2, img, 1).cpu().flatten()
        ws = torch.empty(p2.workspace_floats, device=dev)
        ops.emit(p2, img, 1, workspace=ws)
        canvas = ws[:3 * st.can_h * st.can_w].cpu().view(3, st.can_h, st.can_w)
        rng = np.random.default_rng(0)
        for _ in range(300):
            c, y, x = int(rng.integers(3)), int(rng.integers(st.can_h)), int(rng.integers(st.can_w))
            for idx in p2.out_index(0, c, y, x):
                assert out[idx].item() == canvas[c, y, x].item()


def test_odd_sizes_take_the_scalar_paths(dev):
    """P_out not a multiple of 4: rows are not 16-byte aligned, so emit / batch-reduce must fall
    back to scalar stores and columns (and still match the oracle, noise included)."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    torch.manual_seed(8)
    H, W, ch, cw, B = 37, 91, 25, 31, 3
    plan = Plan.llava(H, W, ch, cw)
    assert plan.out_numel % 4 != 0
    img = torch.rand(3, H, W, requires_grad=True)
    z = torch.randn(B, 3, ch, cw)
    ref = LlavaOracle(ch, cw).process(img)["pixel_values"].repeat(B, 1, 1, 1) + z * 0.02
    up = torch.randn_like(ref)
    ref.backward(up)
    sig = torch.tensor([0.02], device=dev)
    out = ops.emit(plan, img.detach().to(dev), B, sigma_dev=sig, unit_noise=z.to(dev))
    assert rel_err(out.cpu().view(ref.shape), ref.detach()) < TIGHT
    g = ops.collect(plan, up.to(dev).view(B, -1), B)
    assert rel_err(g.cpu(), img.grad) < TIGHT
    noisy = ops.emit(plan, img.detach().to(dev), B, sigma_dev=sig, philox=(1, 1))
    d = ((noisy - ops.emit(plan, img.detach().to(dev), B)) / 0.02).cpu()
    assert abs(float(d.mean())) < 0.05 and abs(float(d.std()) - 1) < 0.05


def test_batch_of_one_and_large_batch(dev):
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    plan = Plan.llava(48, 48, 48, 48)
    img = torch.rand(3, 48, 48, device=dev)
    one = ops.emit(plan, img, 1)
    many = ops.emit(plan, img, 130)                       # more rows than batch slices
    assert torch.equal(many, one.expand(130, -1))
    g = torch.randn(130, plan.out_numel, device=dev)
    assert rel_err(ops.collect(plan, g, 130).cpu(), ops.collect(plan, g.double().sum(0, keepdim=True).float(), 1).cpu()) < 1e-5


@pytest.mark.parametrize("blur,crop", [(None, None), ((5, 1.3), None), (None, (4, 7, 40, 61)), ((9, 0.7), (4, 7, 40, 61))])
def test_forward_multi_is_image_fwd_then_emit_multi(dev, blur, crop):
    """advx_forward_multi against advx_image_fwd + advx_emit_multi: same pixel_values, image, argument and
    statistics (the reduction of the statistics rides in the plans' resize launch), one and three plans."""
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    H, W = 60, 90
    gen = torch.Generator().manual_seed(31)
    x0 = torch.rand(3, H, W, generator=gen).to(dev)
    p = (torch.randn(3, H, W, generator=gen) * 0.4).to(dev)
    for n in (1, 3):
        mk = lambda: [Plan.phi3(H, W), Plan.qwen2vl(H, W, min_pixels=28 * 28 * 4, max_pixels=28 * 28 * 64),
                      Plan.mllama(H, W, tile=32)][:n]
        batches = [2, 3, 1][:n]
        runs = []
        for one_call in (False, True):
            plans = mk()
            stats = torch.zeros(L.STATS_N, device=dev)
            stats[L.STAT_QERR_STD] = 0.02
            scr = ops.image_scratch(H, W, blur[0] if blur else 0, dev)
            ws = [torch.empty(pl.workspace_floats, device=dev) for pl in plans]
            s = torch.empty_like(x0)
            arg_buf = torch.empty_like(x0) if crop is not None else None
            if one_call:
                outs, arg = ops.forward_multi(p, x0, 0.5, stats, scr, plans, batches, s, argument=arg_buf, blur=blur, crop=crop,
                                              philox=(9, [70, 71, 72][:n]), workspaces=ws)
            else:
                _, arg = ops.image_fwd(p, x0, 0.5, stats, scr, blur=blur, crop=crop, s=s, argument=arg_buf)
                outs = ops.emit_multi(plans, arg, batches, sigma_dev=stats[L.STAT_SIGMA:L.STAT_SIGMA + 1],
                                      philox=(9, [70, 71, 72][:n]), workspaces=ws)
            runs.append([o.clone() for o in outs] + [s.clone(), arg.clone(), stats.clone()])
        for a, b in zip(*runs):
            assert torch.equal(a, b)
        assert float(runs[0][-1][L.STAT_SIGMA]) == pytest.approx(0.02)     # rotated from the previous QERR_STD


def test_forward_multi_with_a_composed_crop_returns_no_argument(dev):
    """one LLaVA plan + a window that composes - advx_forward_multi never forms the resized window, so
    ops.forward_multi returns None for it (not an uninitialised buffer) and leaves a caller's buffer untouched; the
    pixel_values equal the two-launch form's at 1e-4 (the composed map drops one float32 rounding)."""
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    H, W = 80, 72
    gen = torch.Generator().manual_seed(5)
    x0 = torch.rand(3, H, W, generator=gen).to(dev)
    p = (torch.randn(3, H, W, generator=gen) * 0.4).to(dev)
    plan, win = Plan.llava(H, W, 64, 64), (5, 4, 60, 56)
    assert ops.crop_composes(plan, H, W, win)
    stats = torch.zeros(L.STATS_N, device=dev)
    scr = ops.image_scratch(H, W, 0, dev)
    s = torch.empty_like(x0)
    sentinel = torch.full_like(x0, -7.0)
    outs, arg = ops.forward_multi(p, x0, 0.5, stats, scr, [plan], [2], s, argument=sentinel, crop=win)
    assert arg is None and bool((sentinel == -7.0).all())
    outs2, arg2 = ops.forward_multi(p, x0, 0.5, stats, scr, [plan], [2], s, crop=win)
    assert arg2 is None and torch.equal(outs[0], outs2[0])
    # the two-launch form (window resized to an image, then the plan's resize)
    stats_b = torch.zeros(L.STATS_N, device=dev)
    _, arg_b = ops.image_fwd(p, x0, 0.5, stats_b, ops.image_scratch(H, W, 0, dev), crop=win, s=torch.empty_like(x0))
    ref = ops.emit(plan, arg_b, 2)
    assert rel_err(outs[0].cpu(), ref.cpu()) < 1e-4
    # a window that does not compose (three plans): the resized window IS returned
    plans = [Plan.llava(H, W, 64, 64), Plan.phi3(H, W), Plan.mllama(H, W, tile=32)]
    outs3, arg3 = ops.forward_multi(p, x0, 0.5, stats, scr, plans, [1, 1, 1], s, crop=win)
    assert arg3 is not None and torch.equal(arg3, arg_b)


@pytest.mark.parametrize("fam", ["llava", "qwen2vl", "phi3", "mllama"])
@pytest.mark.parametrize("size", [(336, 336), (512, 512), (400, 600)])
def test_full_size_reference_captures(dev, fam, size):
    """The HIP processors at BASELINE's image sizes with the models' real parameters, DIRECTLY against captures from the
    reference's own classes (full_size_reference.npz: checksums + sampled entries of pixel_values and image.grad) - not
    through the oracle."""
    from adversarialvlm_amd.plan import Plan
    g = load_golden("full_size_reference.npz")
    H, W = size
    k = f"{fam}_{H}x{W}"
    salt = int(g[f"{k}_salt"])
    plan = {"llava": lambda: Plan.llava(H, W), "qwen2vl": lambda: Plan.qwen2vl(H, W), "phi3": lambda: Plan.phi3(H, W, num_crops=6),
            "mllama": lambda: Plan.mllama(H, W, tile=560, max_tiles=4)}[fam]()
    shape = tuple(int(v) for v in g[f"{k}_shape"])
    assert tuple(plan.out_shape) == shape
    pv, grad = _run(plan, lcg_tensor((3, H, W), salt) + 0.5, lcg_tensor(shape, salt + 1), dev)
    d = pv.double()
    assert abs(float(d.sum()) - float(g[f"{k}_pv_sum"])) <= 2e-6 * float(d.abs().sum())
    assert abs(float((d * d).sum()) - float(g[f"{k}_pv_sumsq"])) <= 1e-5 * float(g[f"{k}_pv_sumsq"])
    assert rel_err(pv.flatten()[g[f"{k}_pv_idx"]], g[f"{k}_pv_val"]) < TIGHT
    gd = grad.double()
    assert abs(float((gd * gd).sum()) - float(g[f"{k}_grad_sumsq"])) <= 1e-5 * float(g[f"{k}_grad_sumsq"])
    assert rel_err(grad.flatten()[g[f"{k}_grad_idx"]], g[f"{k}_grad_val"]) < TIGHT
    ints = [int(v) for v in g[f"{k}_ints"]]
    i = plan.info
    if fam == "phi3":
        assert [int(i.image_h), int(i.image_w), int(i.num_img_tokens)] == ints
    elif fam == "mllama":
        assert [int(i.num_tiles)] == ints
    elif fam == "qwen2vl":
        assert [int(i.grid_h) * int(i.grid_w)] == ints


@pytest.mark.slow
@pytest.mark.timeout(900)
@pytest.mark.parametrize("fam,size", [("llava", (2160, 3840)), ("llava", (3000, 17)), ("qwen2vl", (2000, 3000)), ("qwen2vl", (31, 2600)),
                                      ("mllama", (1352, 1988)), ("mllama", (4000, 700)), ("phi3", (1988, 1352)), ("phi3", (200, 3000))])
def test_very_large_and_very_oblong_images_against_the_oracle(dev, fam, size):
    """Sizes far beyond BASELINE's 336 / 512: a 4K frame into LLaVA's 336 crop (35-tap antialiased rows, 8.3 M pixels per channel),
    images that are almost lines, the largest canvases the processors' own limits allow (Qwen2-VL's max_pixels, four Mllama tiles,
    Phi-3.5's sixteen-crop budget is 6 here).  Forward and gradient against the oracle at the processors' usual bar - index
    arithmetic (32-bit per sample, 64-bit across the batch), tap-table row lengths and launch grids are what this exercises."""
    from adversarialvlm_amd.plan import Plan
    H, W = size
    plan, oracle = {"llava": lambda: (Plan.llava(H, W), LlavaOracle()), "qwen2vl": lambda: (Plan.qwen2vl(H, W), Qwen2VLOracle()),
                    "mllama": lambda: (Plan.mllama(H, W), MllamaOracle()), "phi3": lambda: (Plan.phi3(H, W), Phi3Oracle())}[fam]()
    img = lcg_tensor((3, H, W), 900 + H % 97) + 0.5
    x = img.clone().requires_grad_(True)
    pv_ref = oracle.process(x)["pixel_values"]
    up = lcg_tensor(tuple(pv_ref.shape), 901)
    pv_ref.backward(up)
    pv, grad = _run(plan, img, up, dev)
    assert tuple(pv.shape) == tuple(pv_ref.shape)
    assert rel_err(pv, pv_ref.detach()) < TIGHT
    assert rel_err(grad, x.grad) < 2 * TIGHT
