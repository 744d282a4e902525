"""GPU tier: the crop window's resize composed with the plan's stage-0 resize into ONE table per axis
(include/advx.h "Composed crop"; csrc build_composed_row).  The reference evaluates the two resizes one after the other
(attack_model.py:307-314: RandomResizedCrop back to H x W, then process()); the composed map w_C = w_plan . w_window drops
the float32 rounding of the intermediate image, so the bar is 1e-4 against the oracle (L2 and elementwise, the trajectory
tests) and - here - agreement with the engine's own two-launch form far inside that bar, plus exact adjointness of the
composed forward and backward."""
import pytest
import torch

from conftest import rel_err
from oracle.pgd import PGDOracle
from oracle.processors import LlavaOracle, MllamaOracle, Phi3Oracle, Qwen2VLOracle

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _cases():
    from adversarialvlm_amd.plan import Plan
    return {
        "llava-down": (96, 128, lambda: Plan.llava(96, 128, 56, 72), lambda: LlavaOracle(56, 72)),
        "llava-same": (64, 64, lambda: Plan.llava(64, 64, 64, 64), lambda: LlavaOracle(64, 64)),
        "mllama-up": (60, 90, lambda: Plan.mllama(60, 90, tile=56, max_tiles=4), lambda: MllamaOracle(tile=56, max_tiles=4)),
        # Phi-3.5: stage 0 up-samples to the HD canvas (two-tap bilinear), a second stage reads that canvas; the larger windows compose
        "phi3-two-stages": (300, 400, lambda: Plan.phi3(300, 400), lambda: Phi3Oracle()),
        "qwen": (120, 150, lambda: Plan.qwen2vl(120, 150, min_pixels=56 * 56, max_pixels=28 * 28 * 64),
                 lambda: Qwen2VLOracle(min_pixels=56 * 56, max_pixels=28 * 28 * 64)),
    }


def _windows(H, W):
    return [(0, 0, H, W), (3, 5, H - 7, W - 9), (H // 4, W // 5, H // 2, W // 2), (H // 3, 0, H - H // 3, W // 3 + 2)]


@pytest.mark.parametrize("name", ["llava-down", "llava-same", "mllama-up", "qwen", "phi3-two-stages"])
@pytest.mark.parametrize("blur", [None, 5])
def test_composed_crop_matches_the_two_launch_form_and_the_oracle(name, blur):
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    H, W, mk_plan, mk_oracle = _cases()[name]
    gen = torch.Generator().manual_seed(5)
    x0 = torch.rand(3, H, W, generator=gen)
    B = 3
    windows = _windows(H, W)

    def run(separate):
        plan = mk_plan()
        eng = PixelPGD(x0.to(DEV), [plan], lr=1e-2, blur_kernel=blur, use_crop=True, allow_fused=False)
        g2 = torch.Generator().manual_seed(6)
        shape = (B * plan.out_shape[0],) + plan.out_shape[1:]
        outs = []
        for t, win in enumerate(windows):
            z = torch.randn(shape, generator=g2)
            g = torch.randn(shape, generator=g2) * 0.05
            composes = ops.crop_composes(plan, H, W, win)          # under whatever switch the caller set
            pv = eng.forward(B, [z.to(DEV)], blur_sigma=0.8 + 0.4 * t if blur else None, crop=win)[0]
            eng.backward_update([g.to(DEV)])
            outs.append((pv.cpu().clone(), eng.grad.cpu().clone(), eng.p.cpu().clone(), composes))
        return outs, eng.stats_dict()

    with ops.separate_crop():
        ref, st_ref = run(True)
    # every geometry through the composed tables, also Qwen2-VL (two gradient copies) and Phi-3.5 (two stages), where the default
    # keeps the two launches because composing was measured slower there
    with ops.compose_crop_everywhere():
        got, st_got = run(False)
    assert not any(r[3] for r in ref)                         # the switch really keeps the two launches
    assert sum(bool(r[3]) for r in got) >= (2 if name.startswith("phi3") else 3)      # and by default these windows compose
    for t, (a, b) in enumerate(zip(got, ref)):
        assert rel_err(a[0], b[0], elementwise=1e-5) < 2e-6, (t, "pixel_values")
        assert rel_err(a[1], b[1], elementwise=1e-5) < 2e-6, (t, "grad")
    # against the oracle, two steps with the windows that are not the whole image (composed wherever the tables fit)
    with ops.compose_crop_everywhere():
        plan = mk_plan()
        eng = PixelPGD(x0.to(DEV), [plan], lr=1e-2, blur_kernel=blur, use_crop=True, allow_fused=False)
        ora = PGDOracle(x0, [mk_oracle()], lr=1e-2, blur_kernel=blur)
        shape = (B * plan.out_shape[0],) + plan.out_shape[1:]
        for t in range(2):
            z, g = torch.randn(shape, generator=gen), torch.randn(shape, generator=gen) * 0.05
            win = windows[1 + t]
            sig = 1.1 if blur else None
            pv_ref = ora.forward(B, [z], blur_sigma=sig, crop=win)[0]
            pv = eng.forward(B, [z.to(DEV)], blur_sigma=sig, crop=win)[0]
            assert rel_err(pv.cpu(), pv_ref) < 1e-5
            ref_b = ora.backward_update([g])
            eng.backward_update([g.to(DEV)])
            assert rel_err(eng.grad.cpu(), ref_b["grad"]) < 1e-4


def test_composed_forward_and_backward_are_adjoint():
    """<C x, g> == <x, C^T g>: the forward and the transposed tables hold the same floats (same expressions, same order), so the
    two inner products agree to summation rounding."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    H, W = 96, 128
    plan = Plan.llava(H, W, 56, 72, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0))
    win = (7, 9, 70, 100)
    assert ops.crop_composes(plan, H, W, win)
    gen = torch.Generator().manual_seed(2)
    p = torch.zeros(3, H, W, device=DEV)
    x = torch.rand(3, H, W, generator=gen).to(DEV)             # with p = 0 the image s is x0 itself
    stats = torch.zeros(16, device=DEV)
    scratch = ops.image_scratch(H, W, 0, DEV)
    s = torch.empty_like(x)
    ws = torch.empty(plan.workspace_floats, device=DEV)
    plan.upload()
    outs, _ = ops.forward_multi(p, x, 0.5, stats, scratch, [plan], [1], s, argument=torch.empty_like(x), crop=win, workspaces=[ws])
    cx = outs[0].double().flatten()
    g = torch.randn(1, plan.out_numel, generator=gen).to(DEV)
    gs = ops.collect_crop(plan, g, 1, win, scratch, grad_s=torch.empty_like(x), workspace=ws)
    lhs = float((cx * g.double().flatten()).sum())
    rhs = float((x.double() * gs.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0), (lhs, rhs)
    assert float(gs[:, :7].abs().max()) == 0.0 and float(gs[:, :, :9].abs().max()) == 0.0       # exact zeros outside the window


def test_windows_that_do_not_compose_take_the_two_launch_path():
    """A window so small that a window row would feed more than sixteen canvas rows does not compose (advx_crop_composes = 0):
    the engine resizes it on its own as before - same API, same bars against the oracle; and advx_collect_crop refuses it."""
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    H, W = 96, 128
    plan = Plan.llava(H, W, 56, 72)
    tiny = (10, 20, 8, 10)
    assert not ops.crop_composes(plan, H, W, tiny) and ops.crop_composes(plan, H, W, (3, 5, 80, 100))
    gen = torch.Generator().manual_seed(1)
    x0 = torch.rand(3, H, W, generator=gen)
    eng = PixelPGD(x0.to(DEV), [plan], lr=1e-2, use_crop=True, allow_fused=False)
    ora = PGDOracle(x0, [LlavaOracle(56, 72)], lr=1e-2)
    for win in (tiny, (3, 5, 80, 100), tiny):
        z, g = torch.randn(2, 3, 56, 72, generator=gen), torch.randn(2, 3, 56, 72, generator=gen) * 0.05
        pv_ref = ora.forward(2, [z], crop=win)[0]
        pv = eng.forward(2, [z.to(DEV)], crop=win)[0]
        assert rel_err(pv.cpu(), pv_ref) < 1e-5
        ref = ora.backward_update([g])
        eng.backward_update([g.to(DEV)])
        assert rel_err(eng.grad.cpu(), ref["grad"]) < 1e-4
    with pytest.raises(L.AdvxError):
        ops.collect_crop(plan, torch.zeros(1, plan.out_numel, device=DEV), 1, tiny, ops.image_scratch(H, W, 0, DEV),
                         grad_s=torch.empty(3, H, W, device=DEV))
