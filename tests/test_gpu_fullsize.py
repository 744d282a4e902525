"""GPU tier: the per-rank workloads of BASELINE.json configs[2..4] at their FULL sizes, against the oracle.

The 8-GPU forms of those configurations shard prompts; what one rank executes is
  * configs[2]: Llama-3.2-Vision, 336x336 image -> 4 x 560x560 tiles, 32 prompts, localized `corner` patch n = 100
    (/root/reference/scripts/attacks/attack_clamp_tanh_llama-localize.sh:25-32) - the prepared chain;
  * configs[3] / [4]: Phi-3.5 + Qwen2-VL + Llama-3.2-Vision over one 336x336 image, Gaussian blur 5 with a sigma drawn
    per step, 16 prompts per model, weights 0.2 / 0.8 / 1.6 (attack_cross.sh:21-54) - the multi-plan generic chain.
Two to three steps each under the trajectory bar of tests/test_gpu_pgd.py (L2 ratio and elementwise), plus the
kept-zero-padding run of the first one against the reference-shaped run.  The oracle needs a few GB of host memory and
tens of seconds of CPU here - these are the slowest tests of the tier."""
import pytest
import torch

import test_gpu_pgd as T
from oracle import pixel_ops as P
from oracle.processors import MllamaOracle, Phi3Oracle, Qwen2VLOracle

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.timeout(900)
def test_config3_rank_workload_mllama_localized_patch_full_size(dev):
    from adversarialvlm_amd.plan import Plan
    x0 = torch.rand(3, 336, 336, generator=torch.Generator().manual_seed(41))
    mask = P.create_mask("corner", 100, (3, 336, 336))
    plan = Plan.mllama(336, 336)
    assert plan.out_shape == (1, 1, 4, 3, 560, 560) and int(plan.info.num_tiles) == 1
    worst = T._trajectory(dev, x0, [MllamaOracle()], [plan], [32], 2, mask=mask, fused_mode="prepared")
    assert worst["grad"] < 1e-5 and worst["pixel_values"] < 1e-5, worst


@pytest.mark.timeout(600)
def test_config3_rank_workload_padding_kept_zero_full_size(dev):
    """noise_on_padding=False at full size (B = 32, 3 of 4 tiles padding): same values and noise wherever an image
    is, exact zeros elsewhere, bit-identical p - in the prepared chain the trainers take."""
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    x0 = torch.rand(3, 336, 336, generator=torch.Generator().manual_seed(42)).to(dev)
    mask = P.create_mask("corner", 100, (3, 336, 336)).to(dev)
    engines = [PixelPGD(x0, [Plan.mllama(336, 336)], seed=9, mask=mask, noise_on_padding=flag) for flag in (True, False)]
    assert all(e.mode == "prepared" for e in engines)
    pl = engines[0].plans[0]
    lo, hi = pl.live_range()
    assert (lo, hi) == (0, 3 * 560 * 560)
    gen = torch.Generator(device=dev).manual_seed(43)
    for t in range(3):
        ref, kept = [e.forward(32)[0].reshape(32, pl.out_numel) for e in engines]
        assert torch.equal(ref[:, lo:hi], kept[:, lo:hi])
        assert not kept[:, hi:].any() and float(ref[:, hi:].abs().max()) > 0.0
        g = torch.randn(32, pl.out_numel, generator=gen, device=dev) * 0.01
        for e in engines:
            e.backward_update([g])
        assert torch.equal(engines[0].p, engines[1].p)
    assert engines[0].stats_dict() == engines[1].stats_dict()


@pytest.mark.timeout(900)
def test_config4_rank_workload_cross_three_models_blur_full_size(dev):
    from adversarialvlm_amd.plan import Plan
    x0 = torch.rand(3, 336, 336, generator=torch.Generator().manual_seed(44))
    plans = [Plan.phi3(336, 336), Plan.qwen2vl(336, 336), Plan.mllama(336, 336)]
    assert plans[0].out_shape == (1, 7, 3, 336, 336) and plans[1].out_shape == (576, 1176)
    sig = [0.37, 1.62]
    worst = T._trajectory(dev, x0, [Phi3Oracle(), Qwen2VLOracle(), MllamaOracle()], plans, [16, 16, 16], 2, blur_kernel=5,
                          blur_sigma_fn=lambda t: sig[t], weights=[0.2, 0.8, 1.6], cross=True, gamma=0.9)
    assert worst["grad"] < 1e-5 and worst["pixel_values"] < 1e-5, worst


@pytest.mark.timeout(900)
def test_llava_512_blur9_crop_batch64_full_size_composed(dev):
    """The reference's production preset for LLaVA (scripts/attacks/attack_clamp_tanh_llava_gblur.sh:24-60 with --use_local_crop:
    512 x 512 image, Gaussian blur 9, a random-resized-crop window per step) at the benchmark's batch, against the oracle: the
    crop window's resize and the processor's 512 -> 336 resize go through the composed tables (round 3), one gather each way."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    from oracle.processors import LlavaOracle
    x0 = torch.rand(3, 512, 512, generator=torch.Generator().manual_seed(45))
    plan = Plan.llava(512, 512)
    windows = [(40, 30, 400, 420), (100, 30, 343, 458), (0, 0, 512, 512)]
    assert all(ops.crop_composes(plan, 512, 512, w) for w in windows)
    sig = [10.0, 7.0, 1.3]
    # Vetting budget: how many pixels have a first gradient within a few adam_eps of zero is N * P(|g| < tau) ~ N * 2 tau / (2.5 sd(g)):
    # here N = 786 432 and the blurred gradient field has max|g| = 0.06 (sd ~ 0.015), i.e. about ten pixels below tau = 3e-7 -
    # measured with the default budget of 2: six (g = 1.9e-7 against max|g| = 6e-2, the gradients agreeing to 3e-9).  Twelve, absolute.
    worst = T._trajectory(dev, x0, [LlavaOracle()], [plan], [64], 3, blur_kernel=9, blur_sigma_fn=lambda t: sig[t],
                          crop_fn=lambda t: windows[t], fused=False, max_ill=12)
    assert worst["grad"] < 1e-5 and worst["pixel_values"] < 1e-5, worst
