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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("case", ["llava-4k-prepared", "qwen-6mp-blur-crop", "mllama-oblong-cross"])
def test_trajectories_on_very_large_images(dev, case):
    """The step itself far beyond BASELINE's sizes: a 4K frame behind LLaVA's processor in the prepared chain (24.9 M optimised
    values, 35-tap rows), a 2000 x 3000 image behind Qwen2-VL with blur 9 and a crop window per step (generic chain), and a
    4000 x 700 strip shared by Llama-3.2-Vision and LLaVA in cross mode - two steps each under the trajectory bar.  The vetting budget of p
    scales with the pixel count here, and says why: a resize that shrinks 11 x spreads every canvas gradient over ~130 image
    pixels, so at 4K the typical first gradient is 1e-4 and N P(|g| < a few adam_eps) is no longer a handful - measured on the
    4K case: 1117 of 24.9 M pixels (4.5e-5 of them) sit at |g| ~ 1.7e-8 = adam_eps, where the two implementations' gradients
    agree to 2e-11 absolute and AdamW's g / (|g| + eps) turns that into 5e-4 of a step.  Budget: 1e-4 of the optimised values;
    the rule itself (gradient agrees elementwise, every gradient seen <= max(1e3 adam_eps, 1e-3 max|g|)) is unchanged."""
    from adversarialvlm_amd.plan import Plan
    from oracle.processors import LlavaOracle
    g = torch.Generator().manual_seed(46)
    if case == "llava-4k-prepared":
        x0 = torch.rand(3, 2160, 3840, generator=g)
        worst = T._trajectory(dev, x0, [LlavaOracle()], [Plan.llava(2160, 3840)], [2], 2, fused_mode="prepared", max_ill=x0.numel() // 10000)
    elif case == "qwen-6mp-blur-crop":
        x0 = torch.rand(3, 2000, 3000, generator=g)
        wins = [(100, 200, 1500, 2400), (0, 0, 2000, 3000)]
        worst = T._trajectory(dev, x0, [Qwen2VLOracle()], [Plan.qwen2vl(2000, 3000)], [2], 2, blur_kernel=9, blur_sigma_fn=lambda t: [3.0, 0.7][t],
                              crop_fn=lambda t: wins[t], fused=False, max_ill=x0.numel() // 10000)
    else:
        x0 = torch.rand(3, 4000, 700, generator=g)
        worst = T._trajectory(dev, x0, [MllamaOracle(), LlavaOracle()], [Plan.mllama(4000, 700), Plan.llava(4000, 700)], [2, 3], 2,
                              weights=[0.6, 1.4], cross=True, gamma=0.9, max_ill=x0.numel() // 10000)
    assert worst["grad"] < 1e-5 and worst["pixel_values"] < 1e-5, worst
