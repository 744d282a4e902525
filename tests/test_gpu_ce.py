"""GPU tier: suffix-only cross entropy - advx_ce_fwd / advx_ce_bwd against
torch's F.cross_entropy on the same logits (what llavaprocessor.py:73-78 computes), and the
trainer's --suffix_only_ce path against its default path."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("B,K,T,V", [(4, 6, 5, 1000), (3, 9, 9, 32000), (2, 4, 1, 77)])
def test_suffix_ce_matches_torch(dtype, tol, B, K, T, V):
    from adversarialvlm_amd.ce import suffix_cross_entropy
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(B * 100 + K)
    base = (torch.randn(B, K, V, generator=gen) * 3.0).to(dtype)
    targets = torch.randint(0, V, (B, T), generator=gen)
    # reference: float64 cross entropy of the SAME (already rounded) logits, mean over B*T rows
    # (torch's own fp32 CPU path is 1e-4 off the float64 value at V = 32000; the kernel is 5e-7 off)
    ref_in = base.double().clone().requires_grad_(True)
    ref = F.cross_entropy(ref_in[:, :T, :].permute(0, 2, 1), targets)
    (ref * 0.37).backward()
    x = base.to(dev).clone().requires_grad_(True)
    loss = suffix_cross_entropy(x, targets.to(dev))
    (loss * 0.37).backward()
    assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=1e-6)
    got, want = x.grad.double().cpu(), ref_in.grad
    assert got.shape == want.shape
    # element-wise: relative to the element (output rounding of the dtype) plus a small absolute floor
    floor = 1e-7 * float(want.abs().max()) + (6.1e-8 if dtype == torch.float16 else 0.0)    # half's subnormal spacing
    assert bool(((got - want).abs() <= tol * want.abs() + floor).all())
    assert not bool(got[:, T:, :].any())                       # unsupervised kept positions: exact zeros


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("B,K,T,V,lead", [
    (3, 4, 3, 32003, 0),          # a vocabulary that is no multiple of the 16-byte vector: rows start at every phase
    (2, 3, 2, 128256, 0),         # Llama-3.2-Vision: the 1024-thread form
    (2, 3, 3, 152064, 0),         # Qwen2-VL
    (2, 3, 2, 200003, 0),         # longer than any register form: two passes over memory
    (3, 5, 4, 4099, 3),           # a view that starts 3 elements into its rows: gradient rows sit at another phase
    (2, 2, 2, 5, 1), (2, 2, 1, 9, 0), (1, 1, 1, 16, 2),      # rows shorter than a vector or two
])
def test_suffix_ce_vector_forms_and_row_phases(dtype, tol, B, K, T, V, lead):
    """Round 4: 16-byte accesses.  A row's elements in front of its first 16-byte boundary and behind its last whole vector
    go one per thread; the forward holds a row in registers (256 x 16 or 1024 x 20 vectors) or streams it twice; the backward
    needs logits and gradient rows at the same phase or goes element by element."""
    from adversarialvlm_amd.ce import suffix_cross_entropy
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(V + 7 * K + lead)
    big = (torch.randn(B, K, V + lead, generator=gen) * 3.0).to(dtype)
    targets = torch.randint(0, V, (B, T), generator=gen)
    targets[0, 0] = V - 1                                   # the last element of a row (a tail element when V % 8 != 0)
    targets[-1, -1] = 0                                     # ... and the first one (a head element in a shifted view)
    ref_in = big[:, :, lead:].double().clone().requires_grad_(True)
    ref = F.cross_entropy(ref_in[:, :T, :].permute(0, 2, 1), targets)
    (ref * 0.37).backward()
    holder = big.to(dev).clone().requires_grad_(True)
    x = holder[:, :, lead:] if lead else holder
    loss = suffix_cross_entropy(x, targets.to(dev))
    (loss * 0.37).backward()
    assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=1e-6)
    got, want = holder.grad[:, :, lead:].double().cpu(), ref_in.grad
    floor = 1e-7 * float(want.abs().max()) + (6.1e-8 if dtype == torch.float16 else 0.0)
    assert bool(((got - want).abs() <= tol * want.abs() + floor).all())
    assert not bool(got[:, T:, :].any())
    if lead:
        assert not bool(holder.grad[:, :, :lead].any())     # nothing written in front of the view


def test_suffix_ce_random_shapes_phases_and_ignored_targets():
    """Sixty random (B, K, T, V, view offset, dtype) draws - every split of a row into head elements, whole vectors, tail elements
    and chunks - with some targets ignored, against float64 cross entropy of the same logits."""
    from adversarialvlm_amd.ce import suffix_cross_entropy
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2024)
    tols = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}
    for case in range(60):
        dtype = [torch.float32, torch.float16, torch.bfloat16][int(rng.integers(0, 3))]
        V = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, 70000)]))
        B, K = int(rng.integers(1, 4)), int(rng.integers(1, 5))
        T, lead = int(rng.integers(1, K + 1)), int(rng.integers(0, 8))
        gen = torch.Generator().manual_seed(1000 + case)
        big = (torch.randn(B, K, V + lead, generator=gen) * 4.0).to(dtype)
        targets = torch.randint(0, V, (B, T), generator=gen)
        if B * T > 1 and case % 3 == 0:
            targets[0, 0] = -100
        ref_in = big[:, :, lead:].double().clone().requires_grad_(True)
        ref = F.cross_entropy(ref_in[:, :T, :].permute(0, 2, 1), targets, ignore_index=-100)
        (ref * 1.7).backward()
        holder = big.to(dev).clone().requires_grad_(True)
        loss = suffix_cross_entropy(holder[:, :, lead:] if lead else holder, targets.to(dev))
        (loss * 1.7).backward()
        what = (case, dtype, B, K, T, V, lead)
        assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=1e-6, abs=1e-6), what
        got, want = holder.grad[:, :, lead:].double().cpu(), ref_in.grad
        floor = 1e-7 * float(want.abs().max()) + (6.1e-8 if dtype == torch.float16 else 0.0)
        assert bool(((got - want).abs() <= tols[dtype] * want.abs() + floor).all()), what
        assert not bool(got[:, T:, :].any()), what
        if lead:
            assert not bool(holder.grad[:, :, :lead].any()), what


def test_suffix_ce_strided_view_and_ignored_targets():
    """logits as a slice of a larger tensor (what `logits_to_keep` hands back is contiguous, a
    user slice need not be) and targets outside the vocabulary are ignored like ignore_index."""
    from adversarialvlm_amd.ce import suffix_cross_entropy
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    big = torch.randn(3, 12, 500, generator=gen).to(dev)
    x = big[:, 2:9, :]                                        # [3, 7, 500], batch stride 12*500
    targets = torch.randint(0, 500, (3, 4), generator=gen)
    targets[1, 2] = -100
    loss = suffix_cross_entropy(x, targets.to(dev))
    ref = F.cross_entropy(x[:, :4, :].permute(0, 2, 1).cpu(), targets, ignore_index=-100)
    assert float(loss) == pytest.approx(float(ref), rel=1e-5)


def test_trainer_suffix_only_ce_equals_the_default_path(tmp_path):
    """--suffix_only_ce changes how the loss is computed, not what it is: identical trajectory
    (same prompts, same noise) within float rounding of the two cross-entropy implementations."""
    from adversarialvlm_amd import attack_model
    tmp = str(tmp_path)
    path = os.path.join(tmp, "gray56.png")
    Image.fromarray(np.full((56, 56, 3), 128, np.uint8)).save(path)
    runs = {}
    for flag in (False, True):
        kw = dict(exp_name=f"ce{int(flag)}", img_orig=path, prompt="list", target_text="sure here it is",
                  model_name="synthetic/tiny-llava", lr=1e-2, num_iterations=4, save_steps=10, batch_size=4,
                  grad_accum_steps=1, scheduler_step_size=100, scheduler_gamma=1.0, restart_num=0, mask_type=None,
                  mask_size=None, clamp_method="tanh", epsilon=0.5, sigma=1e-3, start_from_white=False,
                  target_text_random=False, base_path=tmp, seed=3, return_engine=True, suffix_only_ce=flag)
        eng, hist = attack_model.train(**kw)
        runs[flag] = (eng.p.cpu().clone(), [h["ce_loss"] for h in hist])
    (p0, l0), (p1, l1) = runs[False], runs[True]
    assert l0 == pytest.approx(l1, rel=2e-3)                   # fp16 model: the default path rounds the loss in half
    assert float((p0 - p1).norm() / p0.norm()) < 2e-2


@pytest.mark.parametrize("k", [0, 1, 2])
def test_suffix_ce_against_the_reference_method(k):
    """The HIP suffix-only path on the last suffix_length + 1 positions against the loss and
    logits gradient the reference's own get_loss produced for the full [B, S, V] tensor
    (tests/golden/suffix_loss.npz): same loss, same gradient where it is non-zero, and the
    reference's gradient is zero everywhere the suffix-only path does not even compute logits."""
    from conftest import load_golden
    from adversarialvlm_amd.ce import suffix_cross_entropy
    dev = torch.device("cuda:0")
    g = load_golden("suffix_loss.npz")
    logits = torch.tensor(g[f"ce{k}_logits"])
    target = torch.tensor(g[f"ce{k}_target"])
    K = int(g[f"ce{k}_suffix_length"]) + 1
    kept = logits[:, -K:, :].to(dev).clone().requires_grad_(True)
    loss = suffix_cross_entropy(kept, target.to(dev))
    loss.backward()
    ref_grad = torch.tensor(g[f"ce{k}_logits_grad"])
    assert float(loss.detach()) == pytest.approx(float(g[f"ce{k}_loss"]), rel=1e-6)
    assert float((kept.grad.cpu() - ref_grad[:, -K:, :]).abs().max()) < 1e-7
    assert not bool(ref_grad[:, :-K, :].any())


def test_suffix_ce_random_shapes():
    """Random shapes: vocabulary sizes that are not multiples of the vector width, one supervised
    position, rows of large logits (stable log-sum-exp), ignored targets, kept positions beyond T."""
    from adversarialvlm_amd.ce import suffix_cross_entropy
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(40)
    for case in range(30):
        B, V = int(rng.integers(1, 6)), int(rng.choice([3, 17, 255, 1001, 4099, 32003]))
        T = int(rng.integers(1, 8))
        K = T + int(rng.integers(0, 3))
        dtype = [torch.float32, torch.float16, torch.bfloat16][case % 3]
        gen = torch.Generator().manual_seed(1000 + case)
        scale = float(rng.choice([0.5, 3.0, 30.0]))
        base = (torch.randn(B, K, V, generator=gen) * scale).to(dtype)
        targets = torch.randint(0, V, (B, T), generator=gen)
        if B * T > 1 and case % 4 == 0:
            targets.view(-1)[int(rng.integers(0, B * T))] = -100
        ref_in = base.double().clone().requires_grad_(True)
        ref = F.cross_entropy(ref_in[:, :T, :].permute(0, 2, 1), targets, ignore_index=-100)
        ref.backward()
        x = base.to(dev).clone().requires_grad_(True)
        loss = suffix_cross_entropy(x, targets.to(dev))
        loss.backward()
        assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=2e-6, abs=1e-6), (case, B, K, T, V, dtype, scale)
        # p = exp(x - lse) in fp32: the absolute rounding of x - lse (half an ulp of the largest logit, twice)
        # is the relative error of p, on top of the output rounding of the dtype
        tol = {torch.float32: 4e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype] + 2.5e-7 * float(base.float().abs().max())
        got, want = x.grad.double().cpu(), ref_in.grad
        floor = 2e-7 * float(want.abs().max()) + (6.1e-8 if dtype == torch.float16 else 0.0)
        bad = (got - want).abs() > tol * want.abs() + floor
        assert not bool(bad.any()), (case, B, K, T, V, dtype, scale, float(((got - want).abs() / (want.abs() + floor))[bad].max()))
