"""GPU tier: this package's `attack_model.train()` - HIP pixel path, the VLM on the GPU - laid beside three runs of the
reference's OWN `attack_model.train()` (tests/golden/trainer_run_reference.npz; captured on the CPU in the build container,
make_golden.py: golden_trainer_run): the same tiny random LLaVA (seed 0), the same image, prompt, target and flags, the
same noise draws (rebuilt from the run's seed and handed in through `unit_noise_fn`).

Compared per iteration: what both trainers log as `loss` (= (CE + image loss) / grad_accum_steps, attack_model.py:330),
`image_loss`, `loss_resaved` (:375-379), the quantise-error mean / std / L1 of the PNG round trip (:366-373,389-391),
`grad norm` (:340, accumulated under gradient accumulation), `lr` (:394), `global_iteration`, the adversarial mean / std;
then the images written: `optimized_image_iter_final.bin` and every checkpoint `.bin` / `.png` NAME the reference wrote
(Q10: the index is the optimiser-step count after the increment).  Bars: 1e-4 relative on the losses (north star), looser only
where stated."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
RUNS = ["a", "b", "c"]


def _close(a, b, tol, floor=0.0):
    return abs(float(a) - float(b)) <= tol * max(abs(float(b)), floor)


@pytest.mark.parametrize("n", RUNS)
def test_train_equals_the_reference_trainers_run(tmp_path, n):
    from adversarialvlm_amd import attack_model
    from adversarialvlm_amd.processors import synthetic
    g = load_golden("trainer_run_reference.npz")
    iters, accum = int(g[f"{n}_iters"]), int(g[f"{n}_accum"])
    # the reference's loop found the CPU generator where building the model from seed 0 left it: rebuild its draws
    synthetic.load_model_and_processor("synthetic/tiny-llava", "cpu", seed=0)
    zs = [torch.randn(2, 3, 56, 56) for _ in range(iters)]
    kind, size = (int(v) for v in g[f"{n}_mask"])
    step, gamma = g[f"{n}_sched"]
    tmp = str(tmp_path)
    Image.fromarray(g[f"{n}_image"]).save(os.path.join(tmp, "in.png"))

    def loader(model_name, device):
        return synthetic.load_model_and_processor("synthetic/tiny-llava", device, seed=0)
    components = (loader, synthetic.AdvLlavaInputs, synthetic.DifferentiableLlavaImageProcessor)
    eng, hist = attack_model.train(
        exp_name="run", img_orig=os.path.join(tmp, "in.png"), prompt="describe this image", target_text="sure here it is",
        model_name="tiny", lr=1e-2, num_iterations=iters, save_steps=2, batch_size=2, grad_accum_steps=accum,
        scheduler_step_size=int(step), scheduler_gamma=float(gamma), restart_num=0,
        mask_type={0: "corner", 1: "bottom_lines", -1: None}[kind], mask_size=size if kind >= 0 else None, clamp_method="tanh",
        epsilon=0.5, sigma=1e-3, start_from_white=bool(int(g[f"{n}_white"])), target_text_random=False, base_path=tmp,
        components=components, return_engine=True, resaved_loss_every=1, log_every=1, unit_noise_fn=lambda it, shape: zs[it].view(shape))
    assert len(hist) == iters
    for t, h in enumerate(hist):
        where = (n, t)
        assert _close(h["loss"], g[f"{n}_loss"][t], 1e-4), where
        assert _close(h["image_loss"], g[f"{n}_image_loss"][t], 1e-4), where
        assert _close(h["loss_resaved"], g[f"{n}_loss_resaved"][t], 1e-4), where
        assert _close(h["grad norm"], g[f"{n}_grad_norm"][t], 1e-3), where            # the two VLM copies' GEMMs run on different devices
        assert _close(h["lr"], g[f"{n}_lr"][t], 1e-6), where
        assert int(h["global_iteration"]) == int(g[f"{n}_global_iteration"][t]), where
        # the PNG round trip: one uint8 level of one pixel may flip between the two devices' gradients (1/(255 n) of the mean)
        npx = g[f"{n}_final"].size
        assert abs(h["resave_error_mean"] - g[f"{n}_resave_error_mean"][t]) <= 1e-4 * g[f"{n}_resave_error_mean"][t] + 2 / (255 * npx), where
        assert abs(h["resave_error_std"] - g[f"{n}_resave_error_std"][t]) <= 1e-4 * g[f"{n}_resave_error_std"][t] + 2 / (255 * (npx - 1) ** 0.5), where
        assert abs(h["resave_error_l1"] - g[f"{n}_resave_error_l1"][t]) <= 1e-4 * g[f"{n}_resave_error_l1"][t] + 2 / 255, where
        assert _close(h["adversarial_mean"], g[f"{n}_adversarial_mean"][t], 1e-3, 1e-6), where
        assert _close(h["adversarial_std"], g[f"{n}_adversarial_std"][t], 1e-3, 1e-6), where
    run = os.path.join(tmp, "run")
    final = np.fromfile(os.path.join(run, "optimized_image_iter_final.bin"), dtype=np.float32)
    # x_0 + x of the last forward; p has taken iters - 1 AdamW steps whose first ones are sign-like: compare where the
    # perturbation is not negligible (as tests/test_gpu_e2e.py does for p)
    want = g[f"{n}_final"]
    assert final.shape == want.shape
    assert float(np.abs(final - want).max()) <= 2e-3 * 0.5 and rel_err(torch.tensor(final), torch.tensor(want), elementwise=None) < 1e-4
    ours = set(os.listdir(run))
    theirs = {str(f) for f in g[f"{n}_files"] if not str(f).startswith("test_results")}      # the generation probe is off here
    assert theirs <= ours, sorted(theirs - ours)
    extra = {f for f in ours - theirs if f.startswith("optimized_image")}
    assert not extra, sorted(extra)                                                           # no checkpoint the reference did not write
    mask = torch.load(os.path.join(run, "mask.pt"))
    assert float(mask.sum()) == float(g[f"{n}_mask_sum"])
