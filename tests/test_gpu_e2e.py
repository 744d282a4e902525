"""GPU tier: the whole loop around a (tiny, random-init) LLaVA-architecture VLM.

BASELINE config 1 in spirit ("LLaVA tanh-clamp attack, 1 prompt, 2 PGD steps"): the reference
trainer cannot be imported here (wandb / torchvision), so its loop is the oracle's restatement
(oracle/pgd.py) driving the SAME random model on the CPU; the HIP engine drives it on the GPU.
Bar: fp32 loss and pixel grads within 1e-4 relative (north star)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import rel_err
from oracle.pgd import PGDOracle
from oracle.processors import LlavaOracle

pytestmark = pytest.mark.gpu


def _tiny(device):
    from adversarialvlm_amd.testing.synthetic import load_model_and_processor
    return load_model_and_processor("synthetic/tiny-llava", device, seed=0)


def _inputs(proc, device, batch, seed):
    from adversarialvlm_amd.processors import load_components
    import random
    _, AdvInputs, _ = load_components("synthetic/tiny-llava")
    return AdvInputs(questions=["what is in the image", "describe the scene please", "hi"], test_questions=["t"],
                     batch_size=batch, original_image=None, processor=proc, device=device, target_text="sure here it is",
                     rng=random.Random(seed))


@pytest.mark.parametrize("batch,chain", [(1, "pair"), (4, "pair"), (4, "step"), (3, "generic")])
def test_two_pgd_steps_loss_and_grad_parity(batch, chain):
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    dev = torch.device("cuda:0")
    model_g, proc = _tiny(dev)
    model_c, _ = _tiny("cpu")
    torch.manual_seed(3)
    x0 = torch.rand(3, 56, 56)
    ora = PGDOracle(x0, [LlavaOracle(56, 56)], lr=1e-2)
    kw = dict(allow_fused=False) if chain == "generic" else dict(fused_mode=chain)
    eng = PixelPGD(x0.to(dev), [Plan.llava(56, 56, 56, 56)], lr=1e-2, **kw)
    ip_g, ip_c = _inputs(proc, dev, batch, 5), _inputs(proc, "cpu", batch, 5)
    gen = torch.Generator().manual_seed(9)
    for step in range(2):
        z = torch.randn(batch, 3, 56, 56, generator=gen)
        inputs_c, inputs_g = ip_c.get_inputs_train(), ip_g.get_inputs_train()
        assert torch.equal(inputs_c["input_ids"], inputs_g["input_ids"].cpu())
        # ---- reference path (CPU, autograd through everything)
        ora.forward(batch, [z])
        losses = {}

        def loss_fn(pv):
            out = model_c(input_ids=inputs_c["input_ids"], attention_mask=inputs_c["attention_mask"], pixel_values=pv)
            l = ip_c.get_loss(out.logits[:, :-1, :])
            losses["c"] = float(l.detach())
            return l
        ref = ora.backward_update(loss_fns=[loss_fn])
        # ---- HIP path (GPU): pixel ops in libadvx, the VLM under torch
        pv = eng.forward(batch, [z.to(dev)])[0].requires_grad_(True)
        out = model_g(input_ids=inputs_g["input_ids"], attention_mask=inputs_g["attention_mask"], pixel_values=pv)
        loss = ip_g.get_loss(out.logits[:, :-1, :])
        (loss * eng.loss_scale(0)).backward()
        eng.backward_update([pv.grad])
        st = eng.stats_dict()
        assert abs(float(loss.detach()) - losses["c"]) <= 1e-4 * abs(losses["c"])
        assert rel_err(eng.grad.cpu(), ref["grad"]) < 1e-4
        assert abs(st["img_loss"] - ref["img_loss"]) <= 1e-4 * max(ref["img_loss"], 1e-12)
        assert abs(st["sigma_next"] - ref["sigma_next"]) <= 1e-4 * ref["sigma_next"]
        # AdamW's first steps are sign-like: a gradient entry that is ~0 may flip between the two
        # devices' GEMMs, so p is compared on the entries whose gradient is not negligible
        gmask = ref["grad"].abs() > 1e-3 * ref["grad"].abs().max()
        # (no elementwise bar here: the two VLM copies run their GEMMs on different devices, see above)
        assert rel_err(eng.p.cpu()[gmask], ora.p.detach()[gmask], elementwise=None) < 1e-3


def _gray(tmp_path, size=56):
    path = os.path.join(tmp_path, f"gray{size}.png")
    Image.fromarray(np.full((size, size, 3), 128, np.uint8)).save(path)
    return path


def test_train_entry_point_artifacts_and_progress(tmp_path):
    from adversarialvlm_amd import attack_model
    img = _gray(str(tmp_path))
    hist = attack_model.train(exp_name="t1", img_orig=img, prompt="list", target_text="sure here it is",
                              model_name="synthetic/tiny-llava", lr=1e-2, num_iterations=12, save_steps=5, batch_size=4,
                              grad_accum_steps=1, scheduler_step_size=100, scheduler_gamma=1.0, restart_num=0,
                              mask_type=None, mask_size=None, clamp_method="tanh", epsilon=0.5, sigma=1e-3,
                              start_from_white=False, target_text_random=False, base_path=str(tmp_path))
    run = os.path.join(str(tmp_path), "t1")
    files = set(os.listdir(run))
    # checkpoint index = global_iteration AFTER the increment (Q10): iterations 0, 5, 10, 11 -> 1, 6, 11, 12
    for k in (1, 6, 11, 12, "final"):
        assert f"optimized_image_iter_{k}.png" in files and f"optimized_image_iter_{k}.bin" in files, (k, sorted(files))
    assert {"mask.pt", "mask.png", "metrics.jsonl"} <= files
    raw = np.fromfile(os.path.join(run, "optimized_image_iter_final.bin"), dtype=np.float32)
    assert raw.size == 3 * 56 * 56
    png = np.array(Image.open(os.path.join(run, "optimized_image_iter_final.png")))
    assert np.array_equal(png, (raw.reshape(3, 56, 56).clip(0, 1) * 255).astype(np.uint8).transpose(1, 2, 0))
    assert len(hist) == 12 and hist[-1]["ce_loss"] < hist[0]["ce_loss"]       # the target gets more likely
    assert all(np.isfinite(h["loss"]) for h in hist)


def test_train_generic_chain_blur_crop_mask_multi_answer(tmp_path):
    from adversarialvlm_amd import attack_model
    img = _gray(str(tmp_path), 64)      # 64 -> 56: true resize, generic chain
    ans = os.path.join(str(tmp_path), "answers.json")
    json.dump(["sure here it is", "of course the answer is"], open(ans, "w"))
    hist = attack_model.train(exp_name="t2", img_orig=img, prompt="list", target_text="unused",
                              model_name="synthetic/tiny-llava", lr=1e-2, num_iterations=6, save_steps=100, batch_size=2,
                              grad_accum_steps=2, scheduler_step_size=1, scheduler_gamma=0.9, restart_num=0,
                              mask_type="corner", mask_size=32, clamp_method="tanh", epsilon=0.5, sigma=1e-3,
                              start_from_white=True, target_text_random=True, use_gaussian_blur=True,
                              gblur_kernel_size=5, gblur_sigma=7, use_local_crop=True, answers_file=ans,
                              base_path=str(tmp_path))
    assert len(hist) == 6 and hist[-1]["global_iteration"] == 3
    assert hist[-1]["lr"] == pytest.approx(1e-2 * 0.9 ** 3)
    raw = np.fromfile(os.path.join(str(tmp_path), "t2", "optimized_image_iter_final.bin"), dtype=np.float32).reshape(3, 64, 64)
    assert np.all(raw[:, 40:, :] == 1.0) and np.any(raw[:, :32, :32] != 1.0)     # only the masked corner moved


def test_cross_trainer_two_models(tmp_path):
    from adversarialvlm_amd import crossattack_models
    img = _gray(str(tmp_path), 70)
    hist = crossattack_models.train(exp_name="t3", img_orig=img, prompt="list", target_text="sure here it is",
                                    model_names=["synthetic/tiny-llava", "synthetic/tiny-llava"], lr=1e-2,
                                    num_iterations=4, save_steps=2, batch_size=2, grad_accum_steps=1,
                                    scheduler_step_size=100, scheduler_gamma=0.9, restart_num=0, mask_type=None,
                                    mask_size=None, clamp_method="tanh", epsilon=0.4, sigma=1e-3, start_from_white=False,
                                    target_text_random=False, DPO_flag=False, model_weights=[0.2, 0.8],
                                    use_gaussian_blur=True, gblur_kernel_size=5, base_path=str(tmp_path),
                                    resaved_loss_every=3, suffix_only_ce=True, pixel_io="model")
    assert len(hist) == 4 and all(np.isfinite(h["loss_per_iteration"]) for h in hist)
    assert [("loss_resaved" in h) for h in hist] == [True, False, False, True] and np.isfinite(hist[3]["loss_resaved"])
    assert "optimized_image_iter_final.png" in os.listdir(os.path.join(str(tmp_path), "t3"))


def _kw(tmp, name, iters, **extra):
    kw = dict(exp_name=name, img_orig=_gray(tmp), prompt="list", target_text="sure here it is",
              model_name="synthetic/tiny-llava", lr=1e-2, num_iterations=iters, save_steps=3, batch_size=4,
              grad_accum_steps=1, scheduler_step_size=2, scheduler_gamma=0.8, restart_num=0, mask_type=None,
              mask_size=None, clamp_method="tanh", epsilon=0.5, sigma=1e-3, start_from_white=False,
              target_text_random=False, base_path=tmp, seed=3)
    kw.update(extra)
    return kw


@pytest.mark.parametrize("size,batch,mode", [(56, 4, "step"), (56, 20, "pair"), (70, 4, "prepared")])
def test_resume_continues_bit_for_bit(tmp_path, size, batch, mode):
    """optimiser moments, schedule, RNG streams and noise counters are saved, so
    4 iterations + resume + 3 more equal 7 iterations in one go, bit for bit - in the one-launch step chain (what
    `auto` picks for a native-size image and up to 16 prompts), in the fused pair (more prompts) and in the prepared
    chain (70x70 image resized to the model's 56x56)."""
    from adversarialvlm_amd import attack_model
    tmp = str(tmp_path)
    img = _gray(tmp, size)
    eng, _ = attack_model.train(**_kw(tmp, "full", 7, img_orig=img, batch_size=batch, return_engine=True))
    assert eng.mode == mode
    attack_model.train(**_kw(tmp, "part", 4, img_orig=img, batch_size=batch))
    # iteration 3 (save_steps=3) wrote state_iter_4.pt: resume from it and run iterations 4..6
    attack_model.train(**_kw(tmp, "rest", 7, img_orig=img, batch_size=batch,
                             resume_from=os.path.join(tmp, "part", "state_iter_4.pt")))
    a = np.fromfile(os.path.join(tmp, "full", "optimized_image_iter_final.bin"), dtype=np.float32)
    b = np.fromfile(os.path.join(tmp, "rest", "optimized_image_iter_final.bin"), dtype=np.float32)
    assert a.size == 3 * size * size and np.array_equal(a, b)


def test_resume_stays_on_the_chain_that_wrote_the_state(tmp_path):
    """the state records its kernel chain.  A run started with 4 prompts (one-launch `step` chain) and resumed with
    20 (where `auto` alone would pick the pair, which addresses the noise generator differently) stays on the step chain; an
    engine on another chain refuses the state instead of going on with another noise stream."""
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import attack_model
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    tmp = str(tmp_path)
    img = _gray(tmp, 56)
    attack_model.train(**_kw(tmp, "part", 4, img_orig=img, batch_size=4))
    path = os.path.join(tmp, "part", "state_iter_4.pt")
    assert attack_model.saved_chain(path) == "step"
    eng, _ = attack_model.train(**_kw(tmp, "rest", 6, img_orig=img, batch_size=20, resume_from=path, return_engine=True))
    assert eng.mode == "step"
    sd = torch.load(path, map_location="cpu")["engine"]
    pair = PixelPGD(torch.rand(3, 56, 56, device="cuda:0"), [Plan.llava(56, 56, 56, 56)], fused_mode="pair")
    with pytest.raises(L.AdvxError, match="written by the 'step' chain"):
        pair.load_state_dict(sd)
    sd.pop("chain")                          # a file from before round 4 carries no chain: accepted as before
    pair.load_state_dict(sd)


def test_accumulated_loss_never_mixes_accumulation_windows(tmp_path):
    """ADVICE r03 (attack_model.py:349-353 of the reference resets at EVERY optimiser step): with the default log cadence the key
    is the sum of the window's `loss` values; under a sparser cadence a window that was only partly logged has no key at all."""
    from adversarialvlm_amd import attack_model
    tmp = str(tmp_path)
    _, dense = attack_model.train(**_kw(tmp, "dense", 8, grad_accum_steps=2, return_engine=True))
    assert [("accumulated_loss" in h) for h in dense] == [False, True] * 4
    for k in (1, 3, 5, 7):
        assert dense[k]["accumulated_loss"] == pytest.approx(dense[k - 1]["loss"] + dense[k]["loss"], rel=1e-12)
    _, sparse = attack_model.train(**_kw(tmp, "sparse", 8, grad_accum_steps=2, log_every=3, return_engine=True))
    assert [h["iteration"] for h in sparse] == [0, 3, 6, 7]
    # iteration 3 steps, but its window (2, 3) was logged in part; iteration 7's window (6, 7) was logged in full
    assert "accumulated_loss" not in sparse[1] and "accumulated_loss" not in sparse[0] and "accumulated_loss" not in sparse[2]
    assert sparse[3]["accumulated_loss"] == pytest.approx(sparse[2]["loss"] + sparse[3]["loss"], rel=1e-12)
    # same run, same numbers: the log cadence changes what is written, not what is computed.  Everything this package
    # computes is compared exactly; the scalar `loss` comes out of torch's F.cross_entropy, whose mean reduction on ROCm is not
    # reproducible to the last bit (300 calls on identical logits: 243 x 6.235054016, 57 x 6.235054493 - tools/diag_flaky.py,
    # round 4), so it gets one part in a million
    for a, b in ((sparse[3], dense[7]), (sparse[1], dense[3])):
        assert a["loss"] == pytest.approx(b["loss"], rel=1e-6) and a["ce_loss"] == pytest.approx(b["ce_loss"], rel=1e-6)
        for key in ("image_loss", "grad norm", "lr", "resave_error_std", "resave_error_mean", "resave_error_l1",
                    "adversarial_mean", "adversarial_std", "noise_sigma"):
            assert a[key] == b[key], key


def test_generation_probe_writes_reference_csv(tmp_path):
    import csv
    from adversarialvlm_amd import attack_model
    tmp = str(tmp_path)
    attack_model.train(**_kw(tmp, "probe", 2, generation_probe=True))
    rows = list(csv.reader(open(os.path.join(tmp, "probe", "test_results_iter_0.csv"))))
    assert rows[0] == ["question", "synthetic/tiny-llava"] and len(rows) == 1 + 8
    logged = [json.loads(l) for l in open(os.path.join(tmp, "probe", "metrics.jsonl"))]
    assert any("test_target_acc" in r for r in logged)


def test_loss_resaved_is_the_forward_of_the_png_image(tmp_path):
    """SURVEY a16 (attack_model.py:366-379): the optional second forward on the image as its PNG
    would be read back - quantised, processed, repeated, NO noise.  The logged value must equal
    the model's loss on pixel_values built by the oracle from the checkpointed image."""
    from adversarialvlm_amd import attack_model
    from oracle import pixel_ops as P
    from oracle.processors import LlavaOracle
    tmp = str(tmp_path)
    eng, hist = attack_model.train(**_kw(tmp, "rs", 3, resaved_loss_every=2, return_engine=True))
    assert "loss_resaved" in hist[0] and "loss_resaved" in hist[2] and "loss_resaved" not in hist[1]
    assert all(np.isfinite(h["loss_resaved"]) for h in (hist[0], hist[2]))
    # the engine's re-saved pixel_values against the oracle's processor on the quantised image
    img = eng.image().cpu()
    want = LlavaOracle(img.shape[1], img.shape[2]).process(P.quantise(img))["pixel_values"].repeat(4, 1, 1, 1)
    got = eng.resaved_pixel_values(4)[0].cpu()
    assert got.shape == want.shape and float((got - want).abs().max()) < 2e-6
    # a noiseless forward of (nearly) the same image: close to, but not the same as, the training loss
    assert abs(hist[2]["loss_resaved"] - hist[2]["ce_loss"]) < 0.5


def test_batched_generation_probe_equals_the_serial_one(tmp_path):
    """one left-padded `generate` per model instead of one per question - same decoded text, same CSV,
    same statistics as the reference's serial loop (train_test.py:42-65), with prompts of different lengths and a
    chunk size that does not divide their number."""
    import csv
    from adversarialvlm_amd.processors import load_components
    from adversarialvlm_amd.train_test import run_model_test
    dev = torch.device("cuda:0")
    model, proc = _tiny(dev)
    _, AdvInputs, DiffProc = load_components("synthetic/tiny-llava")
    questions = ["what is in the image", "hi", "describe the scene in a few words please", "and now", "tell me more about it",
                 "one two three four five six seven", "why"]
    ip = AdvInputs(questions=questions, test_questions=questions, batch_size=2, original_image=None, processor=proc, device=dev,
                   target_text="sure here it is")
    ap = DiffProc(proc.image_processor, dev)
    img = Image.fromarray((np.random.default_rng(3).random((56, 56, 3)) * 255).astype(np.uint8))
    tmp = str(tmp_path)
    res = {}
    for name, kw in (("serial", dict(batched=False)), ("batched", dict(batched=True, probe_batch=3))):
        first, log = run_model_test([model], [proc], [ip], ["synthetic/tiny-llava"], questions, "sure here it is", tmp, name, img,
                                    adv_processors=[ap], max_new_tokens=12, **kw)
        res[name] = (first, log, list(csv.reader(open(os.path.join(tmp, f"test_results_iter_{name}.csv")))))
    assert res["serial"] == res["batched"]
    assert len(res["batched"][2]) == 1 + len(questions) and len({len(r[1].split()) for r in res["batched"][2][1:]}) > 1
