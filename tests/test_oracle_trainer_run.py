"""CPU tier: the oracle's PGD step sequence (oracle/pgd.py) against what the reference's OWN `attack_model.train()` logged
when it ran, in the build container, on the CPU around the tiny random LLaVA (tests/golden/trainer_run_reference.npz,
make_golden.py: golden_trainer_run - three runs: corner mask + StepLR decay on a resized image; gradient accumulation 2
with the default mask `(x_0 != 0)` at native size; start-from-white with bottom lines).

This pins row a1 (the loop: order of operations, sigma_noise(t+1) = std|q(s_t) - s_t| through the PNG round trip, the
masked gradient, the AdamW / StepLR cadence under accumulation, what is logged as what, what the final image is) to the
reference's code itself rather than to a reading of it.  The noise is rebuilt from the run's seed: `torch.randn_like` on
the global CPU generator is the only draw of the loop, one per iteration, after the model was built from seed 0."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import load_golden, rel_err
from oracle import pixel_ops as P
from oracle.pgd import PGDOracle
from oracle.processors import LlavaOracle

RUNS = ["a", "b", "c", "d", "e", "f", "g", "h", "i", "j"]   # j = Qwen2-VL, ragged prompts in one batch; i = run a with --restart_num 2 (a no-op, Q5); g = BASELINE configs[0]: 1 prompt, 2 PGD steps; h = the Phi-3.5 plugin pair
TOL = 2e-5        # same torch ops on both sides; the model's GEMMs may take another code path on another CPU


def families():
    """name in the captures -> (loader(device), AdvInputs, DifferentiableProcessor, oracle processor factory)"""
    from adversarialvlm_amd.testing import synthetic, synthetic_phi3v as F3, synthetic_vlms as S
    from oracle.processors import MllamaOracle, Phi3Oracle, Qwen2VLOracle
    llava0 = (lambda d: synthetic.load_model_and_processor("synthetic/tiny-llava", d, seed=0), synthetic.AdvLlavaInputs,
              synthetic.DifferentiableLlavaImageProcessor, lambda: LlavaOracle(56, 56))
    return {"tiny": llava0, "tiny-llava-0": llava0,
            "tiny-llava-1": (lambda d: synthetic.load_model_and_processor("synthetic/tiny-llava", d, seed=1), synthetic.AdvLlavaInputs,
                             synthetic.DifferentiableLlavaImageProcessor, lambda: LlavaOracle(56, 56)),
            "tiny-mllama": (lambda d: S.load_model_and_processor("synthetic/tiny-mllama", d, seed=2), S.AdvMllamaInputs,
                            S.DifferentiableMllamaImageProcessor, lambda: MllamaOracle(tile=S.MLLAMA_TILE, max_tiles=S.MLLAMA_MAX_TILES)),
            "tiny-phi3v": (lambda d: F3.load_model_and_processor("synthetic/tiny-phi3v", d, seed=4), F3.AdvPhiInputs,
                           F3.DifferentiablePhi3VImageProcessor, lambda: Phi3Oracle(num_crops=F3.PHI_NUM_CROPS)),
            "tiny-qwen2vl": (lambda d: S.load_model_and_processor("synthetic/tiny-qwen2vl", d, seed=3), S.AdvQwen2VLInputs,
                             S.DifferentiableQwen2VLImageProcessor,
                             lambda: Qwen2VLOracle(min_pixels=S.QWEN_MIN_PIXELS, max_pixels=S.QWEN_MAX_PIXELS))}


def run_setup(g, n):
    """-> dict(x0, mask, kwargs of the optimiser, iterations, noise draws, model, processor, ...) of run n"""
    fam = families()[str(g[f"{n}_model"])]
    model, proc = fam[0]("cpu")                       # leaves the CPU generator where the reference's loop found it
    iters, B = int(g[f"{n}_iters"]), int(g[f"{n}_batch"])
    img = g[f"{n}_image"]
    x0 = torch.tensor(img.astype(np.float32) / 255).permute(2, 0, 1).contiguous()
    if int(g[f"{n}_white"]):
        x0 = torch.ones_like(x0)
    oracle = fam[3]()
    pv = oracle.process(x0)["pixel_values"]
    shape = (B * pv.shape[0],) + tuple(pv.shape[1:])
    zs = [torch.randn(shape) for _ in range(iters)]
    kind, size = (int(v) for v in g[f"{n}_mask"])
    mask = P.create_mask({0: "corner", 1: "bottom_lines"}[kind], size, x0.shape) if kind >= 0 else (x0 != 0).float()
    assert float(mask.sum()) == float(g[f"{n}_mask_sum"])
    step, gamma = g[f"{n}_sched"]
    prompt, pool = str(g[f"{n}_prompt"]), [str(v) for v in g[f"{n}_questions"]]
    answers = [str(v) for v in g[f"{n}_answers"]]
    return dict(x0=x0, mask=mask, iters=iters, B=B, zs=zs, model=model, proc=proc, img=Image.fromarray(img), fam=fam, oracle=oracle,
                seed=int(g[f"{n}_seed"]), questions=pool if prompt == "list" else [prompt], pool=pool, answers=answers, prompt=prompt,
                target_random=bool(int(g[f"{n}_target_random"])),
                opt=dict(lr=1e-2, epsilon=0.5, sigma0=1e-3, scheduler_step_size=int(step), scheduler_gamma=float(gamma),
                         grad_accum_steps=int(g[f"{n}_accum"])))


def make_inputs(setup, device="cpu"):
    """The plugin as train() builds it (attack_model.py:263-270): the pool or the one prompt, the answer list under
    --target_text_random; prompts are drawn from the GLOBAL `random` stream, as in the reference."""
    return setup["fam"][1](questions=setup["questions"], test_questions=["hi"], batch_size=setup["B"], original_image=setup["img"],
                           processor=setup["proc"], device=device,
                           target_text=setup["answers"] if setup["target_random"] else "sure here it is")


def probe_inputs(setup):
    """The plugin the capture script asked for one batch after each run: one prompt, one target."""
    return setup["fam"][1](questions=["describe this image"], test_questions=["hi"], batch_size=setup["B"], original_image=setup["img"],
                           processor=setup["proc"], device="cpu", target_text="sure here it is")


def close(a, b, tol=TOL, floor=0.0):
    return abs(float(a) - float(b)) <= tol * max(abs(float(b)), floor)


@pytest.mark.parametrize("n", RUNS)
def test_plugin_batches_equal_the_reference_plugins(n):
    """get_inputs_train() of this package's AdvLlavaInputs (cached tokenisation) against the batch the reference's class
    assembled for the same prompt (llavaprocessor.py:80-108): ids, mask, suffix length and shift."""
    g = load_golden("trainer_run_reference.npz")
    s = run_setup(g, n)
    ip = probe_inputs(s)
    keys = sorted(k[len(n) + 4:] for k in g.files if k.startswith(f"{n}_in_"))
    for bound in (False, True):
        random.seed(1234)
        if bound:
            # index tensors from the PLAN geometry (what the trainers use) instead of the HF pass over the original image
            H, W = s["x0"].shape[1:]
            ip.bind_geometry(s["fam"][2](s["proc"].image_processor, "cpu"), H, W)
        enc = ip.get_inputs_train()
        assert sorted(enc.keys()) == keys
        for k in keys:
            want = torch.tensor(g[f"{n}_in_{k}"])
            assert enc[k].dtype == want.dtype and torch.equal(enc[k], want), (k, bound)
    assert [ip.suffix_length, ip.shift] == [int(v) for v in g[f"{n}_suffix"]]
    # ... and the inference prompt of the generation probe (get_inputs_inference of the reference's plugins)
    inf = ip.get_inputs_inference(s["img"], question="what is in this picture")
    keys = sorted(k[len(n) + 5:] for k in g.files if k.startswith(f"{n}_inf_"))
    assert sorted(k for k in inf.keys() if k != "pixel_values") == keys
    for k in keys:
        assert torch.equal(inf[k], torch.tensor(g[f"{n}_inf_{k}"])), k


@pytest.mark.parametrize("n", RUNS)
def test_oracle_loop_reproduces_the_reference_trainers_log(n):
    g = load_golden("trainer_run_reference.npz")
    s = run_setup(g, n)
    model, ip = s["model"], make_inputs(s)
    accum = s["opt"]["grad_accum_steps"]
    ora = PGDOracle(s["x0"], [s["oracle"]], mask=s["mask"], **s["opt"])
    last_s = None
    random.seed(s["seed"])                  # the capture seeded the global stream right before train()

    def side(inputs):
        return {k: v for k, v in inputs.items() if k != "pixel_values"}
    for t in range(s["iters"]):
        if s["target_random"]:
            ip.set_target_text(random.choice(ip.target_texts))                                     # :283-290
        inputs = ip.get_inputs_train()                                                             # :292 random.choices(pool, k=B)
        sigma = float(ora.sigma)
        noise = s["zs"][t] * sigma
        assert close(noise.std(), g[f"{n}_noise_std"][t], 1e-5, 1e-12) and abs(float(noise.mean()) - g[f"{n}_noise_mean"][t]) < 1e-9
        ora.forward(s["B"], [s["zs"][t]])
        ce = {}

        def loss_fn(pv):
            out = model(**side(inputs), pixel_values=pv)
            ce["v"] = ip.get_loss(out.logits[:, :-1, :])
            return ce["v"]
        ref = ora.backward_update(loss_fns=[loss_fn])
        assert close((float(ce["v"].detach()) + ref["img_loss"]) / accum, g[f"{n}_loss"][t])                 # attack_model.py:330
        assert close(ref["img_loss"], g[f"{n}_image_loss"][t])
        assert close(ref["grad_norm"], g[f"{n}_grad_norm"][t], 1e-4)                                # :340 (accumulated under accum)
        assert close(ref["sigma_next"], g[f"{n}_resave_error_std"][t], 1e-5, 1e-9)                  # :372
        assert close(ref["qerr_mean"], g[f"{n}_resave_error_mean"][t], 1e-5, 1e-9)
        assert close(ref["qerr_l1"], g[f"{n}_resave_error_l1"][t], 1e-5, 1e-6)
        assert close(ref["x_mean"], g[f"{n}_adversarial_mean"][t], 1e-4, 1e-7) and close(ref["x_std"], g[f"{n}_adversarial_std"][t], 1e-4, 1e-7)
        assert close(ora.current_lr(), g[f"{n}_lr"][t], 1e-12)                                      # scheduler.get_last_lr() after the step
        assert ora.opt_steps == int(g[f"{n}_global_iteration"][t])
        with torch.no_grad():                                                                       # :375-379 the re-saved forward
            pv1 = s["oracle"].process(P.quantise(ref["s"]))["pixel_values"]
            out = model(**side(inputs), pixel_values=pv1.repeat((s["B"],) + (1,) * (pv1.dim() - 1)))
            assert close(ip.get_loss(out.logits[:, :-1, :]), g[f"{n}_loss_resaved"][t])
        last_s = ref["s"]
    # the image written at the end is x_0 + x of the LAST iteration's forward, i.e. before that iteration's update (:473-477)
    assert rel_err(last_s.flatten(), g[f"{n}_final"]) <= 1e-6
    # ... and its PNG: uint8 TRUNCATION of clamp(s, 0, 1) * 255 (tensor2pil, llavaprocessor.py:151-155; Q1), level for level
    png = (P.quantise(last_s) * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
    assert np.array_equal(png, g[f"{n}_final_png"])


def test_restart_num_is_a_no_op_in_the_reference():
    """Q5, from the reference's own runs: with --restart_num 2 its train() logs the same numbers and writes the same image as
    without (the clamp-and-requantise of :447-457 rebinds a local the next iteration overwrites)."""
    g = load_golden("trainer_run_reference.npz")
    assert int(g["i_restart"]) == 2 and int(g["a_restart"]) == 0
    for k in ("loss", "image_loss", "loss_resaved", "resave_error_std", "grad_norm", "lr", "final"):
        assert np.array_equal(g[f"a_{k}"], g[f"i_{k}"]), k


# ------------------------------------------------------------------------------------------ the cross-model trainer
CROSS = ["x1", "x2", "x3", "x4"]          # x4 = BASELINE configs[3] by name: Phi-3.5 + Qwen2-VL + Llama-3.2-Vision


def cross_setup(g, n, device="cpu"):
    """Models built in the reference run's order (that is where its loop found the CPU generator), then its noise draws."""
    fam = families()
    names = [str(v) for v in g[f"{n}_names"]]
    loaded = [fam[m][0]("cpu") for m in names]                     # always on the CPU first: the draws follow
    img = g[f"{n}_image"]
    x0 = torch.tensor(img.astype(np.float32) / 255).permute(2, 0, 1).contiguous()
    oracles = [fam[m][3]() for m in names]
    B, iters = int(g[f"{n}_batch"]), int(g[f"{n}_iters"])
    shapes = []
    for o in oracles:
        pv = o.process(x0)["pixel_values"]
        shapes.append((B * pv.shape[0],) + tuple(pv.shape[1:]))
    zs = [[torch.randn(sh) for sh in shapes] for _ in range(iters)]
    kind, size = (int(v) for v in g[f"{n}_mask"])
    mask = P.create_mask({0: "corner", 1: "bottom_lines"}[kind], size, x0.shape) if kind >= 0 else (x0 != 0).float()
    step, gamma = g[f"{n}_sched"]
    prompt, pool = str(g[f"{n}_prompt"]), [str(v) for v in g[f"{n}_questions"]]
    return dict(names=names, fam=fam, loaded=loaded, x0=x0, oracles=oracles, B=B, iters=iters, zs=zs, mask=mask,
                img=Image.fromarray(img), weights=[float(w) for w in g[f"{n}_weights"]], seed=int(g[f"{n}_seed"]),
                questions=pool if prompt == "list" else [prompt], pool=pool, answers=[str(v) for v in g[f"{n}_answers"]], prompt=prompt,
                target_random=bool(int(g[f"{n}_target_random"])), dpo=bool(g[f"{n}_dpo"][0]), refuse_prob=float(g[f"{n}_dpo"][1]),
                opt=dict(lr=1e-2, epsilon=0.4, sigma0=1e-3, scheduler_step_size=int(step), scheduler_gamma=float(gamma),
                         grad_accum_steps=int(g[f"{n}_accum"])))


@pytest.mark.parametrize("n", CROSS)
def test_oracle_loop_reproduces_the_reference_cross_trainers_log(n):
    """oracle/pgd.py in cross mode against the reference's OWN `crossattack_models.train()` (cross_trainer_run_reference.npz):
    x1 two LLaVA models, weights 0.7 / 0.3, accumulation 2 (which only changes the optimiser's cadence there, :349-350),
    StepLR; x2 a LLaVA, a Llama-3.2-Vision and a Qwen2-VL architecture behind the reference's own plugin classes.  Logged per
    model: w_i CE_i + image loss (:369); `loss_resaved` = mean of the models' CE on the re-saved image (:434-447)."""
    g = load_golden("cross_trainer_run_reference.npz")
    s = cross_setup(g, n)
    models = [m for m, _ in s["loaded"]]
    ips = [s["fam"][m][1](questions=s["questions"], test_questions=["hi"], batch_size=s["B"], original_image=s["img"],
                          processor=proc, device="cpu", target_text=s["answers"] if s["target_random"] else "sure here it is")
           for m, (_, proc) in zip(s["names"], s["loaded"])]
    random.seed(s["seed"])
    ora = PGDOracle(s["x0"], s["oracles"], mask=s["mask"], model_weights=s["weights"], cross_mode=True, **s["opt"])
    k = len(models)
    for t in range(s["iters"]):
        if s["dpo"] or s["target_random"]:                                                         # crossattack_models.py:303-321
            coin = random.random()
            if s["dpo"] and coin < s["refuse_prob"]:
                for ip in ips:
                    ip.set_target_text(random.choice(ip.refuses))
            elif s["target_random"]:
                text = random.choice(ips[-1].target_texts)
                for ip in ips:
                    ip.set_target_text(text)
        inputs = [ip.get_inputs_train() for ip in ips]
        sigma = float(ora.sigma)
        assert close((s["zs"][t][-1] * sigma).std(), g[f"{n}_noise_std"][t], 1e-5, 1e-12)      # `noise` of the last model is logged
        ora.forward(s["B"], s["zs"][t])
        ces = [None] * k

        def make(i):
            def loss_fn(pv):
                out = models[i](**{kk: vv for kk, vv in inputs[i].items() if kk != "pixel_values"}, pixel_values=pv)
                ces[i] = ips[i].get_loss(out.logits[:, :-1, :])
                return ces[i]
            return loss_fn
        ref = ora.backward_update(loss_fns=[make(i) for i in range(k)])
        per_model = [s["weights"][i] * float(ces[i].detach()) + ref["img_loss"] for i in range(k)]
        for i in range(k):
            assert close(per_model[i], g[f"{n}_model_losses"][t][i]), (t, i)
        assert close(sum(per_model) / k, g[f"{n}_loss_per_iteration"][t])
        assert close(ref["img_loss"], g[f"{n}_img_loss"][t])
        assert close(ref["grad_norm"], g[f"{n}_grad_norm"][t], 1e-4)
        assert close(ref["sigma_next"], g[f"{n}_resave_error_std"][t], 1e-5, 1e-9)
        assert close(ref["qerr_mean"], g[f"{n}_resave_error_mean"][t], 1e-5, 1e-9) and close(ref["qerr_l1"], g[f"{n}_resave_error_l1"][t], 1e-5, 1e-6)
        assert close(ora.current_lr(), g[f"{n}_lr"][t], 1e-12) and ora.opt_steps == int(g[f"{n}_global_iteration"][t])
        with torch.no_grad():
            q = P.quantise(ref["s"])
            rs = []
            for i in range(k):
                pv1 = s["oracles"][i].process(q)["pixel_values"]
                pv = pv1.repeat((s["B"],) + (1,) * (pv1.dim() - 1))
                out = models[i](**{kk: vv for kk, vv in inputs[i].items() if kk != "pixel_values"}, pixel_values=pv)
                rs.append(float(ips[i].get_loss(out.logits[:, :-1, :])))
            assert close(sum(rs) / k, g[f"{n}_loss_resaved"][t])
        last_s = ref["s"]
    assert rel_err(last_s.flatten(), g[f"{n}_final"]) <= 1e-6


@pytest.mark.parametrize("n,batched", [("d", True), ("d", False), ("e", True), ("e", False), ("h", True), ("h", False)])
def test_generation_probe_equals_the_reference_probes_csv(tmp_path, n, batched):
    """`train_test.run_model_test` of this package (one left-padded `generate` per chunk of questions, or the reference's serial
    form) against the CSV and the statistics the reference's own `run_model_test` (train_test.py:6-86) produced at iteration 0 of
    its runs d (Llama-3.2-Vision architecture) and e (Qwen2-VL architecture): same header, same questions, same generated text,
    same four rates.  On the CPU: these two processors make their own pixel_values from the PNG, the HIP path is not involved."""
    import csv

    from adversarialvlm_amd.train_test import run_model_test
    g = load_golden("trainer_run_reference.npz")
    s = run_setup(g, n)
    ip = make_inputs(s)
    name = str(g[f"{n}_model"])
    want = [[str(c) for c in row] for row in g[f"{n}_probe0"]]
    questions = [row[0] for row in want[1:]]
    first, log = run_model_test([s["model"]], [s["proc"]], [ip], [name], questions, ip.target_texts[0], str(tmp_path), 0, s["img"],
                                batched=batched)
    with open(tmp_path / "test_results_iter_0.csv", newline="", encoding="utf-8") as f:
        got = [row for row in csv.reader(f)]
    assert got == want
    assert first == want[1]
    stats = [log[k] for k in ("test_target_first_word_acc", "test_target_acc", "test_refuse_count", "test_total_questions")]
    assert stats == [float(v) for v in g[f"{n}_probe0_stats"]]
