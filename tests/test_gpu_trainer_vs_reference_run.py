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
RUNS = ["a", "b", "c", "d", "e", "f", "g", "h", "i", "j"]   # j: Qwen2-VL, ragged prompts in one batch; i: run a with --restart_num 2; h: the reference's Phi-3.5 plugin pair around the interface twin; g: BASELINE configs[0] (1 prompt, 2 PGD steps); d: Llama-3.2-Vision architecture + localized patch (configs[2]); e: Qwen2-VL;
                                         # f: prompts sampled from the pool + a target drawn per iteration (the global `random` stream)


def _pools(tmp, s):
    import json
    qf, af = os.path.join(tmp, "questions.json"), os.path.join(tmp, "answers.json")
    json.dump(s["pool"], open(qf, "w"))
    json.dump(s["answers"], open(af, "w"))
    return dict(questions_file=qf, answers_file=af, seed=s["seed"])


def _close(a, b, tol, floor=0.0):
    return abs(float(a) - float(b)) <= tol * max(abs(float(b)), floor)


@pytest.mark.parametrize("n", RUNS)
def test_train_equals_the_reference_trainers_run(tmp_path, n):
    from test_oracle_trainer_run import run_setup

    from adversarialvlm_amd import attack_model
    g = load_golden("trainer_run_reference.npz")
    iters, accum = int(g[f"{n}_iters"]), int(g[f"{n}_accum"])
    # the reference's loop found the CPU generator where building its model left it: run_setup rebuilds the draws from there
    s = run_setup(g, n)
    zs, fam = s["zs"], s["fam"]
    kind, size = (int(v) for v in g[f"{n}_mask"])
    step, gamma = g[f"{n}_sched"]
    tmp = str(tmp_path)
    Image.fromarray(g[f"{n}_image"]).save(os.path.join(tmp, "in.png"))

    components = ((lambda model_name, device: fam[0](device)), fam[1], fam[2])
    eng, hist = attack_model.train(
        exp_name="run", img_orig=os.path.join(tmp, "in.png"), prompt=s["prompt"], target_text="sure here it is",
        model_name=str(g[f"{n}_model"]), lr=1e-2, num_iterations=iters, save_steps=2, batch_size=s["B"], grad_accum_steps=accum,
        scheduler_step_size=int(step), scheduler_gamma=float(gamma), restart_num=int(g[f"{n}_restart"]),
        mask_type={0: "corner", 1: "bottom_lines", -1: None}[kind], mask_size=size if kind >= 0 else None, clamp_method="tanh",
        epsilon=0.5, sigma=1e-3, start_from_white=bool(int(g[f"{n}_white"])), target_text_random=s["target_random"], base_path=tmp,
        components=components, **_pools(tmp, s), return_engine=True, resaved_loss_every=1, log_every=1, unit_noise_fn=lambda it, shape: zs[it].view(shape))
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
    # every scalar key the reference handed to wandb.log is in this trainer's records (the media keys - a table and two images -
    # have no counterpart in the JSONL log); `accumulated_loss` once per optimiser step, with the reference's values
    media = {"generated_text", "optimized_image", "optimized_tensor"}
    probe = {"test_refuse_count", "test_target_acc", "test_target_first_word_acc", "test_total_questions"}      # the probe is off here
    probe |= {"fix_error_mean", "fix_error_std"}     # --restart_num's two numbers compare a 0..255 image with a [0, 1] perturbation (Q5: not kept)
    ours_keys = {k for h in hist for k in h}
    assert {str(k) for k in g[f"{n}_log_keys"]} - media - probe <= ours_keys, sorted({str(k) for k in g[f"{n}_log_keys"]} - media - probe - ours_keys)
    acc = [h["accumulated_loss"] for h in hist if "accumulated_loss" in h]
    assert len(acc) == len(g[f"{n}_accumulated_loss"]) and all(_close(a, b, 1e-4) for a, b in zip(acc, g[f"{n}_accumulated_loss"]))
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
    # the PNG the evaluation side reads (SafeBench_universal.py:34): the reference's, but for a uint8 level here and there where
    # the two devices' gradients put s on either side of a level
    png = np.asarray(Image.open(os.path.join(run, "optimized_image_iter_final.png")).convert("RGB")).astype(np.int32)
    d = np.abs(png - g[f"{n}_final_png"].astype(np.int32))
    assert d.max() <= 1 and int((d > 0).sum()) <= max(3, d.size // 2000), (int(d.max()), int((d > 0).sum()))


@pytest.mark.parametrize("n", ["x1", "x2", "x3", "x4"])      # x4: Phi-3.5 + Qwen2-VL + Llama-3.2-Vision (configs[3]); x3: the coin, a refusal per model or one target for all, sampled prompts
def test_cross_train_equals_the_reference_cross_trainers_run(tmp_path, n):
    """`crossattack_models.train()` of this package beside the reference's own (cross_trainer_run_reference.npz): x1 two LLaVA
    models with weights and gradient accumulation; x2 one model of each family whose architecture ships with transformers, the
    reference side running its own AdvMllamaInputs / AdvQwen2VLInputs / Differentiable*Processor classes.  Per model
    w_i CE_i + image loss (:369), their mean, `loss_resaved`, the quantise-error statistics, gradient norm, learning rate,
    optimiser-step count, the final image and the checkpoint names."""
    from test_oracle_trainer_run import cross_setup

    from adversarialvlm_amd import crossattack_models
    g = load_golden("cross_trainer_run_reference.npz")
    s = cross_setup(g, n)                                   # builds the models on the CPU in the run's order, then the draws
    names, iters, accum = s["names"], s["iters"], s["opt"]["grad_accum_steps"]
    kind, size = (int(v) for v in g[f"{n}_mask"])
    tmp = str(tmp_path)
    Image.fromarray(g[f"{n}_image"]).save(os.path.join(tmp, "in.png"))
    components = {m: ((lambda name, device, m=m: s["fam"][m][0](device)), s["fam"][m][1], s["fam"][m][2]) for m in names}
    eng, hist = crossattack_models.train(
        exp_name="run", img_orig=os.path.join(tmp, "in.png"), prompt=s["prompt"], target_text="sure here it is",
        model_names=names, lr=1e-2, num_iterations=iters, save_steps=2, batch_size=s["B"], grad_accum_steps=accum,
        scheduler_step_size=s["opt"]["scheduler_step_size"], scheduler_gamma=s["opt"]["scheduler_gamma"], restart_num=0,
        mask_type={0: "corner", 1: "bottom_lines", -1: None}[kind], mask_size=size if kind >= 0 else None, clamp_method="tanh",
        epsilon=0.3, sigma=5e-3, start_from_white=False, target_text_random=s["target_random"], DPO_flag=s["dpo"],   # Q3: neither is used
        refuse_prob=s["refuse_prob"], attack_norm=0.4, model_weights=s["weights"], base_path=tmp, components=components, **_pools(tmp, s), return_engine=True, resaved_loss_every=1, log_every=1,
        unit_noise_fn=lambda it, i, shape: s["zs"][it][i].view(shape))
    assert len(hist) == iters
    npx = g[f"{n}_final"].size
    for t, h in enumerate(hist):
        where = (n, t)
        for i, m in enumerate(names):
            assert _close(h[f"loss_{i}_{m}"], g[f"{n}_model_losses"][t][i], 1e-4), (where, i)
        assert _close(h["loss_per_iteration"], g[f"{n}_loss_per_iteration"][t], 1e-4), where
        assert _close(h["img_loss"], g[f"{n}_img_loss"][t], 1e-4), where
        assert _close(h["loss_resaved"], g[f"{n}_loss_resaved"][t], 1e-4), where
        assert _close(h["grad_norm"], g[f"{n}_grad_norm"][t], 1e-3), where
        assert _close(h["lr"], g[f"{n}_lr"][t], 1e-6) and int(h["global_iteration"]) == int(g[f"{n}_global_iteration"][t]), where
        assert abs(h["resave_error_std"] - g[f"{n}_resave_error_std"][t]) <= 1e-4 * g[f"{n}_resave_error_std"][t] + 2 / (255 * (npx - 1) ** 0.5), where
        assert _close(h["adversarial_mean"], g[f"{n}_adversarial_mean"][t], 1e-3, 1e-6), where
        assert _close(h["adversarial_std"], g[f"{n}_adversarial_std"][t], 1e-3, 1e-6), where
    media = {"generated_text", "optimized_image", "optimized_tensor"}
    probe = {"test_refuse_count", "test_target_acc", "test_target_first_word_acc", "test_total_questions"}
    ours_keys = {k for h in hist for k in h}
    assert {str(k) for k in g[f"{n}_log_keys"]} - media - probe <= ours_keys, sorted({str(k) for k in g[f"{n}_log_keys"]} - media - probe - ours_keys)
    acc = [h["accumulated_loss"] for h in hist if "accumulated_loss" in h]
    assert len(acc) == len(g[f"{n}_accumulated_loss"]) and all(_close(a, b, 1e-4) for a, b in zip(acc, g[f"{n}_accumulated_loss"]))
    run = os.path.join(tmp, "run")
    final = np.fromfile(os.path.join(run, "optimized_image_iter_final.bin"), dtype=np.float32)
    want = g[f"{n}_final"]
    assert final.shape == want.shape
    assert float(np.abs(final - want).max()) <= 2e-3 * 0.4 and rel_err(torch.tensor(final), torch.tensor(want), elementwise=None) < 1e-4
    ours = set(os.listdir(run))
    theirs = {str(f) for f in g[f"{n}_files"] if not str(f).startswith("test_results")}
    assert theirs <= ours, sorted(theirs - ours)
    assert not {f for f in ours - theirs if f.startswith("optimized_image")}


# ------------------------------------------------------------------------------ data parallelism against the reference run
def _dp_rank(rank, world, port, tmp, n, transport, out):
    """One rank of a two-rank run of `attack_model.train()`: the reference run's batch of 2 is sharded, rank r takes sample r
    of every noise draw; the image gradient is all-reduced (peer segments or the host library, gloo here)."""
    import torch.distributed as dist
    from test_oracle_trainer_run import run_setup

    from adversarialvlm_amd import attack_model
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    g = load_golden("trainer_run_reference.npz")
    s = run_setup(g, n)
    assert s["B"] == world
    fam = s["fam"]
    kind, size = (int(v) for v in g[f"{n}_mask"])
    step, gamma = g[f"{n}_sched"]
    eng, hist = attack_model.train(
        exp_name=f"dp_{transport}", img_orig=os.path.join(tmp, "in.png"), prompt=s["prompt"], target_text="sure here it is",
        model_name=str(g[f"{n}_model"]), lr=1e-2, num_iterations=s["iters"], save_steps=2, batch_size=s["B"],
        grad_accum_steps=int(g[f"{n}_accum"]), scheduler_step_size=int(step), scheduler_gamma=float(gamma), restart_num=0,
        mask_type={0: "corner", 1: "bottom_lines", -1: None}[kind], mask_size=size if kind >= 0 else None, clamp_method="tanh",
        epsilon=0.5, sigma=1e-3, start_from_white=bool(int(g[f"{n}_white"])), target_text_random=False, base_path=tmp,
        components=((lambda model_name, device: fam[0](device)), fam[1], fam[2]), return_engine=True, log_every=1,
        exchange_transport=transport, seed=s["seed"],
        unit_noise_fn=lambda it, shape: s["zs"][it][rank:rank + 1].reshape(shape))
    out[rank] = (eng.p.cpu(), hist, eng.exchange_report)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n,transport", [("a", "peer"), ("a", "rccl"), ("b", "peer")])
def test_two_ranks_reproduce_the_reference_trainers_batch_of_two(tmp_path, n, transport):
    """SURVEY 8(e) against the reference itself: the reference has no data parallelism - its run with batch 2 in ONE process is
    the truth - and two ranks of this package's trainer with one prompt each, the image gradient summed over the exchange and
    every loss pre-scaled by 1/world, must arrive where it arrived: same image loss, learning rate, optimiser-step count and
    quantise-error statistics every iteration (rank 0's log; its CE is the local sample's), same final image; replicas bit-identical."""
    import socket

    import torch.multiprocessing as mp
    g = load_golden("trainer_run_reference.npz")
    tmp = str(tmp_path)
    Image.fromarray(g[f"{n}_image"]).save(os.path.join(tmp, "in.png"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_dp_rank, args=(2, port, tmp, n, transport, out), nprocs=2, join=True)
    (p0, hist, report), (p1, _, _) = out[0], out[1]
    assert torch.equal(p0, p1)
    assert report["chosen"] == ("peer" if transport == "peer" else "host"), report
    npx = g[f"{n}_final"].size
    for t, h in enumerate(hist):
        assert _close(h["image_loss"], g[f"{n}_image_loss"][t], 1e-4), t
        assert _close(h["lr"], g[f"{n}_lr"][t], 1e-6) and int(h["global_iteration"]) == int(g[f"{n}_global_iteration"][t]), t
        # a gradient-accumulation window is exchanged ONCE, at its end (the reduction is linear): inside the window the logged
        # norm is the rank's own share; at the window's end it is the norm of the summed gradient, the reference's number
        if (t + 1) % int(g[f"{n}_accum"]) == 0:
            assert _close(h["grad norm"], g[f"{n}_grad_norm"][t], 1e-3), t
        assert abs(h["resave_error_std"] - g[f"{n}_resave_error_std"][t]) <= 1e-4 * g[f"{n}_resave_error_std"][t] + 2 / (255 * (npx - 1) ** 0.5), t
    final = np.fromfile(os.path.join(tmp, f"dp_{transport}", "optimized_image_iter_final.bin"), dtype=np.float32)
    want = g[f"{n}_final"]
    assert float(np.abs(final - want).max()) <= 2e-3 * 0.5 and rel_err(torch.tensor(final), torch.tensor(want), elementwise=None) < 1e-4


def _dp_cross_rank(rank, world, port, tmp, n, transport, out):
    """One rank of `crossattack_models.train()` under data parallelism: rank r holds model r % n_models (one model per rank
    group, BASELINE configs[3]) and, inside its group, shard r // n_models of the reference run's batch."""
    import torch.distributed as dist
    from test_oracle_trainer_run import cross_setup

    from adversarialvlm_amd import crossattack_models
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    g = load_golden("cross_trainer_run_reference.npz")
    s = cross_setup(g, n)
    names = s["names"]
    group, shard = world // len(names), rank // len(names)
    local = s["B"] // group
    kind, size = (int(v) for v in g[f"{n}_mask"])
    components = {m: ((lambda name, device, m=m: s["fam"][m][0](device)), s["fam"][m][1], s["fam"][m][2]) for m in names}
    eng, hist = crossattack_models.train(
        exp_name=f"dpx_{world}_{transport}", img_orig=os.path.join(tmp, "in.png"), prompt=s["prompt"], target_text="sure here it is",
        model_names=names, lr=1e-2, num_iterations=s["iters"], save_steps=2, batch_size=s["B"], grad_accum_steps=s["opt"]["grad_accum_steps"],
        scheduler_step_size=s["opt"]["scheduler_step_size"], scheduler_gamma=s["opt"]["scheduler_gamma"], restart_num=0,
        mask_type={0: "corner", 1: "bottom_lines", -1: None}[kind], mask_size=size if kind >= 0 else None, clamp_method="tanh",
        epsilon=0.3, sigma=5e-3, start_from_white=False, target_text_random=False, DPO_flag=False, attack_norm=0.4,
        model_weights=s["weights"], base_path=tmp, components=components, return_engine=True, log_every=1, seed=s["seed"],
        exchange_transport=transport,
        unit_noise_fn=lambda it, i, shape: s["zs"][it][i][shard * local:(shard + 1) * local].reshape(shape))
    out[rank] = (eng.p.cpu(), hist)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,transport", [(2, "peer"), (4, "peer"), (4, "rccl")])
def test_model_groups_reproduce_the_reference_cross_trainers_run(tmp_path, world, transport):
    """BASELINE configs[3] against the reference itself: the reference runs its models one after the other in ONE process
    (crossattack_models.py:352-391) - run x1, two models with weights 0.7 / 0.3, batch 2 - and this package runs one model per rank
    group, 2 ranks (one per model) or 4 (two per model, one prompt each), the image gradient averaged inside a group and SUMMED
    across groups by the one exchange: same image loss, learning rate, optimiser-step count and quantise-error statistics every
    iteration, rank 0's own model loss where its batch is the whole one, the same final image; replicas bit-identical."""
    import socket

    import torch.multiprocessing as mp
    n = "x1"
    g = load_golden("cross_trainer_run_reference.npz")
    tmp = str(tmp_path)
    Image.fromarray(g[f"{n}_image"]).save(os.path.join(tmp, "in.png"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_dp_cross_rank, args=(world, port, tmp, n, transport, out), nprocs=world, join=True)
    for r in range(1, world):
        assert torch.equal(out[0][0], out[r][0]), r
    hist = out[0][1]
    names = [str(v) for v in g[f"{n}_names"]]
    accum = int(g[f"{n}_accum"])
    npx = g[f"{n}_final"].size
    for t, h in enumerate(hist):
        assert _close(h["img_loss"], g[f"{n}_img_loss"][t], 1e-4), t
        assert _close(h["lr"], g[f"{n}_lr"][t], 1e-6) and int(h["global_iteration"]) == int(g[f"{n}_global_iteration"][t]), t
        if world == len(names):                                  # rank 0 sees its model's whole batch
            assert _close(h[f"loss_0_{names[0]}"], g[f"{n}_model_losses"][t][0], 1e-4), t
        if (t + 1) % accum == 0:                                 # the exchange happens where the optimiser steps
            assert _close(h["grad_norm"], g[f"{n}_grad_norm"][t], 1e-3), t
        assert abs(h["resave_error_std"] - g[f"{n}_resave_error_std"][t]) <= 1e-4 * g[f"{n}_resave_error_std"][t] + 2 / (255 * (npx - 1) ** 0.5), t
    final = np.fromfile(os.path.join(tmp, f"dpx_{world}_{transport}", "optimized_image_iter_final.bin"), dtype=np.float32)
    want = g[f"{n}_final"]
    assert float(np.abs(final - want).max()) <= 2e-3 * 0.4 and rel_err(torch.tensor(final), torch.tensor(want), elementwise=None) < 1e-4
