"""GPU tier: BASELINE configs[2] and [3] at the level this box allows - the TRAINERS under data parallelism with the
non-LLaVA families (ranks share the one GPU, gloo carries the host collectives, the image gradient goes through the
peer exchange or the host all-reduce).

  configs[2]  Llama-3.2-Vision tanh + localized-patch attack, prompts sharded over the ranks, one all-reduce of the
              pixel gradient per step: attack_model.train on two ranks with synthetic/tiny-mllama and a corner mask;
  configs[3]  cross-model attack, one model per rank group, gradients summed across groups
              (crossattack_models.py:352-391): crossattack_models.train on THREE ranks holding a LLaVA, a Mllama and a
              Qwen2-VL architecture, blur + crop + weights.
What is asserted: every rank ends with bit-identical (p, image) - the replica check after EVERY step inside the trainers
passes only if all ranks drew the same shared values and applied the same reduced gradient -, the masked region alone moved,
the artefacts exist.  The 8-GPU forms of these configs have never run on hardware."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from PIL import Image

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _image(tmp, h, w):
    path = os.path.join(tmp, f"gray_{h}x{w}.png")
    if not os.path.exists(path):
        part = f"{path}.{os.getpid()}.tmp"
        Image.fromarray(np.full((h, w, 3), 128, np.uint8)).save(part, format="PNG")
        os.replace(part, path)
    return path


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo")


def _mllama_rank(rank, world, port, tmp, transport, out):
    from adversarialvlm_amd import attack_model
    _init(rank, world, port)
    eng, hist = attack_model.train(
        exp_name=f"mllama_dp_{transport}", img_orig=_image(tmp, 60, 90), prompt="list", target_text="sure here it is",
        model_name="synthetic/tiny-mllama", lr=1e-2, num_iterations=5, save_steps=2, batch_size=4, grad_accum_steps=1,
        scheduler_step_size=100, scheduler_gamma=1.0, restart_num=0, mask_type="corner", mask_size=40, clamp_method="tanh",
        epsilon=0.5, sigma=1e-3, start_from_white=False, target_text_random=False, base_path=tmp, seed=7, return_engine=True,
        replica_check_every=1, exchange_transport=transport)
    out[rank] = (eng.p.cpu(), eng.image().cpu(), eng.mode, eng.world, len(hist))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("transport", ["peer", "rccl"])
def test_mllama_localized_patch_two_ranks(tmp_path, transport):
    tmp = str(tmp_path)
    _image(tmp, 60, 90)
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_mllama_rank, args=(2, _free_port(), tmp, transport, out), nprocs=2, join=True)
    (p0, s0, mode0, world0, n0), (p1, s1, _, _, n1) = out[0], out[1]
    assert torch.equal(p0, p1) and torch.equal(s0, s1)                    # replicas bit-identical
    assert mode0 == "prepared" and world0 == 2 and n0 == 5 and n1 == 0   # four launches + exchange; only rank 0 logs
    assert float(p0[:, :40, :40].abs().max()) > 0 and float(p0[:, 40:, :].abs().max()) == 0 and float(p0[:, :, 40:].abs().max()) == 0
    files = set(os.listdir(os.path.join(tmp, f"mllama_dp_{transport}")))
    assert {"optimized_image_iter_1.png", "optimized_image_iter_5.png", "optimized_image_iter_final.bin", "state_iter_5.pt"} <= files


def _cross_rank(rank, world, port, tmp, out):
    from adversarialvlm_amd import crossattack_models
    _init(rank, world, port)
    names = ["synthetic/tiny-llava", "synthetic/tiny-mllama", "synthetic/tiny-qwen2vl"]
    eng, hist = crossattack_models.train(
        exp_name="cross3_dp", img_orig=_image(tmp, 70, 70), prompt="list", target_text="sure here it is", model_names=names,
        lr=1e-2, num_iterations=4, save_steps=2, batch_size=2, grad_accum_steps=1, scheduler_step_size=100, scheduler_gamma=0.9,
        restart_num=0, mask_type=None, mask_size=None, clamp_method="tanh", epsilon=0.4, sigma=1e-3, start_from_white=False,
        target_text_random=False, DPO_flag=False, model_weights=[0.5, 0.3, 0.2], use_gaussian_blur=True, gblur_kernel_size=5,
        use_local_crop=True, base_path=tmp, return_engine=True, replica_check_every=1, seed=4)
    out[rank] = (eng.p.cpu(), eng.image().cpu(), len(eng.plans), eng.plans[0].kind, len(hist))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_cross_model_one_family_per_rank(tmp_path):
    tmp = str(tmp_path)
    _image(tmp, 70, 70)
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_cross_rank, args=(3, _free_port(), tmp, out), nprocs=3, join=True)
    for r in (1, 2):
        assert torch.equal(out[0][0], out[r][0]) and torch.equal(out[0][1], out[r][1]), r
    assert [out[r][2] for r in range(3)] == [1, 1, 1]                      # one model (one plan) per rank
    assert len({out[r][3] for r in range(3)}) == 3                         # three different processor kinds
    assert out[0][4] == 4 and out[1][4] == 0 and out[2][4] == 0
    assert float(out[0][0].abs().max()) > 0 and bool(torch.isfinite(out[0][0]).all())
    files = set(os.listdir(os.path.join(tmp, "cross3_dp")))
    assert {"optimized_image_iter_1.png", "optimized_image_iter_3.png", "optimized_image_iter_final.bin", "state_iter_3.pt"} <= files
