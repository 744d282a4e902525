"""CPU tier, world_size 2 over gloo: the data-parallel exchange of the loop.

Two processes each hold a replica of the image state and HALF of the prompt batch, pre-scale
their loss with the product's `dp.Scales`, all-reduce(sum) the image gradient with the
product's `dp.allreduce_image_grad_`, and must end every step bit-identical to each other and
(to rounding) equal to one process that saw the whole batch.  The per-rank compute is the CPU
oracle - the exchange logic is what is under test here; the HIP engine uses the same helpers."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from adversarialvlm_amd import dp
from oracle import pixel_ops as P
from oracle.pgd import PGDOracle
from oracle.processors import LlavaOracle, Qwen2VLOracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, _ = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    gen = torch.Generator().manual_seed(0)
    x0 = torch.rand(3, 24, 24, generator=gen)
    B = 4
    z = [torch.randn(B, 3, 24, 24, generator=gen) for _ in range(3)]
    g = [torch.randn(B, 3, 24, 24, generator=gen) * 0.01 for _ in range(3)]
    if mode == "dp":
        # both ranks run the same model on half of the batch
        local = dp.shard_batch(B, world)
        sl = slice(rank * local, (rank + 1) * local)
        scales = dp.Scales(1, [1.0], accum=1, cross_mode=False, prescale=1.0 / world)
        ora = PGDOracle(x0, [LlavaOracle(24, 24)], lr=1e-2)
        for t in range(3):
            ora.forward(local, [z[t][sl]])
            # emulate backward_update with the DP scales: loss = sum(pv*g) * loss_scale, img * imgfit_scale
            gr = ora._graph
            # the model loss is a MEAN over the local batch (F.cross_entropy default), like the CE
            total = (gr["pvs"][0] * g[t][sl]).sum() / local * scales.loss_scale(0) + gr["img_loss"] * scales.imgfit_scale()
            total.backward()
            dp.allreduce_image_grad_(ora.p.grad)
            with torch.no_grad():
                ora.p.grad.mul_(ora.mask)
            ora.opt.step(); ora.opt.zero_grad(); ora.sched.step()
            ora.sigma = P.quantise_error_stats(gr["s"].detach())[0]     # identical on every rank
            ora._graph = None
        out[rank] = ora.p.detach().clone()
    else:
        # cross-model: rank 0 = "model A" (LLaVA layout), rank 1 = "model B" (Qwen layout);
        # weighted SUM across the two groups of size 1, image-fit once per model
        procs = [LlavaOracle(24, 24), Qwen2VLOracle(min_pixels=28 * 28, max_pixels=28 * 28 * 4)]
        weights = [0.2, 1.6]
        scales = dp.Scales(1, [weights[rank]], accum=1, cross_mode=True, prescale=1.0)
        ora = PGDOracle(x0, [procs[rank]], lr=1e-2)
        gen2 = torch.Generator().manual_seed(5)
        for t in range(2):
            pv = ora.forward(2)[0]
            ups = [torch.randn(2 * s[0], *s[1:], generator=gen2) * 0.01 for s in [(1, 3, 24, 24), (4, 1176)]]
            gr = ora._graph
            total = (gr["pvs"][0] * ups[rank]).sum() * scales.loss_scale(0) + gr["img_loss"] * scales.imgfit_scale()
            total.backward()
            dp.allreduce_image_grad_(ora.p.grad)
            ora.opt.step(); ora.opt.zero_grad(); ora.sched.step()
            ora.sigma = P.quantise_error_stats(gr["s"].detach())[0]
            ora._graph = None
        out[rank] = ora.p.detach().clone()
    torch.distributed.destroy_process_group()


def _spawn(mode):
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, mode, out), nprocs=2, join=True)
    return out[0], out[1]


@pytest.mark.timeout(120)
def test_dp_two_ranks_equal_single_process_big_batch():
    p0, p1 = _spawn("dp")
    assert torch.equal(p0, p1)                         # replicas stay bit-identical
    gen = torch.Generator().manual_seed(0)
    x0 = torch.rand(3, 24, 24, generator=gen)
    z = [torch.randn(4, 3, 24, 24, generator=gen) for _ in range(3)]
    g = [torch.randn(4, 3, 24, 24, generator=gen) * 0.01 for _ in range(3)]
    ref = PGDOracle(x0, [LlavaOracle(24, 24)], lr=1e-2)
    for t in range(3):
        ref.forward(4, [z[t]])
        ref.backward_update([g[t] / 4])            # mean over the whole batch of 4
    # sigma differs per step only through the shared image -> identical; sums are reordered
    assert float((p0 - ref.p.detach()).norm() / ref.p.detach().norm()) < 1e-5


@pytest.mark.timeout(120)
def test_cross_model_groups_sum_like_single_process():
    p0, p1 = _spawn("cross")
    assert torch.equal(p0, p1)
    x0 = torch.rand(3, 24, 24, generator=torch.Generator().manual_seed(0))
    ref = PGDOracle(x0, [LlavaOracle(24, 24), Qwen2VLOracle(min_pixels=28 * 28, max_pixels=28 * 28 * 4)], lr=1e-2,
                    model_weights=[0.2, 1.6], cross_mode=True)
    gen2 = torch.Generator().manual_seed(5)
    for t in range(2):
        ref.forward(2)
        ups = [torch.randn(2 * s[0], *s[1:], generator=gen2) * 0.01 for s in [(1, 3, 24, 24), (4, 1176)]]
        ref.backward_update(ups)
    assert float((p0 - ref.p.detach()).norm() / ref.p.detach().norm()) < 1e-5


def _check_rank(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from types import SimpleNamespace
    dp.init_from_env("gloo")
    gen = torch.Generator().manual_seed(1)
    eng = SimpleNamespace(p=torch.randn(3, 8, 8, generator=gen), m=torch.randn(3, 8, 8, generator=gen),
                          v=torch.rand(3, 8, 8, generator=gen), peer=None)
    res = [dp.check_replicas(eng)]
    if rank == 1:
        eng.m[2, 3, 4] += 1e-6              # one element of one moment, on one rank
    res.append(dp.check_replicas(eng))
    # a peer exchange whose sticky time-out word is set on ONE rank ends the run on every rank
    eng.peer = SimpleNamespace(timed_out=lambda: rank == 0)
    res.append(dp.check_replicas(eng))
    out[rank] = res
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_replica_check_is_collective_and_agreed():
    """dp.check_replicas: identical replicas pass; one differing element or one rank's time-out word is
    reported with the same reason on EVERY rank (the trainers turn it into dp.ReplicaError / exit 3)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_check_rank, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == out[1]
    ok, differ, lost = out[0]
    assert ok is None and "differ" in differ and "timed out" in lost
