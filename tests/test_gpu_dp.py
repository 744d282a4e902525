"""GPU tier: the data-parallel chain of the HIP engine (fused_bwd grad-only -> all-reduce ->
fused_update -> prepared forward) with TWO ranks sharing the one GPU of the test box.  RCCL
refuses two ranks on one device, so the exchange runs over gloo here; the engine code is the
same one the 2/4/8-GPU benchmark runs over RCCL."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, chain, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo")
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    H = 64
    x0 = torch.rand(3, H, H, generator=gen)
    B = 8
    zs = [torch.randn(B, 3, H, H, generator=gen) for _ in range(3)]
    gs = [torch.randn(B, 3, H, H, generator=gen) * 0.01 for _ in range(3)]
    local = B // world
    sl = slice(rank * local, (rank + 1) * local)
    kw = dict(allow_fused=False) if chain == "generic" else dict(fused_mode="pair")
    eng = PixelPGD(x0.to(dev), [Plan.llava(H, H, H, H)], lr=1e-2, process_group=torch.distributed.group.WORLD, **kw)
    assert eng.world == world
    for t in range(3):
        eng.forward(local, [zs[t][sl].to(dev)])
        # mean-type loss over the LOCAL batch, then the engine's DP pre-scale
        eng.backward_update([gs[t][sl].to(dev) / local * eng.loss_scale(0)])
    st = eng.stats_dict()
    out[rank] = (eng.p.cpu(), st["grad_norm"], st["sigma_next"])
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("chain", ["pair", "generic"])
def test_two_ranks_match_single_process(chain):
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_rank, args=(2, port, chain, out), nprocs=2, join=True)
    (p0, n0, s0), (p1, n1, s1) = out[0], out[1]
    assert torch.equal(p0, p1) and n0 == n1 and s0 == s1        # replicas bit-identical
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    x0 = torch.rand(3, 64, 64, generator=gen)
    zs = [torch.randn(8, 3, 64, 64, generator=gen) for _ in range(3)]
    gs = [torch.randn(8, 3, 64, 64, generator=gen) * 0.01 for _ in range(3)]
    ref = PixelPGD(x0.to(dev), [Plan.llava(64, 64, 64, 64)], lr=1e-2)
    for t in range(3):
        ref.forward(8, [zs[t].to(dev)])
        ref.backward_update([gs[t].to(dev) / 8])
    rst = ref.stats_dict()
    err = float((p0 - ref.p.cpu()).norm() / ref.p.cpu().norm())
    assert err < 1e-5, err
    assert n0 == pytest.approx(rst["grad_norm"], rel=1e-5) and s0 == pytest.approx(rst["sigma_next"], rel=1e-6)


def _rccl_single(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    torch.distributed.init_process_group("nccl", device_id=dev)
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(0)
    x0 = torch.rand(3, 64, 64, generator=gen).to(dev)
    zs = [torch.randn(4, 3, 64, 64, generator=gen).to(dev) for _ in range(3)]
    gs = [(torch.randn(4, 3, 64, 64, generator=gen) * 0.01).to(dev) for _ in range(3)]
    res = []
    for force in (True, False):
        eng = PixelPGD(x0, [Plan.llava(64, 64, 64, 64)], process_group=torch.distributed.group.WORLD, force_exchange=force)
        for t in range(3):
            eng.forward(4, [zs[t]])
            eng.backward_update([gs[t]])
        res.append((eng.p.cpu(), eng.stats_dict()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    out["same_p"] = bool(torch.equal(res[0][0], res[1][0]))
    out["stats"] = (res[0][1], res[1][1])


@pytest.mark.timeout(300)
def test_rccl_exchange_chain_single_rank():
    """backend "nccl" IS RCCL on ROCm: a one-rank group drives the real data-parallel chain
    (fused_bwd grad-only -> RCCL all-reduce -> fused_update -> prepared forward) on the GPU and
    must reproduce the single-launch update bit for bit."""
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_rccl_single, args=(_free_port(), out), nprocs=1, join=True)
    assert out["same_p"]
    a, b = out["stats"]
    for k in a:
        assert a[k] == pytest.approx(b[k], rel=1e-6, abs=1e-12), k
