#!/usr/bin/env python3
"""bench.py - the owned pixel-space hot path of the PGD loop on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LLaVA-1.5 processor geometry, synthetic 336x336x3
image, 64-prompt batch PER GPU, tanh-clamp attack, in-kernel Philox noise, masked AdamW.
One step = fused forward (p -> 64 noisy pixel_values) -> [the VLM is not owned: its
backward is replaced by a fixed synthetic upstream gradient g ~ N(0,1) resident in HBM]
-> fused backward (sum over the batch, /std, image-fit term, tanh', mask, AdamW) ->
quantise-error statistics.  With N > 1 every rank owns 64 prompts (weak scaling) and the
shared image gradient (1.355 MB) is all-reduced once per step over RCCL.

`value` = prompts * steps / s over all ranks.  The VLM forward/backward (PyTorch-ROCm,
~1e15 FLOP per 64-prompt step) is NOT inside this number - see DESIGN.md "Measurement".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

H = W = 336
BATCH = 64
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(seconds_budget=15.0):
    """The oracle (torch-CPU restatement of the reference's step, 'port') on host cores."""
    from oracle.pgd import PGDOracle
    from oracle.processors import LlavaOracle
    threads = max(1, min(os.cpu_count() or 1, 16))
    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(0)
    x0 = torch.rand(3, H, W, generator=gen)
    g = torch.randn(BATCH, 3, H, W, generator=gen)
    ora = PGDOracle(x0, [LlavaOracle(H, W)], lr=1e-2)

    def step():
        z = torch.randn(BATCH, 3, H, W)          # the reference draws randn_like every step
        ora.forward(BATCH, [z])
        ora.backward_update([g])

    for _ in range(2):
        step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds_budget or n >= 200:
            break
    return dict(value=round(n * BATCH / dt, 2), unit="prompt-steps/s", cores=threads, kind="port",
                sample=f"{n} steps of the same workload (336x336x3, B=64), {dt:.1f} s", steps_per_s=round(n / dt, 3))


class stdout_to_stderr:
    """RCCL prints a version banner on fd 1 when a communicator is created; keep stdout for the
    single JSON line by pointing fd 1 at stderr while the process group comes up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="time the generic (unfused) kernel chain instead")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                       "the multi-rank path on one GPU)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="rehearsal: run the data-parallel chain (RCCL all-reduce included) with a single rank")
    ap.add_argument("--exchange", default="auto", choices=["auto", "peer", "rccl"],
                    help="transport of the per-step image-gradient all-reduce: peer = IPC-mapped segments over xGMI "
                         "(advx_comm_*), rccl = torch.distributed, auto = peer if it passes its self-test here")
    ap.add_argument("--chain", default="auto", choices=["auto", "pair", "step"],
                    help="fused chain: step = one launch per step (single GPU), pair = two launches")
    ap.add_argument("--io", default="f32", choices=["f32", "f16", "bf16"],
                    help="dtype of pixel_values / their gradient at the VLM boundary (f32 = the reference's own "
                         "boundary and the headline; f16/bf16 = emit in the model's dtype, pair chain only)")
    args = ap.parse_args()
    io_dtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.io]
    io_bytes = 4 if args.io == "f32" else 2

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    local_rank = local_rank % max(1, torch.cuda.device_count())      # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            if args.backend == "nccl":
                torch.distributed.init_process_group("nccl", device_id=dev)
            else:
                torch.distributed.init_process_group(args.backend)
            pg = torch.distributed.group.WORLD
            warm = torch.ones(1, device=dev)
            torch.distributed.all_reduce(warm)          # communicator creation happens here
            torch.cuda.synchronize()

    from adversarialvlm_amd.build import build_library
    if rank == 0:
        build_library()              # no-op when __graft_entry__.build() has run; one builder, not N
    if world > 1:
        torch.distributed.barrier()
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan

    gen = torch.Generator().manual_seed(0)
    x0 = torch.rand(3, H, W, generator=gen).to(dev)
    g = torch.randn(BATCH, 3, H, W, generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    plan = Plan.llava(H, W)
    eng = PixelPGD(x0, [plan], epsilon=0.5, lr=1e-2, sigma0=1e-3, seed=1234 + rank, process_group=pg,
                   allow_fused=not args.no_fused, fused_mode=args.chain, force_exchange=args.force_exchange, io_dtype=io_dtype,
                   exchange_transport=args.exchange)
    if not eng.exchange:
        exchange = "none (single rank)"
    else:
        exchange = f"peer ({eng.peer.mem_kind} IPC segments)" if eng.peer is not None else f"{args.backend} all-reduce"
        timing = getattr(eng.peer, "peer_vs_host_seconds", None)
        if timing is not None:
            exchange += f"; at start-up peer {timing[0] * 1e6:.1f} us vs {args.backend} {timing[1] * 1e6:.1f} us per all-reduce"
    # every rank pre-scales its share so that the SUM all-reduce is the DP average
    gs = (g * eng.loss_scale(0)).to(io_dtype)

    def step():
        eng.forward(BATCH)
        eng.backward_update([gs])

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    from adversarialvlm_amd import ops
    fence()
    # per-kernel device time: every launch of the B*P_out movers inside the timed region carries
    # its own start/stop HIP event pair on the launch stream (advx_profile_*, hipExtLaunchKernelGGL)
    ops.profile_begin(max(args.steps, 1), stride=16)      # every 16th launch: the loop stays unperturbed
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = ops.profile_end()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    fwd_avg, bwd_avg, step_avg = prof["fwd"][0], prof["bwd"][0], prof["step"][0]
    # after the timed region: the replicas of p must still hold the same bits on every rank, and no
    # barrier of the peer exchange may have timed out
    replicas_identical = None
    if world > 1:
        digest = torch.stack([eng.p.double().sum(), eng.p.double().abs().sum(), (eng.p.double() ** 2).sum()])
        lo, hi = digest.clone(), digest.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))
    exchange_timed_out = bool(eng.peer.timed_out()) if eng.peer is not None else None

    n_in = 3 * H * W
    bytes_fwd = io_bytes * BATCH * n_in + 4 * 2 * n_in     # write B*P_out, read p,x0
    bytes_bwd = io_bytes * BATCH * n_in + 4 * 8 * n_in     # read B*P_out; p,x0,mask,m,v in; p,m,v(+grad) out
    bytes_step = bytes_fwd + bytes_bwd                     # 4*(2*B*P_out + 10*P_in) at f32
    steps_per_s = args.steps / dt
    if eng.mode == "step":
        # one launch per step: backward of step t + forward of step t+1 in the same kernel
        dom_name, dom_bytes, dom_ms = "k_fused_step_wave", bytes_step, step_avg
    elif eng.mode == "pair":
        dom_name, dom_bytes, dom_ms = (("k_fused_fwd", bytes_fwd, fwd_avg) if fwd_avg >= bwd_avg
                                       else ("k_fused_bwd", bytes_bwd, bwd_avg))
    else:
        # generic chain: k_emit / k_batch_reduce are not instrumented; price the whole step
        dom_name, dom_bytes, dom_ms = "generic chain (whole step, wall)", bytes_step, dt / args.steps * 1e3
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            traffic = json.load(f).get(dom_name if args.io == "f32" else f"{dom_name}:{args.io}", {}) \
                .get("traffic_bytes_per_launch")
    if rank == 0:
        line = {
            "metric": "adversarial PGD steps/sec x prompt-batch, LLaVA-1.5-7B pixel path at 1/2/4/8 MI355X",
            "value": round(steps_per_s * BATCH * world, 1),
            "unit": "prompt-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LLaVA-1.5 tanh-clamp attack, 336x336x3 image, 64-prompt batch per GPU, "
                                   "owned pixel path isolated (synthetic upstream gradient in HBM; VLM fwd/bwd not included)",
                       "prompts_per_gpu": BATCH, "global_prompts": BATCH * world, "image": [3, H, W],
                       "noise": "in-kernel Philox4x32-10", "optimizer": "AdamW", "parallelism": f"dp{world}",
                       "path": eng.mode, "boundary_dtype": args.io, "exchange": exchange,
                       "replicas_identical": replicas_identical, "exchange_timed_out": exchange_timed_out},
            "steps_per_s": round(steps_per_s, 1),
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(dom_ms, 5),
                         "kernel_ms": {"k_fused_fwd": round(fwd_avg, 5), "k_fused_bwd": round(bwd_avg, 5),
                                       "k_fused_step_wave": round(step_avg, 5)},
                         "timed_launches": {k: v[1] for k, v in prof.items()},
                         "step_algorithmic_bytes": bytes_step,
                         "step_frac_of_hbm_peak": round(bytes_step * steps_per_s / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
