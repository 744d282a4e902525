#!/usr/bin/env python3
"""bench.py - the owned pixel-space hot path of the PGD loop on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
        (N > 1 without a launcher: this process starts the N ranks itself, before it touches the GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LLaVA-1.5 processor geometry, synthetic 336x336x3
image, 64-prompt batch PER GPU, tanh-clamp attack, in-kernel Philox noise, masked AdamW.
One step = fused forward (p -> 64 noisy pixel_values) -> [the VLM is not owned: its
backward is replaced by a fixed synthetic upstream gradient g ~ N(0,1) resident in HBM]
-> fused backward (sum over the batch, /std, image-fit term, tanh', mask, AdamW) ->
quantise-error statistics.  With N > 1 every rank owns 64 prompts (weak scaling) and the
shared image gradient (1.355 MB) is all-reduced once per step - over the library's peer
exchange (IPC segments over xGMI) when it passes its self-test and is not slower here, else
over RCCL, said loudly (`config.exchange_report`, `config.exchange_fell_back`); a `strong`
object (64 prompts in total on the same engine) is reported beside the weak `value`.

`value` = prompts * steps / s over all ranks.  The VLM forward/backward (PyTorch-ROCm,
~1e15 FLOP per 64-prompt step) is NOT inside this number - see DESIGN.md "Measurement".

Two timings, reported together:
 (A) the owned pixel path isolated = `value`, `ms_per_step`, `roofline`.  The K-step region is timed by a HIP event pair on the
     launch stream (`ms_per_step`; max over ranks) with the wall clock between two fences beside it (`ms_per_step_wall`; for
     a short K the two clocks get a region of K steps each, see timed()); no per-launch event sits inside the K steps - the per-kernel averages of `roofline` come from the 1000-step region timed right
     after it (`long_run`), where every stride-th launch carries its own event pair.
 (B) end to end = `e2e`: the same loop around a random-init LLaVA-1.5-7B architecture in fp16 (tools/e2e_bench.py), run in a
     FRESH CHILD PROCESS that is started and finished BEFORE this process makes its first GPU call, under a wall budget
     (`--e2e-budget`, 240 s); `e2e: {"skipped": reason}` when the budget is hit or the child fails - (A) is never lost to (B).
     Single GPU only (at N > 1 the self-launching parent runs it on one GPU before it starts the ranks).

Cache state.  The two B*P_out tensors of a step (86.7 MB each) fit the 256 MiB Infinity Cache, and a loop
that reuses ONE gradient tensor and ONE output block never leaves it; in the real loop a 7B VLM runs between
the two launches.  The timed region therefore ROTATES through `--ring` (default 8) distinct gradient tensors
and output blocks (1.39 GB, 5x the cache): every byte of both streams comes from / goes to DRAM
("cold", the headline `value` and `roofline.frac`).  The same K steps on one resident pair are timed right
after it and reported as `in_cache` (`roofline.frac_in_cache`).  `--scaling strong` keeps the GLOBAL batch at
64 prompts (64/N per rank, SURVEY 8(e) form A) instead of 64 per rank.
"""
import argparse
import collections
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# dmabuf IPC (RCCL and the peer exchange between processes need it on this driver); must be in the environment before the first
# HIP call of the process, whoever launched it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

H = W = 336
BATCH = 64
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def host_cores():
    """Cores this process may run on: the affinity mask, capped by a cgroup CPU quota if one is set
    (threads beyond the quota only add context switches).  -> (usable, os.cpu_count())."""
    total = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = total
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return usable, total


def cpu_baseline(seconds_budget=15.0):
    """The oracle (torch-CPU restatement of the reference's step, 'port') on ALL host cores this process
    may use; `cores` = the threads actually used."""
    from oracle.pgd import PGDOracle
    from oracle.processors import LlavaOracle
    threads, machine = host_cores()
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
    return dict(value=round(n * BATCH / dt, 2), unit="prompt-steps/s", cores=threads, host_cpu_count=machine, kind="port",
                sample=f"{n} steps of the same workload (336x336x3, B=64), {dt:.1f} s", steps_per_s=round(n / dt, 3))


def profile_stride(steps):
    """Every stride-th launch of a kernel carries HIP events: at least 10 timed launches per kernel whatever K >= 10 is
    (64 from K = 640 up) - a timed launch costs ~0.6 us, so not every one is timed."""
    return max(1, steps // (64 if steps >= 640 else 10))


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


def count_devices():
    """GPUs this process could use, counted by a short-lived CHILD: torch.cuda.device_count() stays off HIP only while
    torch's amdsmi route works - its fallback is hipGetDeviceCount, which would create a HIP context in a process that is
    about to start GPU children.  The child may initialise whatever it likes; this process stays clean."""
    try:
        res = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=300)
        return int(res.stdout.strip().splitlines()[-1]) if res.returncode == 0 and res.stdout.strip() else 0
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return 0


def run_e2e_child(args):
    """Timing (B) of SURVEY 8(d): tools/e2e_bench.py as a fresh child, before this process touches the GPU.
    -> the `e2e` object of the line.  Never raises: whatever goes wrong becomes {"skipped": reason}."""
    assert not torch.cuda.is_initialized(), "the e2e child must run before this process makes a GPU call"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "e2e_bench.py"), "--model", args.e2e_model, "--batch", str(args.e2e_batch),
           "--micro", str(args.e2e_micro), "--steps", str(args.e2e_steps), "--warmup", "1", "--image", str(args.e2e_image)]
    t0 = time.monotonic()
    # MIOpen's default find mode searches convolution kernels the first time it meets the patch embedding (measured on a fresh
    # box: first step 143 s, 7.7 s with FAST; the steps themselves 1.65 s either way for this architecture - profiles/r04)
    env = dict(os.environ)
    env.setdefault("MIOPEN_FIND_MODE", "FAST")
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, cwd=ROOT, env=env)     # stderr inherited: its progress lines
    except OSError as e:
        return {"skipped": f"could not start tools/e2e_bench.py: {e}"}
    try:
        out, _ = proc.communicate(timeout=args.e2e_budget)
    except subprocess.TimeoutExpired:
        proc.terminate()                                  # exactly the PID started above
        try:
            proc.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.communicate()
        return {"skipped": f"wall budget of {args.e2e_budget:.0f} s hit (child stopped)", "model": args.e2e_model}
    took = time.monotonic() - t0
    if proc.returncode != 0:
        return {"skipped": f"child exit status {proc.returncode}", "model": args.e2e_model, "wall_s": round(took, 1)}
    try:
        rec = json.loads([ln for ln in out.splitlines() if ln.strip().startswith("{")][-1])
    except (IndexError, ValueError) as e:
        return {"skipped": f"no JSON line from the child ({type(e).__name__})", "model": args.e2e_model}
    return {"s_per_step": rec["s_per_step"], "prompt_steps_per_s": rec["e2e_prompt_steps_per_s"],
            "first_step_s": rec.get("first_step_s"), "model": rec["model"], "batch": rec["batch"], "micro_batch": rec["micro_batch"],
            "steps": args.e2e_steps, "seq_len": rec.get("seq_len"), "dtype": rec.get("dtype"), "chain": rec.get("chain"),
            "approx_model_tflops": rec.get("approx_model_tflops"), "peak_mem_gb": rec.get("peak_mem_gb"), "wall_s": round(took, 1),
            "n_gpus": 1,
            "note": "random-init architecture (no weights offline), VLM forward + backward-to-pixels under PyTorch-ROCm: ~1e15 "
                    "FLOP of dense GEMM per 64-prompt step that this repository does not own; fresh child process, finished "
                    "before the pixel-path run started"}


def launch_ranks(n, argv, backend, limit_s, e2e=None):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks as fresh child
    processes - one per GPU, torchrun's environment variables, rendezvous on 127.0.0.1 - BEFORE this process makes
    any GPU call (a process that has initialised HIP must never fork/exec GPU work), wait for them and hand rank 0's
    single JSON line through on stdout.  -> exit status (0 only if every rank returned 0).
    The reference runs its models serially in one process (crossattack_models.py:352-391); this is the
    one-process-per-GPU form of SURVEY 8(e).  With fewer GPUs than ranks (a one-GPU box) the ranks fold onto the
    devices there are and the host collectives travel over gloo, since RCCL refuses two ranks on one device: a
    rehearsal of the entry point, said so in the line's `config.launcher`."""
    n_dev = count_devices()                    # asked of a child: this process makes no HIP call, whatever torch falls back to
    assert not torch.cuda.is_initialized(), "the launcher must not hold a HIP context when it starts the ranks"
    if n_dev == 0:
        print("bench.py needs a GPU (there is no CPU fallback)", file=sys.stderr)
        return 2
    extra, note = [], f"self-launched: {n} ranks on {min(n, n_dev)} GPU(s)"
    if n_dev < n and backend == "nccl":
        extra = ["--backend", "gloo"]
        note += f"; REHEARSAL: {n} ranks folded onto {n_dev} device(s), host collectives over gloo (RCCL refuses two ranks per device)"
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ADVX_BENCH_LAUNCHER=note)
        if e2e is not None and r == 0:
            env["ADVX_BENCH_E2E"] = json.dumps(e2e)     # (B), measured by the parent before the ranks: rank 0 prints it
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL and the peer exchange need it
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores()[0] // n)))
        # rank 0 inherits stdout (its one JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv) + extra, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.monotonic() + limit_s
    failed_at = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        now = time.monotonic()
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = now                   # a rank died: its peers get 15 s to notice, then they are stopped
        if now > deadline or (failed_at is not None and now - failed_at > 15.0):
            for p in procs:                   # exactly the PIDs started above
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            codes = [p.returncode for p in procs]
            print(f"bench.py: ranks stopped ({'time limit' if now > deadline else 'a rank failed'}), exit codes {codes}",
                  file=sys.stderr)
            return 1
        time.sleep(0.05)
    if any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    return 0


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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak = 64 prompts per GPU (default); strong = 64 prompts in total, 64/N per GPU")
    ap.add_argument("--cache", default="both", choices=["both", "cold", "hot"],
                    help="cold = rotate through --ring gradient / output buffers (beyond the Infinity Cache; the headline), "
                         "hot = one resident pair (the round-1 loop), both = cold timed first, hot reported as in_cache")
    ap.add_argument("--ring", type=int, default=8, help="distinct gradient tensors / output blocks of the cold loop")
    ap.add_argument("--graph", action="store_true",
                    help="supplementary figure: the cold loop as a captured hipGraph, on an engine of its own")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1: skip the supplementary strong-scaling region (64 prompts in total) after the weak one")
    ap.add_argument("--launch-timeout", type=float, default=1200.0,
                    help="self-launched ranks (no WORLD_SIZE in the environment) are stopped after this many seconds "
                         "(below the driver's own 1500 s limit, so that this one fires first)")
    ap.add_argument("--no-e2e", action="store_true", help="skip timing (B): the end-to-end child run around a random-init VLM")
    ap.add_argument("--e2e-budget", type=float, default=240.0, help="wall budget of the end-to-end child, seconds")
    ap.add_argument("--e2e-model", default="synthetic/llava-1.5-7b")
    ap.add_argument("--e2e-batch", type=int, default=BATCH)
    ap.add_argument("--e2e-micro", type=int, default=32)
    ap.add_argument("--e2e-steps", type=int, default=2)
    ap.add_argument("--e2e-image", type=int, default=H)
    ap.add_argument("--nt-loads", action="store_true", help="experiment: the pair's backward reads grad_out non-temporally")
    ap.add_argument("--xcd-map", type=int, default=None, choices=[0, 1, 2],
                    help="experiment (ADVX_TUNE_XCD_MAP): 0 = grids as in rounds 1-3, 1 = gx padded to a multiple of 8, "
                         "2 = padded + a contiguous range of column blocks per XCD (the library's default)")
    ap.add_argument("--bwd-xcd", type=int, default=None, choices=[0, 1],
                    help="experiment (ADVX_TUNE_BWD_XCD): XCD-aware block map in the pair's backward / the batch reductions (default 1)")
    ap.add_argument("--lean", action="store_true", help="experiment: the pair without the s / v / grad_p streams (ADVX_TUNE_PAIR_LEAN)")
    args = ap.parse_args()
    io_dtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.io]
    io_bytes = 4 if args.io == "f32" else 2

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's form, `python bench.py --gpus N`: no launcher around us - be the launcher.  Nothing above
        # this line has created a HIP context (importing torch and counting devices do not).
        e2e = None if args.no_e2e else run_e2e_child(args)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:] + ["--no-e2e"], args.backend, args.launch_timeout, e2e))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # (B) first: a fresh child, started and finished before this process makes any GPU call
    if "ADVX_BENCH_E2E" in os.environ:
        e2e = json.loads(os.environ["ADVX_BENCH_E2E"])
    elif args.no_e2e:
        e2e = {"skipped": "--no-e2e"}
    elif world > 1:
        e2e = {"skipped": "external launcher with N > 1: the ranks are already running (python bench.py --gpus N runs it first)"}
    else:
        e2e = run_e2e_child(args)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start {args.gpus} ranks (or none: bench.py launches them)")
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
    if args.nt_loads:
        from adversarialvlm_amd import _lib
        _lib.check(_lib.load().advx_set_tuning(2, 1), "advx_set_tuning")
    if args.xcd_map is not None:
        from adversarialvlm_amd import _lib
        _lib.check(_lib.load().advx_set_tuning(6, args.xcd_map), "advx_set_tuning")
    if args.bwd_xcd is not None:
        from adversarialvlm_amd import _lib
        _lib.check(_lib.load().advx_set_tuning(7, args.bwd_xcd), "advx_set_tuning")
    if args.lean:
        from adversarialvlm_amd import _lib
        _lib.check(_lib.load().advx_set_tuning(5, 1), "advx_set_tuning")

    if args.scaling == "strong":
        if BATCH % world:
            raise SystemExit(f"--scaling strong: {BATCH} prompts do not divide over {world} ranks")
        B = BATCH // world
    else:
        B = BATCH
    ring = max(1, args.ring) if args.cache != "hot" else 1
    gen = torch.Generator().manual_seed(0)
    x0 = torch.rand(3, H, W, generator=gen).to(dev)
    plan = Plan.llava(H, W)
    eng = PixelPGD(x0, [plan], epsilon=0.5, lr=1e-2, sigma0=1e-3, seed=1234 + rank, process_group=pg,
                   allow_fused=not args.no_fused, fused_mode=args.chain, force_exchange=args.force_exchange, io_dtype=io_dtype,
                   exchange_transport=args.exchange)
    if not eng.exchange:
        exchange = "none (single rank)"
    else:
        exchange = f"peer ({eng.peer.mem_kind} IPC segments)" if eng.peer is not None else f"{args.backend} all-reduce"
        rep = eng.exchange_report or {}
        if "peer_us" in rep:
            exchange += f"; at start-up peer {rep['peer_us']:.1f} us vs {args.backend} {rep['host_us']:.1f} us per all-reduce"
    # the synthetic upstream gradients: `ring` distinct tensors, every rank its own; each rank pre-scales its share
    # so that the SUM all-reduce is the DP average
    dgen = torch.Generator(device=dev).manual_seed(1 + rank)
    gs_ring = [(torch.randn(B, 3, H, W, generator=dgen, device=dev) * eng.loss_scale(0)).to(io_dtype) for _ in range(ring)]
    ring_bytes = (2 * ring + 1) * B * 3 * H * W * io_bytes      # `ring` gradient tensors + `ring`+1 output blocks

    held = collections.deque(maxlen=ring)        # keeps the last `ring` output blocks allocated
    if ring > 1:
        # reserve the ring's output blocks now (the caching allocator keeps them): no hipMalloc inside a timed
        # region however short the warm-up is
        pre = [torch.empty((B, 3 * H * W), dtype=io_dtype, device=dev) for _ in range(ring + 1)]
        del pre
    counter = [0]

    def step_cold():
        # the allocator can only hand out a block whose last use lies `ring` steps back
        held.append(eng.forward(B)[0])
        eng.backward_update([gs_ring[counter[0] % ring]])
        counter[0] += 1

    def step_hot():
        eng.forward(B)
        eng.backward_update([gs_ring[0]])

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    from adversarialvlm_amd import ops

    def make_gate(target_us=100.0):
        """A launch that keeps the stream busy for ~target_us WITHOUT touching memory (torch's spin kernel, calibrated here with an
        event pair): queued ahead of the first event of a short timed region - see timed().  None if the build has no such kernel."""
        spin = getattr(torch.cuda, "_sleep", None)
        if spin is None:
            return None
        try:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            spin(10000)
            torch.cuda.synchronize()
            probe = 400000
            e0.record()
            spin(probe)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3
            if not (us > 0):
                return None
            cycles = max(1000, min(int(probe * target_us / us), 50 * probe))
            return lambda: spin(cycles)
        except RuntimeError:
            return None

    gate = make_gate()

    def timed(step, steps=None, warmup=None, profile=False, split=False):
        """W warm-up steps, then exactly K steps between two fences.  -> (device seconds, wall seconds, per-kernel profile | None),
        each the MAX over ranks.  Device seconds = a HIP event pair recorded on the launch stream (torch's current stream IS the
        stream every advx_* launch of the engine goes to) right before the first and right after the last launch.
        split=True (the K-step region of `value`, K small): TWO regions of exactly K steps each.  The first is bracketed by the
        wall clock alone (fence, K steps, fence) -> wall seconds.  The second is the event-timed one: a ~100 us spin kernel
        (no memory traffic) is queued ahead of the first event, so that the K steps' launches are already waiting in the queue
        when the device reaches the event - the host's latency of the first launch after an idle fence (~20 us here) is not
        device time of the path; in the trainers the VLM's kernels keep the queue busy.  (A 256 MiB fill as the gate was tried
        first: its write-back slowed the steps behind it, 0.0417 vs 0.0390 ms.)  Otherwise one region carries both clocks.  profile=True: every stride-th launch of the B*P_out movers also carries its own start/stop event pair
        (advx_profile_*, hipExtLaunchKernelGGL; at least 10 per kernel, 64 from K = 640 up) - used for the 1000-step region
        only, so that no event packet sits inside the K steps that `value` is computed from (a timed launch costs +0.6 us)."""
        steps = args.steps if steps is None else steps
        for _ in range(args.warmup if warmup is None else warmup):
            step()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dt_wall = None
        if split:
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            fence()
            dt_wall = time.perf_counter() - t0
        fence()
        if profile:
            ops.profile_begin(max(steps, 1), stride=profile_stride(steps))
        t0 = time.perf_counter()
        if split and gate is not None:
            gate()
        ev0.record()
        for _ in range(steps):
            step()
        ev1.record()
        fence()
        if dt_wall is None:
            dt_wall = time.perf_counter() - t0
        dt_dev = ev0.elapsed_time(ev1) * 1e-3
        prof = ops.profile_end() if profile else None
        if world > 1:
            t = torch.tensor([dt_dev, dt_wall], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt_dev, dt_wall = float(t[0].item()), float(t[1].item())
        return dt_dev, dt_wall, prof

    runs = {}
    if args.cache in ("both", "cold"):
        runs["cold"] = timed(step_cold, split=args.steps < 500)
    held.clear()
    if args.cache in ("both", "hot"):
        runs["hot"] = timed(step_hot, split=args.steps < 500)
    main_key = "cold" if "cold" in runs else "hot"
    # The profiled region: the same loop as the headline over 1000 steps with per-launch event pairs on every stride-th
    # launch.  It supplies the per-kernel averages of `roofline` and is reported as `long_run`; `value` stays the K steps asked
    # for, timed above without any event packet between their launches.
    PROFILED_STEPS = 1000
    long_run = timed(step_cold if main_key == "cold" else step_hot, steps=PROFILED_STEPS, warmup=0, profile=True)
    held.clear()
    hot_prof = None
    if main_key == "cold" and "hot" in runs:
        hot_prof = timed(step_hot, steps=PROFILED_STEPS, warmup=0, profile=True)
    # Strong scaling beside the weak figure (N > 1): the same cold loop with the GLOBAL batch held at 64 prompts
    # (64/N per rank, SURVEY 8(e) form A) on the same engine and exchange - the batch is an argument of forward().
    strong_run = None
    if world > 1 and args.scaling == "weak" and not args.no_strong and "cold" in runs and BATCH % world == 0:
        Bs = BATCH // world
        gs_s = [(torch.randn(Bs, 3, H, W, generator=dgen, device=dev) * eng.loss_scale(0)).to(io_dtype) for _ in range(ring)]
        held_s = collections.deque(maxlen=ring)
        cnt_s = [0]

        def step_strong():
            held_s.append(eng.forward(Bs)[0])
            eng.backward_update([gs_s[cnt_s[0] % ring]])
            cnt_s[0] += 1

        sdt, sdt_wall, _ = timed(step_strong, steps=max(args.steps, 200), warmup=max(args.warmup, 10))
        strong_run = (sdt, sdt_wall, max(args.steps, 200), Bs)
        held_s.clear()
    # The same cold steps as a captured hipGraph (--graph; supplementary figure): `ring` steps of the pair - forward and
    # backward with their per-step scalars in device memory (advx_fused_*_sched) - captured once and replayed; no host work
    # per launch.  Single rank, pair chain only, on an engine of its OWN: whatever happens during capture, the engine the
    # replica / time-out checks below read is not involved.  Only "capture is not supported here" is reduced to a note.
    graph_run = None
    if args.graph and args.cache != "hot" and world == 1 and eng.mode == "pair" and not eng.exchange and ring % 2 == 0:
        geng = PixelPGD(x0, [Plan.llava(H, W)], epsilon=0.5, lr=1e-2, sigma0=1e-3, seed=4321, io_dtype=io_dtype, fused_mode="pair")
        geng.forward(B)
        geng.backward_update([gs_ring[0]])              # one eager step: the replayable form needs a prepared engine
        outs_g = [torch.empty((B, 3 * H * W), dtype=io_dtype, device=dev) for _ in range(ring)]
        replays = max(1, args.steps // ring)
        sched = geng.make_schedule(ring * (replays + 1))
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(graph):
                for k in range(ring):
                    geng.forward_sched(B, sched, outs_g[k])
                    geng.backward_update_sched(gs_ring[k], sched)
        except RuntimeError as e:
            if "captur" not in str(e).lower():
                raise
            graph_run = f"{type(e).__name__}: {e}"
        else:
            graph.replay()                                   # warm-up replay (also the first real `ring` steps)
            fence()
            t0 = time.perf_counter()
            for _ in range(replays):
                graph.replay()
            fence()
            gdt = time.perf_counter() - t0
            geng.advance(ring * (replays + 1))
            graph_run = (gdt, ring * replays)
        del geng
    dt, dt_wall, _ = runs[main_key]
    prof = long_run[2]
    # after the timed regions: the replicas of p must still hold the same bits on every rank, and no
    # barrier of the peer exchange may have timed out
    replicas_identical = None
    if world > 1:
        digest = torch.stack([eng.p.double().sum(), eng.p.double().abs().sum(), (eng.p.double() ** 2).sum()])
        lo, hi = digest.clone(), digest.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))
    exchange_timed_out = bool(eng.peer.timed_out()) if eng.peer is not None else None
    # ... and the state the steps left must be numbers: a kernel that went wrong over a long region shows here, not in the timing
    state_finite = bool(torch.isfinite(eng.p).all() and torch.isfinite(eng.m).all() and torch.isfinite(eng.v).all())
    if not state_finite:
        raise SystemExit("bench: the optimised tensor or the optimiser state holds non-finite values after the timed region")

    n_in = 3 * H * W
    bytes_fwd = io_bytes * B * n_in + 4 * 2 * n_in     # write B*P_out, read p,x0
    bytes_bwd = io_bytes * B * n_in + 4 * 8 * n_in     # read B*P_out; p,x0,mask,m,v in; p,m,v(+grad) out
    bytes_step = bytes_fwd + bytes_bwd                 # 4*(2*B*P_out + 10*P_in) at f32

    def dominant(prof_, step_ms):
        fwd_avg, bwd_avg, step_avg = prof_["fwd"][0], prof_["bwd"][0], prof_["step"][0]
        if eng.mode == "step":
            # one launch per step: backward of step t + forward of step t+1 in the same kernel
            return "k_fused_step_wave", bytes_step, step_avg
        if eng.mode == "pair":
            return (("k_fused_fwd", bytes_fwd, fwd_avg) if fwd_avg >= bwd_avg else ("k_fused_bwd", bytes_bwd, bwd_avg))
        # generic chain: k_emit / k_batch_reduce are not instrumented; price the whole step
        return "generic chain (whole step, HIP events)", bytes_step, step_ms

    def kernel_ms(prof_):
        return {"k_fused_fwd": round(prof_["fwd"][0], 5), "k_fused_bwd": round(prof_["bwd"][0], 5),
                "k_fused_step_wave": round(prof_["step"][0], 5)}

    def kernel_fracs(prof_):
        out = {}
        for name, nbytes, key in (("k_fused_fwd", bytes_fwd, "fwd"), ("k_fused_bwd", bytes_bwd, "bwd"),
                                  ("k_fused_step_wave", bytes_step, "step")):
            if prof_[key][1]:
                out[name] = round(nbytes / (prof_[key][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        return out

    steps_per_s = args.steps / dt
    ldt, ldt_wall, _ = long_run
    dom_name, dom_bytes, dom_ms = dominant(prof, ldt / PROFILED_STEPS * 1e3)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    traffic = traffic_source = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and args.scaling == "weak":
        # measured by separate rocprofv3 --pmc passes of `bench.py --cache <state>` (tools/pmc_summary.py);
        # keyed by cache state, kernel and boundary dtype so that it follows whichever kernel dominates here
        with open(pmc_path) as f:
            table = json.load(f)
        meta = table.get("_meta", {}).get(main_key)
        table = table.get(main_key, table)
        traffic = table.get(dom_name if args.io == "f32" else f"{dom_name}:{args.io}", {}).get("traffic_bytes_per_launch")
        if traffic is not None:
            traffic_source = ("stored rocprofv3 PMC pass (profiles/pmc_traffic.json), not a live counter of this run: " +
                              (f"commit {meta['commit']}, {meta['date_utc']} UTC" if meta else "round 2, unstamped"))
    # who ran this: one row per rank - the line itself answers "did the backend see N ranks on N devices?"
    topo = None
    if world > 1 or args.force_exchange:
        props = torch.cuda.get_device_properties(dev)
        bus = ":".join(f"{int(getattr(props, k)):02x}" for k in ("pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(props, k))
        mine = {"rank": rank, "device": torch.cuda.current_device(), "pci": bus or None,
                "uuid": str(getattr(props, "uuid", "")) or None, "host": socket.gethostname(), "pid": os.getpid()}
        rows = [None] * torch.distributed.get_world_size()
        torch.distributed.all_gather_object(rows, mine)
        keys = {(r["host"], r["pci"] or r["uuid"] or r["device"]) for r in rows}
        topo = {"ranks": rows, "backend_world_size": torch.distributed.get_world_size(),
                "backend_name": torch.distributed.get_backend(), "devices_distinct": len(keys) == len(rows)}
    if rank == 0:
        roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_source,
                    "cache_state": ("cold: every step reads a gradient tensor and writes an output block last touched "
                                    f"{ring} steps ago ({ring_bytes / 2**20:.0f} MiB in rotation vs the 256 MiB Infinity Cache)"
                                    if main_key == "cold" else "in_cache: one resident gradient tensor and output block"),
                    "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(dom_ms, 5),
                    "kernel_ms": kernel_ms(prof), "kernel_frac": kernel_fracs(prof),
                    "timed_launches": {k: v[1] for k, v in prof.items()},
                    "kernel_timing": f"per-launch HIP event pairs on the launch stream, every {profile_stride(PROFILED_STEPS)}th launch of the "
                                     f"{PROFILED_STEPS}-step region (`long_run`) that follows the K steps of `value`; none inside the K steps",
                    "step_algorithmic_bytes": bytes_step,
                    "step_frac_of_hbm_peak": round(bytes_step * steps_per_s / 1e9 / HBM_PEAK_GBS, 4)}
        line = {
            "metric": "adversarial PGD steps/sec x prompt-batch, LLaVA-1.5-7B pixel path at 1/2/4/8 MI355X",
            "value": round(steps_per_s * B * world, 1),
            "unit": "prompt-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5),
            "ms_per_step_wall": round(dt_wall / args.steps * 1e3, 5),
            "value_wall": round(args.steps / dt_wall * B * world, 1),
            "timing": "value / ms_per_step: HIP event pair on the launch stream around exactly K steps, max over ranks; "
                      "*_wall: time.perf_counter() around exactly K steps between two barrier + synchronize fences" +
                      ("; K < 500: two regions of K steps each, the event-timed one behind a queued ~100 us spin kernel so that the "
                       "host's first-launch latency after the idle fence is not counted as device time (bench.py: timed)"
                       if args.steps < 500 else "; one region carries both clocks"),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"LLaVA-1.5 tanh-clamp attack, 336x336x3 image, {B}-prompt batch per GPU, "
                                   "owned pixel path isolated (synthetic upstream gradient in HBM; VLM fwd/bwd not included)",
                       "prompts_per_gpu": B, "global_prompts": B * world, "image": [3, H, W],
                       "noise": "in-kernel Philox4x32-10", "optimizer": "AdamW", "parallelism": f"dp{world}",
                       "path": eng.mode, "boundary_dtype": args.io, "exchange": exchange,
                       "exchange_report": eng.exchange_report,
                       "exchange_fell_back": bool(eng.exchange_report and eng.exchange_report["asked"] == "auto"
                                                  and eng.exchange_report["chosen"] == "host"
                                                  and not eng.exchange_report["reason"].startswith("gloo")) if eng.exchange else None,
                       "backend": args.backend if eng.exchange else None,
                       "launcher": os.environ.get("ADVX_BENCH_LAUNCHER", "external (torchrun environment)" if world > 1 else "none"),
                       "cache_state": main_key, "ring": ring,
                       "replicas_identical": replicas_identical, "exchange_timed_out": exchange_timed_out, "state_finite": state_finite},
            "steps_per_s": round(steps_per_s, 1),
            "roofline": roofline,
        }
        if topo is not None:
            line["config"].update(topo)
            line["config"]["multi_gpu_hardware_evidence"] = (
                "this line" if topo["devices_distinct"] and topo["backend_name"] == "nccl" else
                "none: ranks share a device or the backend is not RCCL - a rehearsal of the entry point, not a scaling figure")
        if main_key == "cold" and "hot" in runs:
            hdt, hdt_wall, _ = runs["hot"]
            hprof = hot_prof[2]
            hname, hbytes, hms = dominant(hprof, hot_prof[0] / PROFILED_STEPS * 1e3)
            line["in_cache"] = {"value": round(args.steps / hdt * B * world, 1), "unit": "prompt-steps/s",
                                "ms_per_step": round(hdt / args.steps * 1e3, 5),
                                "ms_per_step_wall": round(hdt_wall / args.steps * 1e3, 5),
                                "steps_per_s": round(args.steps / hdt, 1),
                                "note": "same K steps on ONE resident gradient tensor / output block (both fit the "
                                        "256 MiB Infinity Cache): the round-1 loop, an upper estimate"}
            roofline["frac_cold"] = roofline["frac"]
            roofline["frac_in_cache"] = round(hbytes / (hms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roofline["kernel_in_cache"] = hname
            roofline["kernel_ms_in_cache"] = kernel_ms(hprof)
            roofline["kernel_frac_in_cache"] = kernel_fracs(hprof)
            roofline["step_frac_of_hbm_peak_in_cache"] = round(bytes_step * (args.steps / hdt) / 1e9 / HBM_PEAK_GBS, 4)
        line["long_run"] = {"value": round(PROFILED_STEPS / ldt * B * world, 1), "unit": "prompt-steps/s", "steps": PROFILED_STEPS,
                            "ms_per_step": round(ldt / PROFILED_STEPS * 1e3, 5),
                            "ms_per_step_wall": round(ldt_wall / PROFILED_STEPS * 1e3, 5), "kernel_ms": kernel_ms(prof),
                            "note": f"the same {main_key} loop over {PROFILED_STEPS} steps, timed after the K = {args.steps} above, "
                                    f"with per-launch event pairs on every {profile_stride(PROFILED_STEPS)}th launch (the source of "
                                    "roofline.kernel_ms); supplementary"}
        if strong_run is not None:
            sdt, sdt_wall, ssteps, Bs = strong_run
            line["strong"] = {"value": round(ssteps / sdt * Bs * world, 1), "unit": "prompt-steps/s", "scaling": "strong",
                              "steps": ssteps, "ms_per_step": round(sdt / ssteps * 1e3, 5),
                              "ms_per_step_wall": round(sdt_wall / ssteps * 1e3, 5), "steps_per_s": round(ssteps / sdt, 1),
                              "prompts_per_gpu": Bs, "global_prompts": Bs * world,
                              "note": "same engine and exchange, global batch held at 64 prompts; timed after the weak region"}
        if isinstance(graph_run, str):
            line["graph_replay"] = {"error": graph_run[:300]}
        elif graph_run is not None:
            gdt, gsteps = graph_run
            line["graph_replay"] = {"value": round(gsteps / gdt * B, 1), "unit": "prompt-steps/s", "steps": gsteps,
                                    "ms_per_step": round(gdt / gsteps * 1e3, 5),
                                    "note": f"the cold loop as a hipGraph: {ring} steps of the pair captured once (per-step scalars in "
                                            "device memory), replayed; supplementary - `value` is the eager loop"}
        line["e2e"] = e2e
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
