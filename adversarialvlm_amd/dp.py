"""Data-parallel bookkeeping of the PGD loop (pure host logic, no device code).

The loop shards PROMPTS: every rank keeps a full replica of (p, m, v, x0, mask) and of the
model, processes `global_batch / world` prompts, and the only exchange is one all-reduce(sum)
of the image gradient (P_in * 4 bytes) per optimiser step over RCCL/xGMI.  Because the loss is
a mean over the batch and every rank has the same local batch, pre-scaling each rank's loss by
1/world turns the SUM into the global mean; the image-fit term is identical on all ranks so the
same pre-scale leaves it unchanged.  Cross-model runs group ranks by model: the pre-scale is
1/group_size, which averages inside a model's group and SUMS across models
(crossattack_models.py:391).  Replicas stay bit-identical because every rank applies the same
update to the same all-reduced gradient.
"""
import os

import torch


def init_from_env(backend="nccl"):
    """(rank, world, local_rank) from torchrun's environment; initialises the group if needed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend)
    return rank, world, local_rank


def shard_batch(global_batch, world):
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
    return global_batch // world


class Scales:
    """Loss / image-fit scale factors of one rank (see module docstring)."""

    def __init__(self, n_models, weights, accum, cross_mode, prescale):
        self.n_models, self.weights, self.accum = int(n_models), list(weights), int(accum)
        self.cross_mode, self.prescale = bool(cross_mode), float(prescale)

    def loss_scale(self, i=0):
        # single: (CE + img)/accum (attack_model.py:330); cross: w_i*CE_i, never divided (:369)
        w = self.weights[i] if self.cross_mode else self.weights[i] / self.accum
        return w * self.prescale

    def imgfit_scale(self):
        # cross: image_fit_loss added once per model (crossattack_models.py:369)
        n = float(self.n_models) if self.cross_mode else 1.0 / self.accum
        return n * self.prescale


class ReplicaError(RuntimeError):
    """The data-parallel replicas of (p, m, v) no longer hold the same bits, or a barrier of the peer
    exchange gave up waiting for a rank.  Raised on EVERY rank (the verdict is all-reduced)."""


def replica_digest(engine):
    """Three float64 sums over p, m and v: replicas that took the same steps agree on them bit for bit."""
    parts = []
    for t in (engine.p, engine.m, engine.v):
        d = t.double()
        parts += [d.sum(), d.abs().sum(), (d * d).sum()]
    return torch.stack(parts)


def check_replicas(engine, group=None):
    """Collective health check of a data-parallel run (call it on every rank at the same iteration):
    the sticky time-out word of the peer exchange (csrc/advx_comm.h: a wait that gives up lets its kernel
    go on with stale sums) and the digests of the replicas.  The sum the exchange replaces
    (crossattack_models.py:383-406) cannot silently drop a term; this is what keeps that property.
    -> None, or the reason as a string (the same on every rank)."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if engine.peer is not None and engine.peer.timed_out():
            return "a barrier of the peer exchange timed out"
        return None
    dev = engine.p.device
    on_cpu = dist.get_backend(group) == "gloo"
    lost = 1.0 if (engine.peer is not None and engine.peer.timed_out()) else 0.0
    dig = replica_digest(engine)
    lo = torch.cat([dig, torch.tensor([-lost], dtype=torch.float64, device=dev)])
    hi = lo.clone()
    if on_cpu:
        lo, hi = lo.cpu(), hi.cpu()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if float(lo[-1]) < 0.0:
        return "a barrier of the peer exchange timed out on at least one rank (a peer was late by more than its limit)"
    if not torch.equal(lo[:-1], hi[:-1]):
        return "the replicas of (p, m, v) differ between ranks"
    return None


def allreduce_image_grad_(grad, group=None):
    """The one exchange of a step."""
    torch.distributed.all_reduce(grad, op=torch.distributed.ReduceOp.SUM, group=group)
    return grad


# ----------------------------------------------------------------------- peer exchange
class _DevicePointer:
    """Lets torch alias comm-owned device memory (zero copy) through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class PeerExchange:
    """All-reduce(sum) of the image gradient by peer access over xGMI (advx_comm_* of
    include/advx.h): every rank exports one uncached exchange segment through HIP IPC, maps the
    others', and an all-reduce is one reduce kernel (meets the peers on entry, sums its slice of
    all send buffers in rank order, posts it to every recv buffer) plus the consumer's wait, on
    the caller's stream.  torch.distributed is used ONCE, to carry the 64-byte handles between processes."""

    MEM_NAMES = {1: "uncached", 2: "fine-grained", 3: "device"}

    def __init__(self, floats, device, group=None, mem_kind=0, timeout_s=5.0):
        import ctypes as C

        from . import _lib as L
        self.L = L
        self.group = group
        self.device = torch.device(device)
        dist = torch.distributed
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.floats = (int(floats) + 3) // 4 * 4
        self.timeout_s = float(timeout_s)
        self.handle = None
        lib = L.load()
        with torch.cuda.device(self.device):
            h = C.c_void_p()
            rc = lib.advx_comm_create(self.rank, self.world, self.floats, int(mem_kind), C.byref(h))
            err = None if rc == 0 else lib.advx_last_error().decode()
            blob = b""
            if rc == 0:
                buf = C.create_string_buffer(64)
                rc = lib.advx_comm_export(h, buf) if self.world > 1 else 0
                err = None if rc == 0 else lib.advx_last_error().decode()
                blob = buf.raw
            # every rank must learn whether EVERY rank got this far, or the survivors would wait
            # in all_gather for a peer that raised
            blobs = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(blobs, (err, blob), group=group)
            else:
                blobs = [(err, blob)]
            errs = [e for e, _ in blobs if e]
            if not errs and self.world > 1:
                allh = C.create_string_buffer(b"".join(b for _, b in blobs), 64 * self.world)
                rc = lib.advx_comm_connect(h, allh)
                err = None if rc == 0 else lib.advx_last_error().decode()
                flags = [None] * self.world
                dist.all_gather_object(flags, err, group=group)
                errs = [e for e in flags if e]
            if errs:
                if h:
                    lib.advx_comm_destroy(h)
                raise L.AdvxError("peer exchange unavailable: " + "; ".join(sorted(set(errs))))
            self.handle = h
            self.mem_kind = self.MEM_NAMES.get(lib.advx_comm_mem_kind(h), "?")
            self.send = torch.as_tensor(_DevicePointer(lib.advx_comm_send_buffer(h), self.floats), device=self.device)
            self.recv = torch.as_tensor(_DevicePointer(lib.advx_comm_recv_buffer(h), self.floats), device=self.device)

    def all_reduce(self, floats=None):
        """send -> (sum over ranks, rank order) -> recv of every rank, on the current stream."""
        L = self.L
        n = self.floats if floats is None else (int(floats) + 3) // 4 * 4
        L.check(L.load().advx_comm_allreduce(self.handle, n, self.timeout_s, L.current_stream(self.device)),
                "advx_comm_allreduce")
        return self.recv

    def all_reduce_(self, tensor):
        """In-place convenience for tensors that live elsewhere (two small device copies)."""
        n = tensor.numel()
        self.send[:n].copy_(tensor.reshape(-1))
        self.all_reduce(n)
        tensor.copy_(self.recv[:n].view_as(tensor))
        return tensor

    def timed_out(self):
        """True once any barrier of this rank gave up waiting (synchronises the stream)."""
        import ctypes as C
        L = self.L
        w = C.c_int32(0)
        L.check(L.load().advx_comm_status(self.handle, C.byref(w), L.current_stream(self.device)), "advx_comm_status")
        return bool(w.value)

    def close(self):
        if self.handle is not None:
            # the views die with the segment
            self.send = self.recv = None
            self.L.load().advx_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def probe_peer_exchange(ex, group=None, rounds=3, seed=1234):
    """Trust the peer path only after it has reproduced, bit for bit, the rank-ordered sum of
    known data on THIS machine; every rank returns the same verdict."""
    dist = torch.distributed
    ok = True
    n = ex.floats
    limit, ex.timeout_s = ex.timeout_s, min(ex.timeout_s, 2.0)    # a broken path should fail fast here
    for r in range(rounds):
        gen = torch.Generator().manual_seed(seed + 1000 * r + ex.rank)
        mine = torch.randn(n, generator=gen).to(ex.device)
        ex.send.copy_(mine)
        if ex.world > 1:
            dist.barrier(group=group)     # ranks enter the exchange together: the short limit is fair
        got = ex.all_reduce().clone()
        if ex.world > 1:
            # the reference sum comes over the host library (staged through the CPU for gloo)
            src = mine.cpu() if dist.get_backend(group) == "gloo" else mine
            parts = [torch.empty_like(src) for _ in range(ex.world)]
            dist.all_gather(parts, src, group=group)
            parts = [q.to(ex.device) for q in parts]
        else:
            parts = [mine]
        want = parts[0].clone()
        for q in parts[1:]:
            want += q
        ok = ok and bool(torch.equal(got, want))
    ok = ok and not ex.timed_out()
    ex.timeout_s = limit
    if ex.world > 1:
        flag = torch.tensor([1 if ok else 0], device=ex.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = bool(flag.item())
    return ok


def _peer_is_competitive(ex, group=None, iters=20, slack=1.25):
    """"auto" keeps the peer path only if it is not slower than the host library's all-reduce of
    the same tensor on THIS machine (worst rank decides; `slack` allows for the launches the
    fused one-call form saves).  All ranks return the same answer."""
    import time
    dist = torch.distributed
    buf = torch.zeros(ex.floats, dtype=torch.float32, device=ex.device)
    times = []
    for fn in (ex.all_reduce, lambda: dist.all_reduce(buf, group=group)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(ex.device)
        dist.barrier(group=group)
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(ex.device)
        times.append(time.perf_counter() - t0)
    t = torch.tensor(times, device=ex.device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    ex.peer_vs_host_seconds = (float(t[0]) / iters, float(t[1]) / iters)
    return bool(t[0] <= slack * t[1]) and not ex.timed_out()


#: what the last make_exchange() call on this rank decided and why (every rank holds the same verdict; PixelPGD keeps a
#: copy as `engine.exchange_report`, bench.py prints it in `config.exchange_report`)
_last_report = {}


def last_exchange_report():
    return dict(_last_report)


def _report(asked, chosen, reason, ex=None):
    import sys
    t = getattr(ex, "peer_vs_host_seconds", None) if ex is not None else _last_report.get("_timing")
    _last_report.clear()
    _last_report.update(asked=asked, chosen=chosen, reason=reason)
    if t is not None:
        _last_report["peer_us"], _last_report["host_us"] = round(t[0] * 1e6, 2), round(t[1] * 1e6, 2)
    if asked == "auto" and chosen == "host" and not reason.startswith("gloo"):
        # a fall-back must not be silent: the first multi-GPU run is also the first test of the peer path
        dist = torch.distributed
        if not dist.is_initialized() or dist.get_rank() == 0:
            print(f"[advx] exchange_transport=auto FELL BACK to the host library's all-reduce: {reason}", file=sys.stderr, flush=True)


def make_exchange(floats, device, group=None, transport="auto", timeout_s=5.0):
    """transport: "rccl" -> None (torch.distributed all-reduce); "peer" -> PeerExchange or an
    error; "auto" -> PeerExchange if it can be set up AND passes the probe, else None - said loudly on
    stderr and recorded in `last_exchange_report()` with the reason."""
    if transport not in ("auto", "peer", "rccl"):
        raise ValueError("transport must be auto, peer or rccl")
    if transport == "rccl":
        _report(transport, "host", "asked for")
        return None
    dist = torch.distributed
    if dist.is_initialized() and dist.get_backend(group) == "gloo" and transport == "auto":
        _report(transport, "host", "gloo group: CPU rehearsal groups keep the host all-reduce")
        return None
    # memory kinds in order of preference (include/advx.h ADVX_COMM_MEM_*): 0 = uncached if it can
    # be exported, 2 = fine-grained.  Every rank sees every rank's errors (they are gathered in the
    # constructor) and the probe's verdict is agreed, so all ranks walk this list in step.
    last_error = None
    for mem_kind in (0, 2):
        try:
            ex = PeerExchange(floats, device, group=group, mem_kind=mem_kind, timeout_s=timeout_s)
        except Exception as e:  # set-up failed on at least one rank
            last_error = e
            continue
        if probe_peer_exchange(ex, group):
            if transport == "auto" and ex.world > 1 and not _peer_is_competitive(ex, group):
                timing = ex.peer_vs_host_seconds
                _last_report["_timing"] = timing
                ex.close()
                _report(transport, "host", f"peer exchange correct but slower here ({timing[0] * 1e6:.1f} us vs "
                                           f"{timing[1] * 1e6:.1f} us per all-reduce)")
                return None
            _report(transport, "peer", f"self-test passed ({ex.mem_kind} segments)", ex)
            return ex
        ex.close()
        from . import _lib as L
        last_error = L.AdvxError("peer exchange failed its self-test on this machine")
    if transport == "peer":
        raise last_error
    _report(transport, "host", f"peer exchange unavailable: {last_error}")
    return None
