#!/usr/bin/env python3
"""Device time of the suffix-only cross entropy (advx_ce_fwd / advx_ce_bwd) against the bytes it has to move
(development tool):   python tools/ce_bench.py
Forward reads the T supervised rows of [B, K, V] once; backward reads them again and writes all K rows of the gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adversarialvlm_amd.ce import suffix_cross_entropy  # noqa: E402

CASES = [  # (name, B, K, T, V, dtype)
    ("llava-1.5 f16", 64, 9, 8, 32064, torch.float16),
    ("llava-1.5 f16 long target", 64, 33, 32, 32064, torch.float16),
    ("llama-3.2-vision bf16", 64, 9, 8, 128256, torch.bfloat16),
    ("qwen2-vl bf16", 64, 9, 8, 152064, torch.bfloat16),
    ("llava f32", 16, 9, 8, 32064, torch.float32),
    ("one prompt f16", 1, 9, 8, 32064, torch.float16),
]


def abi_times(x, tg, B, K, T, V):
    """The two C-ABI calls on their own: HIP events around 50 back-to-back calls each (no autograd, no allocations)."""
    import ctypes as C
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd.ops import io_code
    lib, dev, io = L.load(), x.device, io_code(x.dtype)
    row_loss = torch.empty(B * T, dtype=torch.float32, device=dev)
    row_lse = torch.empty_like(row_loss)
    mean_n = torch.empty(2, dtype=torch.float32, device=dev)
    scratch = torch.empty(int(lib.advx_ce_scratch_floats(B * T, V, io)), dtype=torch.float32, device=dev)
    up = torch.ones(1, dtype=torch.float32, device=dev)
    grad = torch.empty_like(x)
    st = L.current_stream(dev)
    raw = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def fwd():
        L.check(lib.advx_ce_fwd(raw(x), io, x.stride(0), x.stride(1), T, L.ptr(tg), B * T, V, L.ptr(row_loss), L.ptr(row_lse),
                                L.ptr(mean_n), L.ptr(scratch), st), "advx_ce_fwd")

    def bwd():
        L.check(lib.advx_ce_bwd(raw(x), io, x.stride(0), x.stride(1), T, K, L.ptr(tg), B * T, V, L.ptr(row_lse), L.ptr(mean_n),
                                L.ptr(up), raw(grad), st), "advx_ce_bwd")
    out = []
    for fn in (fwd, bwd):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50 * 1e3)
    return out


def main():
    dev = torch.device("cuda:0")
    for name, B, K, T, V, dt in CASES:
        x = (torch.randn(B, K, V, device=dev) * 3.0).to(dt).requires_grad_(True)
        tg = torch.randint(0, V, (B, T), device=dev)
        es = x.element_size()
        fwd_bytes, bwd_bytes = B * T * V * es, B * T * V * es + B * K * V * es
        for _ in range(3):
            loss = suffix_cross_entropy(x, tg)
            loss.backward()
            x.grad = None
        reps = 30
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(reps):
            ev[0].record()
            loss = suffix_cross_entropy(x, tg)
            ev[1].record()
            loss.backward()
            ev[2].record()
            torch.cuda.synchronize()
            tf += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
            x.grad = None
        tf, tb = tf / reps * 1e3, tb / reps * 1e3
        kf, kb = abi_times(x.detach(), tg, B, K, T, V)
        print(f"{name:28s} B={B:3d} K={K:2d} T={T:2d} V={V:6d}: advx_ce_fwd {kf:6.1f} us = {fwd_bytes / kf / 1e6:5.2f} TB/s of {fwd_bytes / 1e6:6.1f} MB"
              f"   advx_ce_bwd {kb:6.1f} us = {bwd_bytes / kb / 1e6:5.2f} TB/s of {bwd_bytes / 1e6:6.1f} MB   [C ABI, events over 50 calls]")
        print(f"{name:28s} B={B:3d} K={K:2d} T={T:2d} V={V:6d}: fwd {tf:7.1f} us ({fwd_bytes / tf / 1e6:5.2f} TB/s of {fwd_bytes / 1e6:6.1f} MB)"
              f"   bwd {tb:7.1f} us ({bwd_bytes / tb / 1e6:5.2f} TB/s of {bwd_bytes / 1e6:6.1f} MB)   [events around the autograd calls]")


if __name__ == "__main__":
    main()
