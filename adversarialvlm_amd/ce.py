"""Suffix-only cross entropy.

The reference computes the whole [B, S, V] logits tensor and slices the target positions out
of it (attack_model.py:324-328, llavaprocessor.py:73-78).  Here the VLM is asked for its last
`suffix_len + 1` positions only (`logits_to_keep`), and the log-softmax + NLL of the supervised
positions - forward and backward - run in libadvx_hip.so (advx_ce_fwd / advx_ce_bwd)."""
import ctypes as C

import torch

from . import _lib as L
from .ops import io_code


def _raw(t):
    # strided [B, K, V] views are addressed through their strides: hand over the base pointer
    return C.c_void_p(t.data_ptr())


class _SuffixCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, T):
        # logits [B, K, V] (any strides along B and K, unit stride along V); targets [B, T] int64
        if not logits.is_cuda:
            raise L.AdvxError("suffix_cross_entropy needs logits on a ROCm device (there is no CPU fallback)")
        if logits.dim() != 3 or logits.stride(2) != 1:
            raise L.AdvxError("logits must be [B, K, V] with a contiguous vocabulary axis")
        B, K, V = logits.shape
        T = int(T)
        if not (1 <= T <= K) or tuple(targets.shape) != (B, T):
            raise L.AdvxError("targets must be [B, T] with T <= K")
        targets = targets.to(device=logits.device, dtype=torch.int64).contiguous()
        dev = logits.device
        row_loss = torch.empty(B * T, dtype=torch.float32, device=dev)
        row_lse = torch.empty(B * T, dtype=torch.float32, device=dev)
        mean_n = torch.empty(2, dtype=torch.float32, device=dev)
        io = io_code(logits.dtype)
        scratch = torch.empty(int(L.load().advx_ce_scratch_floats(B * T, V, io)), dtype=torch.float32, device=dev)
        L.check(L.load().advx_ce_fwd(_raw(logits), io, logits.stride(0), logits.stride(1), T,
                                     L.ptr(targets), B * T, V, L.ptr(row_loss), L.ptr(row_lse), L.ptr(mean_n),
                                     L.ptr(scratch), L.current_stream(dev)), "advx_ce_fwd")
        ctx.save_for_backward(logits, targets, row_lse, mean_n)
        ctx.T = T
        return mean_n[0].clone()

    @staticmethod
    def backward(ctx, grad_loss):
        logits, targets, row_lse, mean_n = ctx.saved_tensors
        B, K, V = logits.shape
        up = grad_loss.to(dtype=torch.float32).reshape(1).contiguous()
        grad = torch.empty_strided(logits.shape, logits.stride(), dtype=logits.dtype, device=logits.device)
        L.check(L.load().advx_ce_bwd(_raw(logits), io_code(logits.dtype), logits.stride(0), logits.stride(1), ctx.T, K,
                                     L.ptr(targets), B * ctx.T, V, L.ptr(row_lse), L.ptr(mean_n), L.ptr(up), _raw(grad),
                                     L.current_stream(logits.device)), "advx_ce_bwd")
        return grad, None, None


def suffix_cross_entropy(logits, targets):
    """mean_r -log softmax(logits[b, t, :])[targets[b, t]] over the first T = targets.shape[1]
    positions of logits [B, K, V] (K >= T); targets outside [0, V) are ignored."""
    return _SuffixCE.apply(logits, targets, targets.shape[1])
