"""torch-facing wrappers over the C ABI (include/advx.h).

torch is plumbing here: it owns device memory and streams; every arithmetic operation of
the pixel path runs in libadvx_hip.so.  Nothing in this file computes on the CPU.
"""
import ctypes as C

import torch

from . import _lib as L


class generic_kernels:
    """Context manager (tests): every call inside takes the general kernels - run-time blur radius, one launch
    per operation - instead of the specialised / merged ones.  Results must be bit-identical."""

    def __enter__(self):
        L.check(L.load().advx_set_tuning(L.TUNE_GENERIC_KERNELS, 1), "advx_set_tuning")

    def __exit__(self, *exc):
        L.check(L.load().advx_set_tuning(L.TUNE_GENERIC_KERNELS, 0), "advx_set_tuning")


class separate_crop:
    """Context manager (tests, measurements): a crop window is never composed with the plan's stage 0 - the window is resized
    into `argument` and the plan resamples that (two launches each way, bit-identical to the unfused kernels)."""

    def __enter__(self):
        L.check(L.load().advx_set_tuning(L.TUNE_SEPARATE_CROP, 1), "advx_set_tuning")

    def __exit__(self, *exc):
        L.check(L.load().advx_set_tuning(L.TUNE_SEPARATE_CROP, 0), "advx_set_tuning")


class compose_crop_everywhere:
    """Context manager (tests): compose a crop window with the plan's stage 0 wherever the tables fit, also for the geometries where
    the default keeps two launches because composing was measured slower (Qwen2-VL's two gradient copies, Phi-3.5's two stages)."""

    def __enter__(self):
        L.check(L.load().advx_set_tuning(L.TUNE_SEPARATE_CROP, 2), "advx_set_tuning")

    def __exit__(self, *exc):
        L.check(L.load().advx_set_tuning(L.TUNE_SEPARATE_CROP, 0), "advx_set_tuning")


class full_tap_rows:
    """Context manager (tests): plans UPLOADED inside keep ATen's full tap rows on the device instead of the rows with the
    zero-weight end taps dropped.  Results must be bit-identical."""

    def __enter__(self):
        L.check(L.load().advx_set_tuning(L.TUNE_FULL_TAP_ROWS, 1), "advx_set_tuning")

    def __exit__(self, *exc):
        L.check(L.load().advx_set_tuning(L.TUNE_FULL_TAP_ROWS, 0), "advx_set_tuning")


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise L.AdvxError("advx ops need tensors on a ROCm device (there is no CPU fallback)")


def _f32c(t):
    if t.dtype != torch.float32:
        raise L.AdvxError("advx ops take float32 tensors")
    return t.contiguous()


def _stream(t):
    return L.current_stream(t.device)


def _plan_dtype(plan):
    """Boundary dtype of a plan's pixel_values / grad_out (Plan.set_io; float32 unless changed)."""
    return getattr(plan, "io_dtype", torch.float32)


def _boundary(plan, t, what):
    if t.dtype != _plan_dtype(plan):
        raise L.AdvxError(f"{what} must be {_plan_dtype(plan)} (the plan's boundary dtype), got {t.dtype}")
    return t.contiguous()


def _crop_arg(crop):
    if crop is None:
        return None, None
    arr = (C.c_int32 * 4)(*[int(v) for v in crop])
    return arr, C.cast(arr, C.c_void_p)


# ------------------------------------------------------------------- processor level
def emit(plan, argument, batch, sigma_dev=None, unit_noise=None, philox=None, workspace=None, out=None, keep_padding=False):
    """process(argument) -> repeat(batch) -> + sigma*noise   (attack_model.py:314-321).

    philox = (seed, offset) switches on the in-kernel generator; unit_noise is the parity
    mode (N(0,1) tensor supplied by the caller).  keep_padding: `out` is a buffer the caller
    keeps across steps whose padding tiles are already zero - only the covered elements are
    written (ADVX_PAD_KEEP)."""
    _require_cuda(argument)
    argument = _f32c(argument)
    dev = argument.device
    if workspace is None:
        workspace = torch.empty(plan.workspace_floats, dtype=torch.float32, device=dev)
    if keep_padding and (out is None or out.numel() != batch * plan.out_numel):
        raise L.AdvxError("keep_padding needs the caller's persistent [batch, out_numel] buffer")
    if out is None:
        out = torch.empty((batch, plan.out_numel), dtype=_plan_dtype(plan), device=dev)
    else:
        out = _boundary(plan, out, "out")
    seed, offset = (philox if philox is not None else (0, 0))
    if unit_noise is not None:
        unit_noise = _f32c(unit_noise)
        if unit_noise.numel() != batch * plan.out_numel:
            raise L.AdvxError("unit_noise has the wrong number of elements")
    L.check(L.load().advx_emit_ex(plan.handle, L.ptr(argument), int(batch), L.ptr(sigma_dev), L.ptr(unit_noise),
                                  int(philox is not None), int(seed), int(offset), L.ptr(out), L.ptr(workspace),
                                  int(workspace.numel()), 1 if keep_padding else 0, _stream(argument)), "advx_emit_ex")
    return out


def collect(plan, grad_out, batch, grad_argument=None, accumulate=False, workspace=None):
    """Backward of `emit`: grad_out [batch, out_numel] -> grad wrt the [3,H,W] argument."""
    _require_cuda(grad_out)
    grad_out = _boundary(plan, grad_out, "grad_out")
    dev = grad_out.device
    if grad_out.numel() != batch * plan.out_numel:
        raise L.AdvxError("grad_out has the wrong number of elements")
    if workspace is None:
        workspace = torch.empty(plan.workspace_floats, dtype=torch.float32, device=dev)
    if grad_argument is None:
        grad_argument = torch.empty((3, plan.in_h, plan.in_w), dtype=torch.float32, device=dev)
        accumulate = False
    L.check(L.load().advx_collect(plan.handle, L.ptr(grad_out), int(batch), L.ptr(grad_argument), int(accumulate),
                                  L.ptr(workspace), int(workspace.numel()), _stream(grad_out)), "advx_collect")
    return grad_argument


def crop_composes(plan, H, W, crop):
    """True when advx_forward_multi applies this window's resize and the plan's stage-0 resize as ONE table per axis
    (one plan, include/advx.h "Composed crop"); the backward of such a step is collect_crop + image_bwd* without a window."""
    if crop is None:
        return False
    _keep, cp = _crop_arg(crop)
    return bool(L.load().advx_crop_composes(plan.handle, int(H), int(W), cp))


def crop_compose_strides(plan, H, W, crop):
    """((forward_h, forward_w), (transposed_h, transposed_w)): the row lengths the composed tables are built with."""
    _keep, cp = _crop_arg(crop)
    f, t = (C.c_int32 * 2)(), (C.c_int32 * 2)()
    L.check(L.load().advx_crop_compose_strides(plan.handle, int(H), int(W), cp, f, t), "advx_crop_compose_strides")
    return (int(f[0]), int(f[1])), (int(t[0]), int(t[1]))


def crop_compose_rows(plan, H, W, crop):
    """(forward, transposed): the composed tables' real longest rows for this window (over both axes)."""
    _keep, cp = _crop_arg(crop)
    f, t = C.c_int32(), C.c_int32()
    L.check(L.load().advx_crop_compose_rows(plan.handle, int(H), int(W), cp, C.byref(f), C.byref(t)), "advx_crop_compose_rows")
    return int(f.value), int(t.value)


def collect_crop(plan, grad_out, batch, crop, image_scratch, grad_s, accumulate=False, workspace=None):
    """Backward of a forward_multi that composed: grad_out [batch, out_numel] -> gradient w.r.t. the IMAGE s [3,H,W]
    (exact zeros outside the window)."""
    _require_cuda(grad_out, grad_s, image_scratch)
    grad_out = _boundary(plan, grad_out, "grad_out")
    if grad_out.numel() != batch * plan.out_numel:
        raise L.AdvxError("grad_out has the wrong number of elements")
    if workspace is None:
        workspace = torch.empty(plan.workspace_floats, dtype=torch.float32, device=grad_out.device)
    _keep, cp = _crop_arg(crop)
    L.check(L.load().advx_collect_crop(plan.handle, L.ptr(grad_out), int(batch), L.ptr(grad_s), int(accumulate), L.ptr(workspace),
                                       int(workspace.numel()), int(plan.in_h), int(plan.in_w), cp,
                                       L.ptr(image_scratch), _stream(grad_out)), "advx_collect_crop")
    return grad_s


def collect_update_supported(plan, H, W, crop=None):
    """Whether collect[_crop] + image_bwd_update of a step without blur can go through `collect_update` (one launch less)."""
    _keep, cp = _crop_arg(crop)
    return bool(L.load().advx_collect_update_supported(plan.handle, int(H), int(W), cp))


def collect_update(plan, grad_out, batch, p, s, epsilon, imgfit_scale, grad_p, mask, m, v, opt, stats, image_scratch,
                   update_scratch, crop=None, accumulate=False, finalize_norm=True, workspace=None, if_supported=False):
    """collect (crop=None) or collect_crop (a composing window) AND image_bwd_update(blur=None) in one call: the transposed
    resize of stage 0 runs inside the optimiser's launch.  Where `collect_update_supported` says no: AdvxError
    (ADVX_E_UNSUPPORTED), or None with if_supported=True (nothing launched)."""
    _require_cuda(grad_out, p, s, grad_p, mask, stats, image_scratch, update_scratch)
    grad_out = _boundary(plan, grad_out, "grad_out")
    if grad_out.numel() != batch * plan.out_numel:
        raise L.AdvxError("grad_out has the wrong number of elements")
    if workspace is None:
        workspace = torch.empty(plan.workspace_floats, dtype=torch.float32, device=grad_out.device)
    _, H, W = p.shape
    _keep, cp = _crop_arg(crop)
    rc = L.load().advx_collect_update(plan.handle, L.ptr(grad_out), int(batch), L.ptr(workspace), int(workspace.numel()),
                                      int(H), int(W), cp, L.ptr(image_scratch), L.ptr(p), L.ptr(s), float(epsilon),
                                      float(imgfit_scale), L.ptr(grad_p), int(accumulate), L.ptr(mask), L.ptr(m), L.ptr(v),
                                      C.byref(opt), L.ptr(stats), L.ptr(update_scratch), int(bool(finalize_norm)), _stream(p))
    if rc == L.E_UNSUPPORTED and if_supported:
        return None            # nothing was launched: the caller takes the two calls
    L.check(rc, "advx_collect_update")
    return grad_p


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def emit_multi(plans, argument, batches, sigma_dev=None, unit_noises=None, philox=None, workspaces=None, outs=None,
               keep_padding=False):
    """`emit` for several plans over one image (cross-model runs): same tensors, bit for bit, with the
    plans' image resizes in one launch.  philox = (seed, [offset per plan])."""
    _require_cuda(argument)
    argument = _f32c(argument)
    dev = argument.device
    n = len(plans)
    if workspaces is None:
        workspaces = [torch.empty(pl.workspace_floats, dtype=torch.float32, device=dev) for pl in plans]
    if outs is None:
        outs = [None] * n
    outs = list(outs)
    for i, (pl, B) in enumerate(zip(plans, batches)):
        if keep_padding and (outs[i] is None or outs[i].numel() != B * pl.out_numel):
            raise L.AdvxError("keep_padding needs the caller's persistent [batch, out_numel] buffers")
        outs[i] = (torch.empty((B, pl.out_numel), dtype=_plan_dtype(pl), device=dev) if outs[i] is None
                   else _boundary(pl, outs[i], "out"))
    zs = [None] * n if unit_noises is None else [None if z is None else _f32c(z) for z in unit_noises]
    for z, pl, B in zip(zs, plans, batches):
        if z is not None and z.numel() != B * pl.out_numel:
            raise L.AdvxError("unit_noise has the wrong number of elements")
    seed, offsets = (philox if philox is not None else (0, [0] * n))
    L.check(L.load().advx_emit_multi(n, (C.c_void_p * n)(*[pl.handle.value for pl in plans]), L.ptr(argument),
                                     (C.c_int32 * n)(*[int(b) for b in batches]), L.ptr(sigma_dev), _ptr_array(zs),
                                     int(philox is not None), int(seed), (C.c_uint64 * n)(*[int(o) for o in offsets]),
                                     _ptr_array(outs), _ptr_array(workspaces),
                                     (C.c_int64 * n)(*[int(w.numel()) for w in workspaces]), 1 if keep_padding else 0,
                                     _stream(argument)), "advx_emit_multi")
    return outs


def forward_multi(p, x0, epsilon, stats, image_scratch, plans, batches, s, argument=None, blur=None, crop=None,
                  unit_noises=None, philox=None, workspaces=None, outs=None, keep_padding=False, image_ready=False):
    """image_fwd + emit_multi in one call (advx_forward_multi).  -> (list of pixel_values, argument).

    The returned `argument` is what the plans read: `s` without a crop window, the window resized back to H x W with one.
    COMPOSED CROP: with ONE plan and a window for which `crop_composes(plan, H, W, crop)` holds, the library applies the
    window's resize and the plan's own as one table (include/advx.h "Composed crop") and the resized window is never formed:
    the second return value is then None (a buffer passed as `argument` is left untouched), and the backward of that step is
    `collect_crop` + `image_bwd*` WITHOUT a crop window.

    image_ready=True (advx_forward_multi_ready): the previous step's `image_step` already ran this step's first image kernel -
    `s` holds the image, its statistics partials and the composed tables' forward rows are in `image_scratch`; only the
    plans' resizes and the emits are launched.  Needs blur, and a crop window only if it composes."""
    _require_cuda(p, x0, stats, image_scratch, s)
    dev = p.device
    _, H, W = p.shape
    n = len(plans)
    composed = bool(crop is not None and n == 1 and crop_composes(plans[0], H, W, crop))
    if crop is not None and argument is None and not composed:
        argument = torch.empty_like(p)
    k, sig = (blur if blur is not None else (0, 0.0))
    keep, cptr = _crop_arg(crop)
    if workspaces is None:
        workspaces = [torch.empty(pl.workspace_floats, dtype=torch.float32, device=dev) for pl in plans]
    outs = [None] * n if outs is None else list(outs)
    for i, (pl, B) in enumerate(zip(plans, batches)):
        if keep_padding and (outs[i] is None or outs[i].numel() != B * pl.out_numel):
            raise L.AdvxError("keep_padding needs the caller's persistent [batch, out_numel] buffers")
        outs[i] = (torch.empty((B, pl.out_numel), dtype=_plan_dtype(pl), device=dev) if outs[i] is None
                   else _boundary(pl, outs[i], "out"))
    zs = [None] * n if unit_noises is None else [None if z is None else _f32c(z) for z in unit_noises]
    for z, pl, B in zip(zs, plans, batches):
        if z is not None and z.numel() != B * pl.out_numel:
            raise L.AdvxError("unit_noise has the wrong number of elements")
    seed, offsets = (philox if philox is not None else (0, [0] * n))
    if image_ready:
        if crop is not None and not composed:
            raise L.AdvxError("forward_multi(image_ready=True): the crop window must compose with the plan")
        L.check(L.load().advx_forward_multi_ready(H, W, int(k), cptr, L.ptr(s), L.ptr(stats), L.ptr(image_scratch), n,
                                                  (C.c_void_p * n)(*[pl.handle.value for pl in plans]),
                                                  (C.c_int32 * n)(*[int(b) for b in batches]), _ptr_array(zs),
                                                  int(philox is not None), int(seed), (C.c_uint64 * n)(*[int(o) for o in offsets]),
                                                  _ptr_array(outs), _ptr_array(workspaces),
                                                  (C.c_int64 * n)(*[int(w.numel()) for w in workspaces]),
                                                  1 if keep_padding else 0, _stream(s)), "advx_forward_multi_ready")
        return outs, (None if composed else s)
    L.check(L.load().advx_forward_multi(L.ptr(p), L.ptr(x0), H, W, float(epsilon), int(k), float(sig), cptr, L.ptr(s),
                                        L.ptr(argument) if argument is not None else None, L.ptr(stats), L.ptr(image_scratch),
                                        n, (C.c_void_p * n)(*[pl.handle.value for pl in plans]),
                                        (C.c_int32 * n)(*[int(b) for b in batches]), _ptr_array(zs), int(philox is not None),
                                        int(seed), (C.c_uint64 * n)(*[int(o) for o in offsets]), _ptr_array(outs),
                                        _ptr_array(workspaces), (C.c_int64 * n)(*[int(w.numel()) for w in workspaces]),
                                        1 if keep_padding else 0, _stream(p)), "advx_forward_multi")
    if composed:
        return outs, None
    return outs, (argument if argument is not None else s)


def collect_multi(plans, grad_outs, batches, grad_argument=None, accumulate=False, workspaces=None):
    """Backward of `emit_multi`: the sum over plans of each plan's image gradient, in plan order."""
    n = len(plans)
    _require_cuda(*grad_outs)
    grad_outs = [_boundary(pl, g, "grad_out") for pl, g in zip(plans, grad_outs)]
    dev = grad_outs[0].device
    for pl, g, B in zip(plans, grad_outs, batches):
        if g.numel() != B * pl.out_numel:
            raise L.AdvxError("grad_out has the wrong number of elements")
    if workspaces is None:
        workspaces = [torch.empty(pl.workspace_floats, dtype=torch.float32, device=dev) for pl in plans]
    if grad_argument is None:
        grad_argument = torch.empty((3, plans[0].in_h, plans[0].in_w), dtype=torch.float32, device=dev)
        accumulate = False
    L.check(L.load().advx_collect_multi(n, (C.c_void_p * n)(*[pl.handle.value for pl in plans]), _ptr_array(grad_outs),
                                        (C.c_int32 * n)(*[int(b) for b in batches]), L.ptr(grad_argument), int(accumulate),
                                        _ptr_array(workspaces), (C.c_int64 * n)(*[int(w.numel()) for w in workspaces]),
                                        _stream(grad_outs[0])), "advx_collect_multi")
    return grad_argument


class ProcessFunction(torch.autograd.Function):
    """Differentiable `process(image)` of the plugin API: forward = advx_emit (batch 1, no
    noise), backward = advx_collect.  Keeps autograd users of the reference API working."""

    @staticmethod
    def forward(ctx, image, plan):
        ctx.plan = plan
        out = emit(plan, image.detach(), 1)
        return out.view(plan.out_shape)

    @staticmethod
    def backward(ctx, grad):
        plan = ctx.plan
        g = collect(plan, grad.contiguous().view(1, plan.out_numel), 1)
        return g, None


# ----------------------------------------------------------------------- image level
def image_scratch(H, W, blur_k, device):
    n = L.load().advx_image_scratch_floats(int(H), int(W), int(blur_k))
    return torch.empty(int(n), dtype=torch.float32, device=device)


def image_fwd(p, x0, epsilon, stats, scratch, blur=None, crop=None, s=None, argument=None):
    """x = eps*tanh(p) -> [blur] -> s = x0 + x -> [crop+resize] -> argument; statistics into
    `stats` (attack_model.py:300-312,329,366-373).  blur = (kernel_size, sigma)."""
    _require_cuda(p, x0, stats, scratch)
    _, H, W = p.shape
    if s is None:
        s = torch.empty_like(p)
    if crop is not None and argument is None:
        argument = torch.empty_like(p)
    k, sig = (blur if blur is not None else (0, 0.0))
    keep, cptr = _crop_arg(crop)
    L.check(L.load().advx_image_fwd(L.ptr(p), L.ptr(x0), H, W, float(epsilon), int(k), float(sig), cptr, L.ptr(s),
                                    L.ptr(argument) if argument is not None else None, L.ptr(stats), L.ptr(scratch),
                                    _stream(p)), "advx_image_fwd")
    return s, (argument if argument is not None else s)


def image_bwd(p, s, grad_argument, epsilon, imgfit_scale, grad_p, scratch, blur=None, crop=None, accumulate=False):
    _require_cuda(p, s, grad_argument, grad_p, scratch)
    _, H, W = p.shape
    k, sig = (blur if blur is not None else (0, 0.0))
    keep, cptr = _crop_arg(crop)
    L.check(L.load().advx_image_bwd(L.ptr(p), L.ptr(s), L.ptr(_f32c(grad_argument)), H, W, float(epsilon), int(k),
                                    float(sig), cptr, float(imgfit_scale), L.ptr(grad_p), int(accumulate), L.ptr(scratch),
                                    _stream(p)), "advx_image_bwd")
    return grad_p


def image_bwd_update(p, s, grad_argument, epsilon, imgfit_scale, grad_p, mask, m, v, opt, stats, image_scratch, update_scratch,
                     blur=None, crop=None, accumulate=False, finalize_norm=True):
    """image_bwd + update in one call (nothing in between: no all-reduce).  finalize_norm=False leaves
    stats[GRAD_NORM] to `update_flush` (call it before reading the statistics)."""
    _require_cuda(p, s, grad_argument, grad_p, mask, stats, image_scratch, update_scratch)
    _, H, W = p.shape
    k, sig = (blur if blur is not None else (0, 0.0))
    keep, cptr = _crop_arg(crop)
    L.check(L.load().advx_image_bwd_update(L.ptr(p), L.ptr(s), L.ptr(_f32c(grad_argument)), H, W, float(epsilon), int(k),
                                           float(sig), cptr, float(imgfit_scale), L.ptr(grad_p), int(accumulate), L.ptr(mask),
                                           L.ptr(m), L.ptr(v), C.byref(opt), L.ptr(stats), L.ptr(image_scratch),
                                           L.ptr(update_scratch), int(bool(finalize_norm)), _stream(p)), "advx_image_bwd_update")
    return grad_p


def image_step_supported(H, W, blur_kernel):
    return bool(blur_kernel) and bool(L.load().advx_image_step_supported(int(H), int(W), int(blur_kernel)))


def image_step(p, m, v, p_out, m_out, v_out, s, grad_s, epsilon, imgfit_scale, grad_p, mask, opt, image_scratch, update_scratch,
               x0, s_next, blur, next_blur_sigma, next_crop=None, next_plan=None):
    """Backward + update of step t and the first image kernel of step t+1 in ONE launch (advx_image_step): `grad_s` is the
    gradient w.r.t. the image s (collect / collect_crop); p, m, v -> p_out, m_out, v_out (other buffers), s_next another buffer
    than s.  The next forward is forward_multi(..., image_ready=True) with the same blur kernel size, `next_crop` and plans."""
    _require_cuda(p, p_out, s, grad_s, grad_p, mask, image_scratch, update_scratch, x0, s_next)
    _, H, W = p.shape
    k, sig = blur
    keep, cptr = _crop_arg(next_crop)
    L.check(L.load().advx_image_step(L.ptr(p), L.ptr(m), L.ptr(v), L.ptr(p_out), L.ptr(m_out), L.ptr(v_out), L.ptr(s),
                                     L.ptr(_f32c(grad_s)), H, W, float(epsilon), int(k), float(sig), float(imgfit_scale),
                                     L.ptr(grad_p), L.ptr(mask), C.byref(opt), L.ptr(image_scratch), L.ptr(update_scratch),
                                     L.ptr(x0), float(next_blur_sigma), cptr,
                                     next_plan.handle if (next_plan is not None and next_crop is not None) else None,
                                     L.ptr(s_next), _stream(p)), "advx_image_step")


def update_flush(n, stats, update_scratch):
    L.check(L.load().advx_update_flush(int(n), L.ptr(stats), L.ptr(update_scratch), _stream(stats)), "advx_update_flush")


def update(p, m, v, grad_p, mask, opt, stats, scratch):
    _require_cuda(p, grad_p, mask, stats, scratch)
    L.check(L.load().advx_update(L.ptr(p), L.ptr(m), L.ptr(v), L.ptr(grad_p), L.ptr(mask), p.numel(), C.byref(opt),
                                 L.ptr(stats), L.ptr(scratch), _stream(p)), "advx_update")


def update_scratch(n, device):
    return torch.empty(int(L.load().advx_update_scratch_floats(int(n))), dtype=torch.float32, device=device)


# ----------------------------------------------------------------------------- fused
def fused_scratch(plan, device):
    # zero-initialised: the header says "nothing pending"
    return torch.zeros(int(L.load().advx_fused_scratch_floats(plan.handle)), dtype=torch.float32, device=device)


def fused_flush(plan, stats, scratch, image_too=False):
    """Run the reductions the fused pair deferred (before reading stats on the host)."""
    L.check(L.load().advx_fused_flush(plan.handle, L.ptr(stats), L.ptr(scratch), int(image_too), _stream(stats)),
            "advx_fused_flush")


# ADVX_IO_* of include/advx.h
IO_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def io_code(dtype):
    try:
        return IO_DTYPES[dtype]
    except KeyError:
        raise L.AdvxError(f"the fused pair reads/writes float32, float16 or bfloat16, not {dtype}") from None


def fused_fwd(plan, p, x0, epsilon, batch, stats, scratch, s_buf, v_buf, prepared, unit_noise=None, philox=None, out=None,
              parity=0, out_dtype=torch.float32, step_chain_noise=False):
    """step_chain_noise: address the generator as advx_fused_step does (ADVX_PHILOX_STEP_CHAIN)."""
    _require_cuda(p, x0, stats, scratch, s_buf, v_buf)
    if out is None:
        out = torch.empty((batch, plan.out_numel), dtype=out_dtype, device=p.device)
    elif not out.is_contiguous() or out.numel() != batch * plan.out_numel:
        raise L.AdvxError("out must be a contiguous [batch, out_numel] tensor")
    seed, offset = (philox if philox is not None else (0, 0))
    L.check(L.load().advx_fused_fwd_io(plan.handle, L.ptr(p), L.ptr(x0), float(epsilon), int(batch), L.ptr(unit_noise),
                                       (L.PHILOX_STEP_CHAIN if step_chain_noise else 1) if philox is not None else 0,
                                       int(seed), int(offset), L.ptr(out), io_code(out.dtype),
                                       L.ptr(s_buf), L.ptr(v_buf), int(bool(prepared)), int(parity), L.ptr(stats),
                                       L.ptr(scratch), _stream(p)), "advx_fused_fwd_io")
    return out


def fused_bwd(plan, grad_out, batch, p, x0, epsilon, imgfit_scale, grad_p, stats, scratch, mask=None, m=None, v=None,
              opt=None, s_next=None, v_buf=None):
    """grad_out may be float32, float16 or bfloat16 (read as is, accumulated in fp32)."""
    _require_cuda(grad_out, p, x0, grad_p, scratch, stats)
    io = io_code(grad_out.dtype)
    grad_out = grad_out.contiguous()
    if grad_out.numel() != batch * plan.out_numel:
        raise L.AdvxError("grad_out has the wrong number of elements")
    L.check(L.load().advx_fused_bwd_io(plan.handle, L.ptr(grad_out), io, int(batch), L.ptr(p), L.ptr(x0), float(epsilon),
                                       float(imgfit_scale), L.ptr(mask), L.ptr(m), L.ptr(v), L.ptr(grad_p),
                                       C.byref(opt) if opt is not None else None, L.ptr(s_next), L.ptr(v_buf),
                                       L.ptr(stats), L.ptr(scratch), _stream(p)), "advx_fused_bwd_io")


def fused_bwd_dp(plan, exchange, grad_out, batch, p, x0, epsilon, imgfit_scale, stats, scratch, mask, m, v, opt, s_next,
                 v_buf):
    """Data-parallel backward of the pair in ONE call: gradient-only backward into the exchange's
    send buffer, peer all-reduce (dp.PeerExchange), fused update from its recv buffer."""
    _require_cuda(grad_out, p, x0, scratch, stats, mask, s_next, v_buf)
    io = io_code(grad_out.dtype)
    grad_out = grad_out.contiguous()
    if grad_out.numel() != batch * plan.out_numel:
        raise L.AdvxError("grad_out has the wrong number of elements")
    L.check(L.load().advx_fused_bwd_dp(plan.handle, exchange.handle, L.ptr(grad_out), io, int(batch), L.ptr(p), L.ptr(x0),
                                       float(epsilon), float(imgfit_scale), L.ptr(mask), L.ptr(m), L.ptr(v), C.byref(opt),
                                       L.ptr(s_next), L.ptr(v_buf), L.ptr(stats), L.ptr(scratch),
                                       float(exchange.timeout_s), _stream(p)), "advx_fused_bwd_dp")


def make_sched(opt_table, first_step, device):
    """Device copy of the per-step scalars of `len(opt_table)` steps starting at `first_step` (advx_sched_fill)."""
    n = len(opt_table)
    lib = L.load()
    host = C.create_string_buffer(int(lib.advx_sched_bytes(n)))
    arr = (L.OptScalars * n)(*opt_table)
    L.check(lib.advx_sched_fill(host, n, arr, int(first_step)), "advx_sched_fill")
    return torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(device)


def fused_fwd_sched(plan, p, x0, epsilon, batch, seed, out, s_buf, v_buf, stats, scratch, sched, offset_base=0):
    """advx_fused_fwd_io (prepared, in-kernel noise) with the Philox offset read from `sched`: replayable."""
    _require_cuda(p, x0, out, s_buf, v_buf, stats, scratch, sched)
    if not out.is_contiguous() or out.numel() != batch * plan.out_numel:
        raise L.AdvxError("out must be a contiguous [batch, out_numel] tensor")
    L.check(L.load().advx_fused_fwd_sched(plan.handle, L.ptr(p), L.ptr(x0), float(epsilon), int(batch), int(seed), int(offset_base),
                                          L.ptr(out), io_code(out.dtype), L.ptr(s_buf), L.ptr(v_buf), L.ptr(stats),
                                          L.ptr(scratch), L.ptr(sched), _stream(p)), "advx_fused_fwd_sched")
    return out


def fused_bwd_sched(plan, grad_out, batch, p, x0, epsilon, imgfit_scale, mask, m, v, grad_p, opt_kind, s_next, v_buf, stats,
                    scratch, sched):
    """advx_fused_bwd_io with the optimiser scalars of this step read from `sched`: replayable."""
    _require_cuda(grad_out, p, x0, mask, grad_p, s_next, v_buf, stats, scratch, sched)
    if not grad_out.is_contiguous() or grad_out.numel() != batch * plan.out_numel:
        raise L.AdvxError("grad_out must be a contiguous [batch, out_numel] tensor")
    L.check(L.load().advx_fused_bwd_sched(plan.handle, L.ptr(grad_out), io_code(grad_out.dtype), int(batch), L.ptr(p), L.ptr(x0),
                                          float(epsilon), float(imgfit_scale), L.ptr(mask), L.ptr(m), L.ptr(v), L.ptr(grad_p),
                                          int(opt_kind), L.ptr(s_next), L.ptr(v_buf), L.ptr(stats), L.ptr(scratch),
                                          L.ptr(sched), _stream(p)), "advx_fused_bwd_sched")


def fused_update(plan, p, m, v, grad_p, mask, x0, epsilon, opt, s_next, v_buf, scratch):
    """DP tail of the pair: mask, ||g|| partials, optimiser step, preparation of the next forward."""
    _require_cuda(p, grad_p, mask, x0, s_next, v_buf, scratch)
    L.check(L.load().advx_fused_update(plan.handle, L.ptr(p), L.ptr(m), L.ptr(v), L.ptr(grad_p), L.ptr(mask), L.ptr(x0),
                                       float(epsilon), C.byref(opt), L.ptr(s_next), L.ptr(v_buf), L.ptr(scratch), _stream(p)),
            "advx_fused_update")


def fused_step_rows(plan):
    a, b = C.c_int32(), C.c_int32()
    L.check(L.load().advx_fused_step_rows(plan.handle, C.byref(a), C.byref(b)), "advx_fused_step_rows")
    return a.value, b.value


def fused_step(plan, grad_out, batch, p, x0, epsilon, imgfit_scale, mask, m, v, grad_p, opt, out_next, s_next, v_buf,
               parity, image_rows_in, norm_rows_in, stats, scratch, unit_noise_next=None, philox=None):
    """bwd(t) + fwd(t+1) in one launch: grad_out of step t in, pixel_values of step t+1 out."""
    _require_cuda(grad_out, p, x0, grad_p, out_next, stats, scratch)
    seed, offset = (philox if philox is not None else (0, 0))
    L.check(L.load().advx_fused_step(plan.handle, L.ptr(_f32c(grad_out)), int(batch), L.ptr(p), L.ptr(x0), float(epsilon),
                                     float(imgfit_scale), L.ptr(mask), L.ptr(m), L.ptr(v), L.ptr(grad_p), C.byref(opt),
                                     L.ptr(unit_noise_next), int(philox is not None), int(seed), int(offset),
                                     L.ptr(out_next), L.ptr(s_next), L.ptr(v_buf), int(parity), int(image_rows_in),
                                     int(norm_rows_in), L.ptr(stats), L.ptr(scratch), _stream(p)), "advx_fused_step")


def fused_step_flush(plan, parity, norm_rows, stats, scratch):
    L.check(L.load().advx_fused_step_flush(plan.handle, int(parity), int(norm_rows), L.ptr(stats), L.ptr(scratch),
                                           _stream(stats)), "advx_fused_step_flush")


def profile_begin(max_launches=4096, stride=1):
    L.check(L.load().advx_profile_begin(int(max_launches), int(stride)), "advx_profile_begin")


def profile_end():
    """{'fwd'|'bwd'|'step': (avg_ms, launches)} of the launches timed since profile_begin."""
    ms = (C.c_double * 3)()
    n = (C.c_int64 * 3)()
    L.check(L.load().advx_profile_end(ms, n), "advx_profile_end")
    return {k: ((ms[i] / n[i]) if n[i] else 0.0, int(n[i])) for i, k in enumerate(("fwd", "bwd", "step"))}


# --------------------------------------------------------------------------- prepared
def prepared_scratch(plan, device):
    return torch.empty(int(L.load().advx_prepared_scratch_floats(plan.handle)), dtype=torch.float32, device=device)


def prepared_rows(plan):
    a, b = C.c_int32(), C.c_int32()
    L.check(L.load().advx_prepared_rows(plan.handle, C.byref(a), C.byref(b)), "advx_prepared_rows")
    return int(a.value), int(b.value)


def prepared_fwd(plan, p, x0, epsilon, batch, stats, scratch, workspace, s_buf, prepared, parity, unit_noise=None,
                 philox=None, out=None, keep_padding=False):
    _require_cuda(p, x0, stats, scratch, workspace, s_buf)
    if keep_padding and (out is None or out.numel() != batch * plan.out_numel):
        raise L.AdvxError("keep_padding needs the caller's persistent [batch, out_numel] buffer")
    if out is None:
        out = torch.empty((batch, plan.out_numel), dtype=_plan_dtype(plan), device=p.device)
    else:
        out = _boundary(plan, out, "out")
    if unit_noise is not None:
        unit_noise = _f32c(unit_noise)
        if unit_noise.numel() != batch * plan.out_numel:
            raise L.AdvxError("unit_noise has the wrong number of elements")
    seed, offset = (philox if philox is not None else (0, 0))
    L.check(L.load().advx_prepared_fwd(plan.handle, L.ptr(p), L.ptr(x0), float(epsilon), int(batch), L.ptr(unit_noise),
                                       int(philox is not None), int(seed), int(offset), L.ptr(out), L.ptr(s_buf),
                                       int(bool(prepared)), int(parity), L.ptr(stats), L.ptr(scratch), L.ptr(workspace),
                                       int(workspace.numel()), 1 if keep_padding else 0, _stream(p)), "advx_prepared_fwd")
    return out


def prepared_bwd(plan, grad_out, batch, p, x0, epsilon, imgfit_scale, mask, m, v, grad_p, opt, s_next, rows_in, parity,
                 stats, scratch, workspace):
    _require_cuda(grad_out, p, x0, mask, grad_p, s_next, stats, scratch, workspace)
    grad_out = _boundary(plan, grad_out, "grad_out")
    if grad_out.numel() != batch * plan.out_numel:
        raise L.AdvxError("grad_out has the wrong number of elements")
    L.check(L.load().advx_prepared_bwd(plan.handle, L.ptr(grad_out), int(batch), L.ptr(p), L.ptr(x0), float(epsilon),
                                       float(imgfit_scale), L.ptr(mask), L.ptr(m), L.ptr(v), L.ptr(grad_p), C.byref(opt),
                                       L.ptr(s_next), int(rows_in), int(parity), L.ptr(stats), L.ptr(scratch),
                                       L.ptr(workspace), int(workspace.numel()), _stream(p)), "advx_prepared_bwd")


def prepared_bwd_grad(plan, grad_out, batch, p, x0, epsilon, imgfit_scale, grad_p, rows_in, parity, stats, scratch,
                      workspace):
    """Data-parallel first half: batch-reduce and this rank's unmasked image gradient -> grad_p."""
    _require_cuda(grad_out, p, x0, grad_p, stats, scratch, workspace)
    grad_out = _boundary(plan, grad_out, "grad_out")
    if grad_out.numel() != batch * plan.out_numel:
        raise L.AdvxError("grad_out has the wrong number of elements")
    L.check(L.load().advx_prepared_bwd_grad(plan.handle, L.ptr(grad_out), int(batch), L.ptr(p), L.ptr(x0), float(epsilon),
                                            float(imgfit_scale), L.ptr(grad_p), int(rows_in), int(parity), L.ptr(stats),
                                            L.ptr(scratch), L.ptr(workspace), int(workspace.numel()), _stream(p)),
            "advx_prepared_bwd_grad")


def prepared_update(plan, p, m, v, grad_p, mask, x0, epsilon, opt, s_next, parity, stats, scratch, workspace):
    """Data-parallel second half (after the all-reduce of grad_p): mask, ||g||, optimiser, next step."""
    _require_cuda(p, grad_p, mask, x0, s_next, stats, scratch, workspace)
    L.check(L.load().advx_prepared_update(plan.handle, L.ptr(p), L.ptr(m), L.ptr(v), L.ptr(grad_p), L.ptr(mask), L.ptr(x0),
                                          float(epsilon), C.byref(opt), L.ptr(s_next), int(parity), L.ptr(stats),
                                          L.ptr(scratch), L.ptr(workspace), int(workspace.numel()), _stream(p)),
            "advx_prepared_update")


def prepared_bwd_dp(plan, exchange, grad_out, batch, p, x0, epsilon, imgfit_scale, mask, m, v, opt, s_next, rows_in, parity,
                    stats, scratch, workspace):
    """Both halves around the peer exchange (dp.PeerExchange) in one call."""
    _require_cuda(grad_out, p, x0, mask, s_next, stats, scratch, workspace)
    grad_out = _boundary(plan, grad_out, "grad_out")
    if grad_out.numel() != batch * plan.out_numel:
        raise L.AdvxError("grad_out has the wrong number of elements")
    L.check(L.load().advx_prepared_bwd_dp(plan.handle, exchange.handle, L.ptr(grad_out), int(batch), L.ptr(p), L.ptr(x0),
                                          float(epsilon), float(imgfit_scale), L.ptr(mask), L.ptr(m), L.ptr(v),
                                          C.byref(opt), L.ptr(s_next), int(rows_in), int(parity), L.ptr(stats),
                                          L.ptr(scratch), L.ptr(workspace), int(workspace.numel()),
                                          float(exchange.timeout_s), _stream(p)), "advx_prepared_bwd_dp")


def quantise(s, out=None):
    """The image after the lossless PNG round trip of attack_model.py:368-371 (uint8 truncation)."""
    _require_cuda(s)
    s = _f32c(s)
    if out is None:
        out = torch.empty_like(s)
    L.check(L.load().advx_quantise(L.ptr(s), L.ptr(out), s.numel(), _stream(s)), "advx_quantise")
    return out


# ------------------------------------------------------------------------ single ops
def tanh_fwd(p, epsilon):
    _require_cuda(p)
    x = torch.empty_like(p)
    L.check(L.load().advx_tanh_fwd(L.ptr(_f32c(p)), float(epsilon), L.ptr(x), p.numel(), _stream(p)), "advx_tanh_fwd")
    return x


def tanh_bwd(p, grad_x, epsilon):
    _require_cuda(p, grad_x)
    g = torch.empty_like(p)
    L.check(L.load().advx_tanh_bwd(L.ptr(_f32c(p)), L.ptr(_f32c(grad_x)), float(epsilon), L.ptr(g), p.numel(), _stream(p)),
            "advx_tanh_bwd")
    return g


def blur_fwd(x, kernel_size, sigma):
    _require_cuda(x)
    _, H, W = x.shape
    y = torch.empty_like(x)
    L.check(L.load().advx_blur_fwd(L.ptr(_f32c(x)), H, W, int(kernel_size), float(sigma), L.ptr(y), _stream(x)), "advx_blur_fwd")
    return y


def blur_bwd(grad_y, kernel_size, sigma):
    _require_cuda(grad_y)
    _, H, W = grad_y.shape
    r = kernel_size // 2
    scratch = torch.empty(3 * (H + 2 * r) * (W + 2 * r), dtype=torch.float32, device=grad_y.device)
    g = torch.empty_like(grad_y)
    L.check(L.load().advx_blur_bwd(L.ptr(_f32c(grad_y)), H, W, int(kernel_size), float(sigma), L.ptr(g), L.ptr(scratch),
                                   _stream(grad_y)), "advx_blur_bwd")
    return g


def crop_resize_fwd(src, crop):
    _require_cuda(src)
    _, H, W = src.shape
    scratch = torch.empty(int(L.load().advx_crop_scratch_floats(H, W)), dtype=torch.float32, device=src.device)
    dst = torch.empty_like(src)
    keep, cptr = _crop_arg(crop)
    L.check(L.load().advx_crop_resize_fwd(L.ptr(_f32c(src)), H, W, cptr, L.ptr(dst), L.ptr(scratch), _stream(src)),
            "advx_crop_resize_fwd")
    return dst


def crop_resize_bwd(grad_dst, crop):
    _require_cuda(grad_dst)
    _, H, W = grad_dst.shape
    scratch = torch.empty(int(L.load().advx_crop_scratch_floats(H, W)), dtype=torch.float32, device=grad_dst.device)
    g = torch.empty_like(grad_dst)
    keep, cptr = _crop_arg(crop)
    L.check(L.load().advx_crop_resize_bwd(L.ptr(_f32c(grad_dst)), H, W, cptr, L.ptr(g), L.ptr(scratch), _stream(grad_dst)),
            "advx_crop_resize_bwd")
    return g


def batch_reduce(g):
    _require_cuda(g)
    B = g.shape[0]
    n = g.numel() // B
    out = torch.empty(n, dtype=torch.float32, device=g.device)
    L.check(L.load().advx_batch_reduce(L.ptr(_f32c(g)), B, n, L.ptr(out), _stream(g)), "advx_batch_reduce")
    return out


def philox_normal(n, seed, offset, device):
    out = torch.empty(int(n), dtype=torch.float32, device=device)
    L.check(L.load().advx_philox_normal(L.ptr(out), int(n), int(seed), int(offset), L.current_stream(device)),
            "advx_philox_normal")
    return out
