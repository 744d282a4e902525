"""ctypes binding of libadvx_hip.so (include/advx.h).

There is NO CPU fallback: if the HIP library is missing or an entry point fails, this
module raises.  `tests/` use the torch-CPU oracle to CHECK results; the product path never
routes through it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libadvx_hip.so")

ADVX_OK = 0
E_UNSUPPORTED = -4          # ADVX_E_UNSUPPORTED
KIND_LLAVA, KIND_MLLAMA, KIND_PHI3, KIND_QWEN2VL = 0, 1, 2, 3
MODE_AA_BILINEAR, MODE_BILINEAR, MODE_BICUBIC = 0, 1, 2
OPT_ADAMW, OPT_SIGN = 0, 1
PHILOX_STEP_CHAIN = 2     # use_philox of advx_fused_fwd: the one-launch chain's counter addressing
STAT_SIGMA, STAT_QERR_STD, STAT_QERR_MEAN, STAT_QERR_L1, STAT_IMGFIT, STAT_X_MEAN, STAT_X_STD, STAT_GRAD_NORM = range(8)
STATS_N = 16
MAX_STAGES = 2
TUNE_RESET_ALL = 0
TUNE_GENERIC_KERNELS = 1
TUNE_FULL_TAP_ROWS = 3
TUNE_SEPARATE_CROP = 4
TUNE_PAIR_LEAN = 5
TUNE_XCD_MAP, TUNE_BWD_XCD, TUNE_ROW_BATCH, TUNE_IMG_XCD, TUNE_HEAD3, TUNE_BLUR_THREADS, TUNE_TAIL3, TUNE_COLLECT_UPDATE, \
    TUNE_DIRECT_BATCH = 6, 7, 8, 9, 10, 12, 13, 14, 15


class AdvxError(RuntimeError):
    pass


class PlanDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("a0", C.c_int64), ("a1", C.c_int64), ("a2", C.c_int64), ("a3", C.c_int64), ("a4", C.c_int64),
                ("mean", C.c_float * 3), ("std", C.c_float * 3)]


class StageInfo(C.Structure):
    _fields_ = [("mode", C.c_int32), ("src", C.c_int32), ("src_h", C.c_int32), ("src_w", C.c_int32),
                ("res_h", C.c_int32), ("res_w", C.c_int32), ("can_h", C.c_int32), ("can_w", C.c_int32),
                ("off_y", C.c_int32), ("off_x", C.c_int32), ("pad_value", C.c_float),
                ("normalise", C.c_int32), ("inner_axis_h", C.c_int32)]


class PlanInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32), ("out_rank", C.c_int32),
                ("out_shape", C.c_int64 * 6), ("out_numel", C.c_int64), ("n_stage", C.c_int32),
                ("stage", StageInfo * MAX_STAGES), ("tiles_h", C.c_int32), ("tiles_w", C.c_int32),
                ("num_tiles", C.c_int32), ("grid_h", C.c_int32), ("grid_w", C.c_int32),
                ("image_h", C.c_int32), ("image_w", C.c_int32), ("num_img_tokens", C.c_int32),
                ("aspect_ratio_id", C.c_int32), ("workspace_floats", C.c_int64)]


class OptScalars(C.Structure):
    _fields_ = [("kind", C.c_int32), ("apply", C.c_int32), ("lr", C.c_float), ("decay", C.c_float),
                ("w1", C.c_float), ("beta2", C.c_float), ("w2", C.c_float), ("bias2_sqrt", C.c_float),
                ("eps", C.c_float), ("neg_step_size", C.c_float)]


_P = C.c_void_p
_I32, _I64, _U64, _F = C.c_int32, C.c_int64, C.c_uint64, C.c_float
_PI32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/advx.h declares
SIGNATURES = {
    "advx_version": (_I32, []),
    "advx_last_error": (C.c_char_p, []),
    "advx_set_tuning": (_I32, [_I32, _I32]),
    "advx_plan_create": (_I32, [C.POINTER(PlanDesc), C.POINTER(_P)]),
    "advx_plan_destroy": (_I32, [_P]),
    "advx_plan_describe": (_I32, [_P, C.POINTER(PlanInfo)]),
    "advx_plan_upload": (_I32, [_P, _P]),
    "advx_plan_taps": (_I32, [_P, _I32, _I32, _I32, _PI32, _PI32, _P, _P, _P]),
    "advx_plan_out_index": (_I32, [_P, _I32, _I32, _I32, _I32, _PI32, C.POINTER(C.c_int64)]),
    "advx_plan_live_range": (_I32, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "advx_plan_set_io": (_I32, [_P, _I32]),
    "advx_plan_get_io": (_I32, [_P]),
    "advx_taps_compute": (_I32, [_I32, _I32, _I32, _I32, _PI32, _PI32, _P, _P, _P]),
    "advx_emit": (_I32, [_P, _P, _I32, _P, _P, _I32, _U64, _U64, _P, _P, _I64, _P]),
    "advx_emit_ex": (_I32, [_P, _P, _I32, _P, _P, _I32, _U64, _U64, _P, _P, _I64, _I32, _P]),
    "advx_collect": (_I32, [_P, _P, _I32, _P, _I32, _P, _I64, _P]),
    "advx_crop_composes": (_I32, [_P, _I32, _I32, _P]),
    "advx_crop_compose_strides": (_I32, [_P, _I32, _I32, _P, _PI32, _PI32]),
    "advx_crop_compose_rows": (_I32, [_P, _I32, _I32, _P, _PI32, _PI32]),
    "advx_image_grid_map": (_I32, [_I32, _I32, _I32, _I32, _PI32, _PI32]),
    "advx_collect_crop": (_I32, [_P, _P, _I32, _P, _I32, _P, _I64, _I32, _I32, _P, _P, _P]),
    "advx_collect_update_supported": (_I32, [_P, _I32, _I32, _P]),
    "advx_collect_update": (_I32, [_P, _P, _I32, _P, _I64, _I32, _I32, _P, _P, _P, _P, _F, _F, _P, _I32, _P, _P, _P, C.POINTER(OptScalars),
                                   _P, _P, _I32, _P]),
    "advx_emit_multi": (_I32, [_I32, _P, _P, _P, _P, _P, _I32, _U64, _P, _P, _P, _P, _I32, _P]),
    "advx_collect_multi": (_I32, [_I32, _P, _P, _P, _P, _I32, _P, _P, _P]),
    "advx_forward_multi": (_I32, [_P, _P, _I32, _I32, _F, _I32, _F, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _I32, _U64, _P, _P, _P,
                                  _P, _I32, _P]),
    "advx_image_scratch_floats": (_I64, [_I32, _I32, _I32]),
    "advx_image_step_supported": (_I32, [_I32, _I32, _I32]),
    "advx_image_step": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _F, _I32, _F, _F, _P, _P, C.POINTER(OptScalars), _P, _P, _P,
                               _F, _P, _P, _P, _P]),
    "advx_forward_multi_ready": (_I32, [_I32, _I32, _I32, _P, _P, _P, _P, _I32, _P, _P, _P, _I32, _U64, _P, _P, _P, _P, _I32, _P]),
    "advx_image_fwd": (_I32, [_P, _P, _I32, _I32, _F, _I32, _F, _P, _P, _P, _P, _P, _P]),
    "advx_image_bwd": (_I32, [_P, _P, _P, _I32, _I32, _F, _I32, _F, _P, _F, _P, _I32, _P, _P]),
    "advx_update": (_I32, [_P, _P, _P, _P, _P, _I64, C.POINTER(OptScalars), _P, _P, _P]),
    "advx_image_bwd_update": (_I32, [_P, _P, _P, _I32, _I32, _F, _I32, _F, _P, _F, _P, _I32, _P, _P, _P, C.POINTER(OptScalars),
                                     _P, _P, _P, _I32, _P]),
    "advx_update_flush": (_I32, [_I64, _P, _P, _P]),
    "advx_update_scratch_floats": (_I64, [_I64]),
    "advx_fused_supported": (_I32, [_P]),
    "advx_fused_fwd": (_I32, [_P, _P, _P, _F, _I32, _P, _I32, _U64, _U64, _P, _P, _P, _I32, _I32, _P, _P, _P]),
    "advx_fused_bwd": (_I32, [_P, _P, _I32, _P, _P, _F, _F, _P, _P, _P, _P, C.POINTER(OptScalars), _P, _P, _P, _P, _P]),
    "advx_fused_fwd_io": (_I32, [_P, _P, _P, _F, _I32, _P, _I32, _U64, _U64, _P, _I32, _P, _P, _I32, _I32, _P, _P, _P]),
    "advx_fused_bwd_io": (_I32, [_P, _P, _I32, _I32, _P, _P, _F, _F, _P, _P, _P, _P, C.POINTER(OptScalars), _P, _P, _P, _P,
                                 _P]),
    "advx_sched_bytes": (_I64, [_I32]),
    "advx_sched_fill": (_I32, [_P, _I32, C.POINTER(OptScalars), _U64]),
    "advx_fused_fwd_sched": (_I32, [_P, _P, _P, _F, _I32, _U64, _U64, _P, _I32, _P, _P, _P, _P, _P, _P]),
    "advx_fused_bwd_sched": (_I32, [_P, _P, _I32, _I32, _P, _P, _F, _F, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _P]),
    "advx_fused_update": (_I32, [_P, _P, _P, _P, _P, _P, _P, _F, C.POINTER(OptScalars), _P, _P, _P, _P]),
    "advx_fused_scratch_floats": (_I64, [_P]),
    "advx_fused_flush": (_I32, [_P, _P, _P, _I32, _P]),
    "advx_fused_step": (_I32, [_P, _P, _I32, _P, _P, _F, _F, _P, _P, _P, _P, C.POINTER(OptScalars), _P, _I32, _U64, _U64,
                               _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "advx_fused_step_rows": (_I32, [_P, _PI32, _PI32]),
    "advx_fused_step_flush": (_I32, [_P, _I32, _I32, _P, _P, _P]),
    "advx_prepared_supported": (_I32, [_P]),
    "advx_prepared_scratch_floats": (_I64, [_P]),
    "advx_prepared_rows": (_I32, [_P, _PI32, _PI32]),
    "advx_prepared_fwd": (_I32, [_P, _P, _P, _F, _I32, _P, _I32, _U64, _U64, _P, _P, _I32, _I32, _P, _P, _P, _I64, _I32, _P]),
    "advx_prepared_bwd": (_I32, [_P, _P, _I32, _P, _P, _F, _F, _P, _P, _P, _P, C.POINTER(OptScalars), _P, _I32, _I32, _P, _P,
                                 _P, _I64, _P]),
    "advx_prepared_bwd_grad": (_I32, [_P, _P, _I32, _P, _P, _F, _F, _P, _I32, _I32, _P, _P, _P, _I64, _P]),
    "advx_prepared_update": (_I32, [_P, _P, _P, _P, _P, _P, _P, _F, C.POINTER(OptScalars), _P, _I32, _P, _P, _P, _I64, _P]),
    "advx_prepared_bwd_dp": (_I32, [_P, _P, _P, _I32, _P, _P, _F, _F, _P, _P, _P, C.POINTER(OptScalars), _P, _I32, _I32, _P,
                                    _P, _P, _I64, C.c_double, _P]),
    "advx_comm_create": (_I32, [_I32, _I32, _I64, _I32, C.POINTER(_P)]),
    "advx_comm_export": (_I32, [_P, _P]),
    "advx_comm_connect": (_I32, [_P, _P]),
    "advx_comm_send_buffer": (_P, [_P]),
    "advx_comm_recv_buffer": (_P, [_P]),
    "advx_comm_mem_kind": (_I32, [_P]),
    "advx_comm_allreduce": (_I32, [_P, _I64, C.c_double, _P]),
    "advx_comm_status": (_I32, [_P, _PI32, _P]),
    "advx_comm_destroy": (_I32, [_P]),
    "advx_fused_bwd_dp": (_I32, [_P, _P, _P, _I32, _I32, _P, _P, _F, _F, _P, _P, _P, C.POINTER(OptScalars), _P, _P, _P, _P,
                                 C.c_double, _P]),
    "advx_ce_fwd": (_I32, [_P, _I32, _I64, _I64, _I32, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "advx_ce_scratch_floats": (_I64, [_I64, _I64, _I32]),
    "advx_ce_bwd": (_I32, [_P, _I32, _I64, _I64, _I32, _I32, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "advx_profile_begin": (_I32, [_I32, _I32]),
    "advx_profile_end": (_I32, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "advx_quantise": (_I32, [_P, _P, _I64, _P]),
    "advx_tanh_fwd": (_I32, [_P, _F, _P, _I64, _P]),
    "advx_tanh_bwd": (_I32, [_P, _P, _F, _P, _I64, _P]),
    "advx_blur_fwd": (_I32, [_P, _I32, _I32, _I32, _F, _P, _P]),
    "advx_blur_bwd": (_I32, [_P, _I32, _I32, _I32, _F, _P, _P, _P]),
    "advx_crop_resize_fwd": (_I32, [_P, _I32, _I32, _P, _P, _P, _P]),
    "advx_crop_resize_bwd": (_I32, [_P, _I32, _I32, _P, _P, _P, _P]),
    "advx_crop_scratch_floats": (_I64, [_I32, _I32]),
    "advx_batch_reduce": (_I32, [_P, _I32, _I64, _P, _P]),
    "advx_philox_normal": (_I32, [_P, _I64, _U64, _U64, _P]),
}

_lib = None


def load():
    """Load libadvx_hip.so; raises AdvxError (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdvxError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch must load ITS HIP runtime first: libadvx_hip.so then binds to the same
    # libamdhip64 (same soname), so torch's streams and device pointers are valid inside the
    # library.  Loading in the other order maps a second runtime that sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    with open("/proc/self/maps") as f:
        runtimes = {ln.split()[-1] for ln in f if "libamdhip64" in ln}
    if len(runtimes) > 1:
        raise AdvxError(f"two HIP runtimes mapped ({sorted(runtimes)}): import torch before loading libadvx_hip.so")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != ADVX_OK:
        msg = load().advx_last_error()
        raise AdvxError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor / None, as the plain integer ctypes takes for a void*
    (no c_void_p object per argument: a step of the pair passes nineteen pointers)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise AdvxError("advx: tensor must be contiguous")
    return t.data_ptr()


_raw_stream = None


def current_stream(device=None):
    """The current HIP stream of `device` as an integer handle (torch's raw-stream accessor: a tenth of the cost of
    building a torch.cuda.Stream object per call; falls back to it where the accessor is missing)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        if device is None:
            index = torch.cuda.current_device()
        else:
            if isinstance(device, str):
                device = torch.device(device)
            index = device.index if isinstance(device, torch.device) else int(device)
            if index is None:
                index = torch.cuda.current_device()
        return _raw_stream(index)
    return torch.cuda.current_stream(device).cuda_stream
