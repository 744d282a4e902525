"""Plans: integer geometry + float32 tap tables of one (H, W) -> processor layout.

Thin wrapper over `advx_plan_*` (include/advx.h).  Creation runs on the host only, so the
geometry and the tables can be checked without a GPU.
"""
import ctypes as C

import numpy as np

from . import _lib as L

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class Plan:
    def __init__(self, kind, in_h, in_w, args=(), mean=CLIP_MEAN, std=CLIP_STD):
        lib = L.load()
        a = list(args) + [0] * (5 - len(args))
        d = L.PlanDesc(kind=kind, in_h=int(in_h), in_w=int(in_w), a0=int(a[0]), a1=int(a[1]), a2=int(a[2]),
                       a3=int(a[3]), a4=int(a[4]))
        for i in range(3):
            d.mean[i] = float(mean[i])
            d.std[i] = float(std[i])
        h = C.c_void_p()
        L.check(lib.advx_plan_create(C.byref(d), C.byref(h)), "advx_plan_create")
        self._h = h
        self.desc = d
        info = L.PlanInfo()
        L.check(lib.advx_plan_describe(h, C.byref(info)), "advx_plan_describe")
        self.info = info
        self.kind = kind
        self.in_h, self.in_w = int(in_h), int(in_w)
        self.out_shape = tuple(int(info.out_shape[i]) for i in range(info.out_rank))
        self.out_numel = int(info.out_numel)
        self.workspace_floats = int(info.workspace_floats)
        self.mean, self.std = tuple(mean), tuple(std)

    # -- factories mirroring the four reference processors
    @classmethod
    def llava(cls, H, W, crop_h=336, crop_w=336, **kw):
        return cls(L.KIND_LLAVA, H, W, (crop_h, crop_w), **kw)

    @classmethod
    def mllama(cls, H, W, tile=560, max_tiles=4, **kw):
        return cls(L.KIND_MLLAMA, H, W, (tile, max_tiles), **kw)

    @classmethod
    def phi3(cls, H, W, num_crops=6, **kw):
        return cls(L.KIND_PHI3, H, W, (num_crops,), **kw)

    @classmethod
    def qwen2vl(cls, H, W, patch=14, merge=2, temporal=2, min_pixels=56 * 56, max_pixels=28 * 28 * 1280, **kw):
        return cls(L.KIND_QWEN2VL, H, W, (patch, merge, temporal, min_pixels, max_pixels), **kw)

    @property
    def handle(self):
        return self._h

    def stage(self, k):
        return self.info.stage[k]

    def taps(self, stage, axis, transposed=False):
        """(start[n], count[n], weight[n, stride]) as numpy arrays (host copy)."""
        lib = L.load()
        n, stride = C.c_int32(), C.c_int32()
        L.check(lib.advx_plan_taps(self._h, stage, axis, int(transposed), C.byref(n), C.byref(stride), None, None, None),
                "advx_plan_taps")
        start = np.zeros(n.value, np.int32)
        count = np.zeros(n.value, np.int32)
        w = np.zeros((n.value, stride.value), np.float32)
        L.check(lib.advx_plan_taps(self._h, stage, axis, int(transposed), C.byref(n), C.byref(stride),
                                   start.ctypes.data_as(C.c_void_p), count.ctypes.data_as(C.c_void_p),
                                   w.ctypes.data_as(C.c_void_p)), "advx_plan_taps")
        return start, count, w

    def out_index(self, stage, c, y, x):
        lib = L.load()
        n = C.c_int32()
        idx = (C.c_int64 * 2)()
        L.check(lib.advx_plan_out_index(self._h, stage, c, y, x, C.byref(n), idx), "advx_plan_out_index")
        return [int(idx[i]) for i in range(n.value)]

    def fused_supported(self):
        return bool(L.load().advx_fused_supported(self._h))

    def set_io(self, dtype):
        """Boundary dtype of this plan's pixel_values / their gradient (torch.float32 / float16 / bfloat16)."""
        from .ops import io_code
        L.check(L.load().advx_plan_set_io(self._h, io_code(dtype)), "advx_plan_set_io")
        self.io_dtype = dtype

    def prepared_supported(self):
        return bool(L.load().advx_prepared_supported(self._h))

    def live_range(self):
        """[lo, hi): flat indices of one sample that an image reaches; the rest is constant padding."""
        lo, hi = C.c_int64(), C.c_int64()
        L.check(L.load().advx_plan_live_range(self._h, C.byref(lo), C.byref(hi)), "advx_plan_live_range")
        return int(lo.value), int(hi.value)

    def upload(self, stream=None):
        L.check(L.load().advx_plan_upload(self._h, stream), "advx_plan_upload")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and L._lib is not None:
                L._lib.advx_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def taps_compute(mode, in_size, out_size, transposed=False, device_rows=False, builder=False):
    """Host tap table of an arbitrary 1-D resize (tests, crop windows).  device_rows: as advx_plan_upload stores them
    (zero-weight end taps dropped); builder: the transposed table the way the device-side builder forms it."""
    lib = L.load()
    n, stride = C.c_int32(), C.c_int32()
    flags = (1 if transposed else 0) | (2 if device_rows else 0) | (4 if builder else 0)
    L.check(lib.advx_taps_compute(mode, in_size, out_size, flags, C.byref(n), C.byref(stride), None, None, None),
            "advx_taps_compute")
    start = np.zeros(n.value, np.int32)
    count = np.zeros(n.value, np.int32)
    w = np.zeros((n.value, stride.value), np.float32)
    L.check(lib.advx_taps_compute(mode, in_size, out_size, flags, C.byref(n), C.byref(stride),
                                  start.ctypes.data_as(C.c_void_p), count.ctypes.data_as(C.c_void_p),
                                  w.ctypes.data_as(C.c_void_p)), "advx_taps_compute")
    return start, count, w
