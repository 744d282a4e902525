"""Explicit tap-table restatement of the three `F.interpolate` modes the reference calls
(oracle side; test infrastructure only).

  * aa_bilinear : `mode='bilinear', align_corners=False, antialias=True`
                  (llavaprocessor.py:143, llama32processor.py:284, qwen2VLprocessor.py:166,
                  and torchvision's RandomResizedCrop at attack_model.py:198-202,310)
  * bilinear    : `mode='bilinear'` without antialias (phi3processor.py:194)
  * bicubic     : `mode='bicubic'` (phi3processor.py:220), Keys kernel A = -0.75

The algorithm is third-party (ATen, torch 2.6 pinned by the reference, 2.10 here); it is
restated from its published semantics and checked in
tests/test_oracle_resample.py against `F.interpolate` itself.  Tables are computed in
float32 with the same operation order as ATen so that they can be compared with the
tables `advx_plan_create` builds on the host.

A table for one axis is (start[out], count[out], weight[out, max_taps]) :
    out[i] = sum_{k < count[i]} weight[i, k] * in[start[i] + k]
"""
import numpy as np

f32 = np.float32


def aa_bilinear_taps(in_size, out_size):
    """ATen `_compute_indices_min_size_weights_aa` with the triangle filter.

    scalar_t is float; literals such as 0.5 are C doubles, so several sub-expressions are
    evaluated in double and rounded to float once - mirrored here with Python floats.
    """
    scale = f32(in_size) / f32(out_size)
    if scale >= f32(1.0):
        support = f32(1.0 * float(scale))   # (interp_size * 0.5) * scale
        invscale = f32(1.0 / float(scale))
    else:
        support = f32(1.0)
        invscale = f32(1.0)
    max_taps = int(np.ceil(support)) * 2 + 1
    start = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    weight = np.zeros((out_size, max_taps), f32)
    for i in range(out_size):
        center = f32(float(scale) * (i + 0.5))
        xmin = max(int(float(f32(center - support)) + 0.5), 0)
        xsize = min(int(float(f32(center + support)) + 0.5), in_size) - xmin
        xsize = max(min(xsize, max_taps), 0)
        total = f32(0.0)
        ws = np.zeros(max_taps, f32)
        for j in range(xsize):
            arg = f32((float(f32(f32(j + xmin) - center)) + 0.5) * float(invscale))
            a = abs(arg)
            w = f32(f32(1.0) - a) if a < f32(1.0) else f32(0.0)
            ws[j] = w
            total = f32(total + w)
        if total != f32(0.0):
            norm = f32(1.0 / float(total))
            for j in range(xsize):
                ws[j] = f32(ws[j] * norm)
        start[i], count[i] = xmin, xsize
        weight[i] = ws
    return start, count, weight


def bilinear_taps(in_size, out_size):
    """ATen `HelperInterpLinear` (align_corners=False): identity shortcut when sizes are
    equal, else src = scale*(i+0.5)-0.5 clamped at 0, floor, lambda clamped to [0,1]."""
    start = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    weight = np.zeros((out_size, 2), f32)
    if in_size == out_size:
        start[:] = np.arange(out_size)
        count[:] = 1
        weight[:, 0] = 1.0
        return start, count, weight
    scale = f32(in_size) / f32(out_size)
    for i in range(out_size):
        src = f32(float(scale) * (i + 0.5) - 0.5)
        if src < f32(0.0):
            src = f32(0.0)
        i0 = min(int(np.floor(src)), in_size - 1)
        lam1 = f32(min(max(f32(src - f32(i0)), f32(0.0)), f32(1.0)))
        lam0 = f32(f32(1.0) - lam1)
        i1 = i0 + 1 if i0 < in_size - 1 else i0
        start[i] = i0
        if i1 == i0:
            count[i] = 1
            weight[i, 0] = f32(lam0 + lam1)
        else:
            count[i] = 2
            weight[i, 0], weight[i, 1] = lam0, lam1
    return start, count, weight


def _cubic1(x, A):
    return ((A + f32(2)) * x - (A + f32(3))) * x * x + f32(1)


def _cubic2(x, A):
    return ((A * x - f32(5) * A) * x + f32(8) * A) * x - f32(4) * A


def bicubic_taps(in_size, out_size):
    """ATen `upsample_bicubic2d`: src = scale*(i+0.5)-0.5 (NOT clamped), floor, 4 taps
    with border-clamped indices.  Clamped duplicates are merged into one weight per
    distinct source index so the table keeps the (start, count, weight) form."""
    A = f32(-0.75)
    start = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    weight = np.zeros((out_size, 4), f32)
    scale = f32(in_size) / f32(out_size)
    for i in range(out_size):
        src = f32(float(scale) * (i + 0.5) - 0.5)
        i0 = min(int(np.floor(src)), in_size - 1)
        t = f32(min(max(f32(src - f32(i0)), f32(0.0)), f32(1.0)))
        coeffs = [_cubic2(t + f32(1), A), _cubic1(t, A),
                  _cubic1(f32(1) - t, A), _cubic2(f32(2) - t, A)]
        idx = [min(max(i0 - 1 + k, 0), in_size - 1) for k in range(4)]
        lo, hi = idx[0], idx[3]
        start[i] = lo
        count[i] = hi - lo + 1
        for k in range(4):
            weight[i, idx[k] - lo] = f32(weight[i, idx[k] - lo] + f32(coeffs[k]))
    return start, count, weight


def taps_to_matrix(taps, in_size):
    start, count, weight = taps
    M = np.zeros((len(start), in_size), f32)
    for i in range(len(start)):
        for k in range(count[i]):
            M[i, start[i] + k] += weight[i, k]
    return M


def transpose_taps(taps, in_size):
    """For each INPUT index the contiguous range of outputs that read it and the weights
    (the table a gather-style backward uses):  gin[j] = sum_k tw[j,k] * gout[ts[j]+k]."""
    start, count, weight = taps
    out_size = len(start)
    lo = np.full(in_size, out_size, np.int64)
    hi = np.full(in_size, -1, np.int64)
    for i in range(out_size):
        for k in range(count[i]):
            j = start[i] + k
            lo[j] = min(lo[j], i)
            hi[j] = max(hi[j], i)
    tcount = np.maximum(hi - lo + 1, 0).astype(np.int32)
    max_t = max(int(tcount.max()), 1)
    tstart = np.where(tcount > 0, lo, 0).astype(np.int32)
    tw = np.zeros((in_size, max_t), f32)
    for i in range(out_size):
        for k in range(count[i]):
            j = start[i] + k
            tw[j, i - tstart[j]] += weight[i, k]
    return tstart, tcount, tw


def resize_separable(img, taps_h, taps_w):
    """img [C,H,W] float32 -> [C,oh,ow]; horizontal pass first, then vertical (the order
    of ATen's separable CPU kernel), fp32 accumulation in tap order."""
    C, H, W = img.shape
    sw, cw, ww = taps_w
    sh, ch, wh = taps_h
    tmp = np.zeros((C, H, len(sw)), f32)
    for x in range(len(sw)):
        acc = np.zeros((C, H), f32)
        for k in range(cw[x]):
            acc = (acc + ww[x, k] * img[:, :, sw[x] + k]).astype(f32)
        tmp[:, :, x] = acc
    out = np.zeros((C, len(sh), len(sw)), f32)
    for y in range(len(sh)):
        acc = np.zeros((C, len(sw)), f32)
        for k in range(ch[y]):
            acc = (acc + wh[y, k] * tmp[:, sh[y] + k, :]).astype(f32)
        out[:, y, :] = acc
    return out
