// advx_taps.h - 1-D resampling tap tables, computed identically on host and device.
//
// Restates the arithmetic of ATen's F.interpolate kernels that the reference calls
// (llavaprocessor.py:143, llama32processor.py:284, qwen2VLprocessor.py:166,
// phi3processor.py:194,220; torchvision RandomResizedCrop at attack_model.py:198-202):
//   AA_BILINEAR : UpSampleKernel.cpp `_compute_indices_min_size_weights_aa`, triangle filter
//   BILINEAR    : `HelperInterpLinear` (align_corners=False)
//   BICUBIC     : `HelperInterpCubic`, Keys kernel A=-0.75, border-clamped indices
// ATen's scalar_t is float but literals like 0.5 are doubles, so some sub-expressions are
// evaluated in double and rounded once; the same mix is kept here.
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef ADVX_HD
#define ADVX_HD __host__ __device__ inline
#endif

#define ADVX_MODE_AA_BILINEAR 0
#define ADVX_MODE_BILINEAR 1
#define ADVX_MODE_BICUBIC 2

namespace advx {

struct TapRow {
  int start;
  int count;
};

ADVX_HD float tap_scale(int in_size, int out_size) { return (float)in_size / (float)out_size; }

// number of weights reserved per output row
ADVX_HD int tap_stride(int mode, int in_size, int out_size) {
  if (mode == ADVX_MODE_BILINEAR) return 2;
  if (mode == ADVX_MODE_BICUBIC) return 4;
  float scale = tap_scale(in_size, out_size);
  float support = (scale >= 1.0f) ? (float)(1.0 * (double)scale) : 1.0f;
  return (int)ceilf(support) * 2 + 1;
}

// [start, start+count) of source indices read by output i
ADVX_HD TapRow tap_bounds(int mode, int in_size, int out_size, int i) {
  TapRow r;
  float scale = tap_scale(in_size, out_size);
  if (mode == ADVX_MODE_AA_BILINEAR) {
    float support = (scale >= 1.0f) ? (float)(1.0 * (double)scale) : 1.0f;
    int max_taps = (int)ceilf(support) * 2 + 1;
    float center = (float)((double)scale * ((double)i + 0.5));
    long xmin = (long)((double)(center - support) + 0.5);
    if (xmin < 0) xmin = 0;
    long xmax = (long)((double)(center + support) + 0.5);
    if (xmax > in_size) xmax = in_size;
    long xsize = xmax - xmin;
    if (xsize < 0) xsize = 0;
    if (xsize > max_taps) xsize = max_taps;
    r.start = (int)xmin;
    r.count = (int)xsize;
    return r;
  }
  if (mode == ADVX_MODE_BILINEAR) {
    if (in_size == out_size) {
      r.start = i;
      r.count = 1;
      return r;
    }
    float src = (float)((double)scale * ((double)i + 0.5) - 0.5);
    if (src < 0.0f) src = 0.0f;
    int i0 = (int)floorf(src);
    if (i0 > in_size - 1) i0 = in_size - 1;
    r.start = i0;
    r.count = (i0 < in_size - 1) ? 2 : 1;
    return r;
  }
  // bicubic
  float src = (float)((double)scale * ((double)i + 0.5) - 0.5);
  int i0 = (int)floorf(src);
  if (i0 > in_size - 1) i0 = in_size - 1;
  int lo = i0 - 1, hi = i0 + 2;
  if (lo < 0) lo = 0;
  if (lo > in_size - 1) lo = in_size - 1;
  if (hi < 0) hi = 0;
  if (hi > in_size - 1) hi = in_size - 1;
  r.start = lo;
  r.count = hi - lo + 1;
  return r;
}

ADVX_HD float cubic1(float x, float A) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; }
ADVX_HD float cubic2(float x, float A) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; }

// weights of output row i into w[0..stride); returns the row bounds. Entries >= count are 0.
ADVX_HD TapRow tap_row(int mode, int in_size, int out_size, int i, int stride, float* w) {
  TapRow r = tap_bounds(mode, in_size, out_size, i);
  for (int k = 0; k < stride; ++k) w[k] = 0.0f;
  float scale = tap_scale(in_size, out_size);
  if (mode == ADVX_MODE_AA_BILINEAR) {
    float support = (scale >= 1.0f) ? (float)(1.0 * (double)scale) : 1.0f;
    float invscale = (scale >= 1.0f) ? (float)(1.0 / (double)scale) : 1.0f;
    float center = (float)((double)scale * ((double)i + 0.5));
    (void)support;
    float total = 0.0f;
    for (int j = 0; j < r.count && j < stride; ++j) {
      float d = (float)(j + r.start) - center;
      float arg = (float)(((double)d + 0.5) * (double)invscale);
      float a = fabsf(arg);
      float wt = (a < 1.0f) ? (1.0f - a) : 0.0f;
      w[j] = wt;
      total += wt;
    }
    if (total != 0.0f) {
      float norm = (float)(1.0 / (double)total);
      for (int j = 0; j < r.count && j < stride; ++j) w[j] *= norm;
    }
    return r;
  }
  if (mode == ADVX_MODE_BILINEAR) {
    if (in_size == out_size) {
      w[0] = 1.0f;
      return r;
    }
    float src = (float)((double)scale * ((double)i + 0.5) - 0.5);
    if (src < 0.0f) src = 0.0f;
    float lam1 = src - (float)r.start;
    lam1 = fminf(fmaxf(lam1, 0.0f), 1.0f);
    float lam0 = 1.0f - lam1;
    if (r.count == 1) {
      w[0] = lam0 + lam1;
    } else {
      w[0] = lam0;
      w[1] = lam1;
    }
    return r;
  }
  // bicubic: clamped duplicates are merged onto their source index
  const float A = -0.75f;
  float src = (float)((double)scale * ((double)i + 0.5) - 0.5);
  int i0 = (int)floorf(src);
  if (i0 > in_size - 1) i0 = in_size - 1;
  float t = src - (float)i0;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  float coeff[4];
  coeff[0] = cubic2(t + 1.0f, A);
  coeff[1] = cubic1(t, A);
  coeff[2] = cubic1(1.0f - t, A);
  coeff[3] = cubic2(2.0f - t, A);
  for (int k = 0; k < 4; ++k) {
    int idx = i0 - 1 + k;
    if (idx < 0) idx = 0;
    if (idx > in_size - 1) idx = in_size - 1;
    int slot = idx - r.start;
    if (slot >= 0 && slot < stride) w[slot] += coeff[k];
  }
  return r;
}

// weight of source index j in output row i (tap_row(...)[j - start], 0 outside the row): the same operations in the
// same order as tap_row, without a row buffer the caller would index at run time
ADVX_HD float tap_weight(int mode, int in_size, int out_size, int i, int j, int stride) {
  if (mode == ADVX_MODE_AA_BILINEAR) {
    TapRow r = tap_bounds(mode, in_size, out_size, i);
    float scale = tap_scale(in_size, out_size);
    float invscale = (scale >= 1.0f) ? (float)(1.0 / (double)scale) : 1.0f;
    float center = (float)((double)scale * ((double)i + 0.5));
    float total = 0.0f, wj = 0.0f;
    for (int q = 0; q < r.count && q < stride; ++q) {
      float d = (float)(q + r.start) - center;
      float arg = (float)(((double)d + 0.5) * (double)invscale);
      float a = fabsf(arg);
      float wt = (a < 1.0f) ? (1.0f - a) : 0.0f;
      total += wt;
      if (q + r.start == j) wj = wt;
    }
    if (total != 0.0f) wj *= (float)(1.0 / (double)total);
    return wj;
  }
  float w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  TapRow r = tap_row(mode, in_size, out_size, i, stride < 4 ? stride : 4, w);
  int slot = j - r.start;
  int lim = stride < 4 ? stride : 4;
  if (slot < 0 || slot >= lim) return 0.0f;
  return slot == 0 ? w[0] : slot == 1 ? w[1] : slot == 2 ? w[2] : w[3];
}

// Outputs that read source index j form a contiguous range because start(i) and
// start(i)+count(i) are both non-decreasing in i: [first i with end(i) > j, last i with
// start(i) <= j].  Returned as (start, count) over OUTPUT indices (count may be 0).
ADVX_HD TapRow tap_bounds_transposed(int mode, int in_size, int out_size, int j) {
  int lo = 0, hi = out_size;  // first i with end(i) > j
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    TapRow r = tap_bounds(mode, in_size, out_size, mid);
    if (r.start + r.count > j) hi = mid; else lo = mid + 1;
  }
  int first = lo;
  lo = 0;
  hi = out_size;  // first i with start(i) > j
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    TapRow r = tap_bounds(mode, in_size, out_size, mid);
    if (r.start > j) hi = mid; else lo = mid + 1;
  }
  int last = lo - 1;
  TapRow t;
  t.start = first;
  t.count = last - first + 1;
  if (t.count < 0) t.count = 0;
  if (t.count == 0) t.start = 0;
  return t;
}

}  // namespace advx
