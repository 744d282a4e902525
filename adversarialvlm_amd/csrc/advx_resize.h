// advx_resize.h - the separable resizes with the tap window's size known at compile time.
//
// The kernels of advx_kernels.h walk a thread's tap window with run-time loops: table lookup, then one memory
// round trip per tap, each behind the previous one - image-sized launches that wait for latency, not for
// bytes.  Here the window is T x T with T a template parameter (the tables' row length, chosen per launch by
// the host): every load of a thread is issued before the first use, taps beyond a row's count are read from a
// clamped (valid) address and never enter the sum.  The in-range taps are accumulated by the same operations in
// the same order as in stage_fwd_value / crop_bwd_value / stage_bwd_value, so the results are bit-identical
// (tests/test_gpu_fastpaths.py).  Grids are (column chunks, rows, channels): the row of a workgroup is uniform, its
// tables are scalar loads and no thread divides to find its pixel.  Measured: windows up to 4 x 4 pay;
// from 5 x 5 on, and for the mixed geometries of a multi-plan launch, the run-time loops of advx_kernels.h are faster.
#pragma once
#include "advx_kernels.h"

namespace advx {

constexpr int kRowBlock = 128;    // two waves along x: little waste on 336 / 512 / 560 / 672-wide rows

// sum_a wy[a] * (sum_b wx[b] * src[(y0+a)*rstride + x0+b]),  a < yc, b < xc  (or the transposed nesting)
template <int T>
__device__ inline float gather_window(const float* __restrict__ plane, int rstride, int y0, int yc, int ymax, int x0, int xc,
                                      int xmax, const float* __restrict__ wy, const float* __restrict__ wx, bool inner_h) {
  float r[T][T];
  float wyv[T], wxv[T];
#pragma unroll
  for (int a = 0; a < T; ++a) {
    const float* rowp = plane + (size_t)min(y0 + a, ymax) * rstride;
#pragma unroll
    for (int b = 0; b < T; ++b) r[a][b] = rowp[min(x0 + b, xmax)];
  }
#pragma unroll
  for (int a = 0; a < T; ++a) wyv[a] = (a < yc) ? wy[a] : 0.0f;
#pragma unroll
  for (int b = 0; b < T; ++b) wxv[b] = (b < xc) ? wx[b] : 0.0f;
  float v = 0.0f;
  if (!inner_h) {
#pragma unroll
    for (int a = 0; a < T; ++a) {
      float h = 0.0f;
#pragma unroll
      for (int b = 0; b < T; ++b) h = (b < xc) ? h + wxv[b] * r[a][b] : h;
      v = (a < yc) ? v + wyv[a] * h : v;
    }
  } else {
#pragma unroll
    for (int b = 0; b < T; ++b) {
      float h = 0.0f;
#pragma unroll
      for (int a = 0; a < T; ++a) h = (a < yc) ? h + wyv[a] * r[a][b] : h;
      v = (b < xc) ? v + wxv[b] * h : v;
    }
  }
  return v;
}

// stage_fwd_value with a T x T window
template <int T>
__device__ inline float stage_fwd_value_t(const DStage& st, const float* __restrict__ src, long long src_cstride,
                                          int src_rstride, int c, int y, int x) {
  const int ry = y - st.off_y, rx = x - st.off_x;
  float v;
  if (ry >= 0 && ry < st.res_h && rx >= 0 && rx < st.res_w) {
    v = gather_window<T>(src + (size_t)c * src_cstride, src_rstride, st.th.start[ry], st.th.count[ry], st.src_h - 1,
                         st.tw.start[rx], st.tw.count[rx], st.src_w - 1, st.th.w + (size_t)ry * st.th.stride,
                         st.tw.w + (size_t)rx * st.tw.stride, st.inner_axis_h != 0);
  } else {
    v = st.pad_value;
  }
  if (st.normalise) v = (v - st.mean[c]) / st.stdv[c];
  return v;
}

// k_stage_fwd / k_stage_fwd_img / k_plan_head in one: block (0,0,0) first reduces whatever the previous launch
// left (statistics partials of the image: img_nblk > 0; ||g|| partials: norm_count != 0, < 0 = count in the slot)
template <int T>
__global__ void __launch_bounds__(kRowBlock) k_stage_fwd_t(DStage st, const float* __restrict__ src, long long src_cstride,
                                                           int src_rstride, float* __restrict__ canvas,
                                                           const double* __restrict__ img_partials, int img_nblk,
                                                           long long n_img, const double* __restrict__ norm_rows,
                                                           int norm_count, float* __restrict__ stats) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    if (img_nblk > 0) finalize_image_block<true, kFinImgU>(img_partials, img_nblk, n_img, stats);
    if (norm_count != 0) finalize_norm_block<kFinU>(norm_rows, norm_count < 0 ? (int)norm_rows[kNormCountSlot] : norm_count, stats);
  }
  const int c = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * kRowBlock + threadIdx.x;
  if (x < st.can_w) canvas[((size_t)c * st.can_h + y) * st.can_w + x] = stage_fwd_value_t<T>(st, src, src_cstride, src_rstride, c, y, x);
}

// crop_bwd_value with a T x T window (T = the transposed tables' row length of this step's window)
template <int T>
__device__ inline float crop_bwd_value_t(const DStage& st, const float* __restrict__ gcan, int ci, int cj, int c, int y, int x) {
  const int ys = y - ci, xs = x - cj;
  float v = 0.0f;
  if (ys >= 0 && ys < st.src_h && xs >= 0 && xs < st.src_w) {
    v = gather_window<T>(gcan + (size_t)c * st.can_h * st.can_w, st.can_w, st.tth.start[ys], st.tth.count[ys], st.can_h - 1,
                         st.ttw.start[xs], st.ttw.count[xs], st.can_w - 1, st.tth.w + (size_t)ys * st.tth.stride,
                         st.ttw.w + (size_t)xs * st.ttw.stride, false);
  }
  return v;
}

template <int T>
__global__ void __launch_bounds__(kRowBlock) k_crop_bwd_t(DStage st, const float* __restrict__ gcan, float* __restrict__ gimg,
                                                          int H, int W, int ci, int cj) {
  const int c = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * kRowBlock + threadIdx.x;
  if (x < W) gimg[((size_t)c * H + y) * W + x] = crop_bwd_value_t<T>(st, gcan, ci, cj, c, y, x);
}

// k_crop_bwd on the (column chunk, row, channel) grid with crop_bwd_value's run-time loops: no thread divides to find its
// pixel (the 1-D grid-stride form spends ~35 instructions per element on two run-time divisions), the window row's
// vertical taps are uniform, and only taps that exist are loaded.  These kernels are bound by the instructions a SIMD has
// to issue for its few waves: 10.4 -> 10.1 us at 512 x 512 (the divisions were not the cost; see k_crop_bwd_rows3).
__global__ void __launch_bounds__(kRowBlock) k_crop_bwd_rows(DStage st, const float* __restrict__ gcan, float* __restrict__ gimg,
                                                             int H, int W, int ci, int cj) {
  const int c = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * kRowBlock + threadIdx.x;
  if (x < W) gimg[((size_t)c * H + y) * W + x] = crop_bwd_value(st, gcan, ci, cj, c, y, x);
}

// ... and with the three channels of a pixel in one thread (taps and weights looked up once, three gathers in flight), for
// images of kRows3MinPositions pixels and more (fewer leave the SIMDs too few waves: measured at 113 k)
__global__ void __launch_bounds__(kRowBlock) k_crop_bwd_rows3(DStage st, const float* __restrict__ gcan, float* __restrict__ gimg,
                                                              int H, int W, int ci, int cj) {
  const int y = blockIdx.y;
  const int x = blockIdx.x * kRowBlock + threadIdx.x;
  if (x >= W) return;
  const int ys = y - ci, xs = x - cj;
  float v[3] = {0.0f, 0.0f, 0.0f};
  if (ys >= 0 && ys < st.src_h && xs >= 0 && xs < st.src_w) {
    const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
    const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
    const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
    const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
    const size_t plane = (size_t)st.can_h * st.can_w;
    for (int a = 0; a < oyc; ++a) {
      const float* rowp = gcan + (size_t)(oy + a) * st.can_w + ox;
      const float wa = wy[a];
      float h[3] = {0.0f, 0.0f, 0.0f};
      for (int b = 0; b < oxc; ++b) {
        const float wb = wx[b];
#pragma unroll
        for (int c = 0; c < 3; ++c) h[c] += wb * rowp[(size_t)c * plane + b];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] += wa * h[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) gimg[((size_t)c * H + y) * W + x] = v[c];
}

}  // namespace advx
