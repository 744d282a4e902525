// advx_blur.h - the Gaussian-blur kernels of the trainers' hot chains, radius known at compile time.
//
// k_blur (advx_kernels.h) handles any radius up to 15 with run-time loops; the reference's presets use
// kernel sizes 5 and 9 (attack_model.py:510, attack_clamp_tanh_llava_gblur.sh:37, attack_cross.sh).  With the
// radius a template parameter
//   * every thread forms the k normalised weights in registers (same expressions, same order as k_blur: no
//     LDS round trip, no section in which one thread normalises for the block),
//   * the tile loads are a fixed number per thread, issued back to back before the first use (k_blur's
//     run-time loop pays one memory round trip per element),
//   * the per-pixel state of the epilogue is fetched before the two LDS passes.
// Element for element the arithmetic is k_blur's (and k_crop_bwd's, k_bwd_update<1>'s, k_tanh_bwd<true>'s): the
// results are bit-identical, which the GPU tests check against the generic kernels.
#pragma once
#include "advx_kernels.h"

namespace advx {

constexpr int kBlurFastMaxR = 7;   // kernel sizes 3..15

template <int R>
__device__ inline void blur_weights(float sigma, float (&w)[2 * R + 1]) {
  constexpr int K = 2 * R + 1;
  // torchvision _get_gaussian_kernel1d: exp(-0.5 (t/sigma)^2) / sum
#pragma unroll
  for (int i = 0; i < K; ++i) {
    float t = (float)(i - R);
    float q = t / sigma;
    w[i] = expf(-0.5f * (q * q));
  }
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < K; ++i) sum += w[i];
#pragma unroll
  for (int i = 0; i < K; ++i) w[i] = w[i] / sum;
}

// The same weights formed ONCE per workgroup: thread t < K computes the k raw weights and their sum in registers and
// publishes its own normalised one; everybody reads them back after the workgroup's next barrier.  (Per thread the k
// exponentials, the sum and the k divisions are ~250 instructions, a tenth of what a tile costs a thread.)
template <int R>
__device__ inline void blur_weights_publish(float sigma, float* __restrict__ wn) {
  constexpr int K = 2 * R + 1;
  if (threadIdx.x < K) {
    float w[K];
    blur_weights<R>(sigma, w);
    float mine = 0.0f;
#pragma unroll
    for (int i = 0; i < K; ++i)
      if (i == (int)threadIdx.x) mine = w[i];
    wn[threadIdx.x] = mine;
  }
}
template <int R>
__device__ inline void blur_weights_fetch(const float* __restrict__ wn, float (&w)[2 * R + 1]) {
#pragma unroll
  for (int i = 0; i < 2 * R + 1; ++i) w[i] = wn[i];
}

// ------------------------------------------------------------------------------ forward
// k_blur<0, IN> with x0 epilogue: s = x0 + blur(IN == 1 ? eps*tanh(in) : in), statistics partials per tile.
// blockIdx.z == 3: tap-table builder blocks riding in the launch (as in k_blur<0, 1>).
template <int IN, int R, int BT = kBlock>   // BT: threads per workgroup; 512 measured SLOWER here at radius 4 (10.7 -> 13.7 us: 3.1 / 2.5 elements per thread leave a quarter of the slots idle), level at radius 2
__global__ void __launch_bounds__(BT) k_blur_fwd_r(const float* __restrict__ in, int H, int W, float sigma,
                                                       const float* __restrict__ x0, float* __restrict__ out,
                                                       double* __restrict__ partials, float scalar, TapBuild taps0,
                                                       TapBuild taps1, int tap_blocks) {
  if (blockIdx.z == 3) {
    const int tb = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    if (tb < 2 * tap_blocks) {
      const int axis = tb / tap_blocks;
      // a branch per axis, not a select between the two argument structs: selecting makes the compiler copy both to scratch
      if (axis == 0) build_taps_row_c(taps0, tb * (int)blockDim.x + (int)threadIdx.x);
      else build_taps_row_c(taps1, (tb - tap_blocks) * (int)blockDim.x + (int)threadIdx.x);
    }
    return;
  }
  constexpr int K = 2 * R + 1, TS = kBlurTile + 2 * R, NE = TS * TS, NL = (NE + BT - 1) / BT;
  constexpr int NR = (TS * kBlurTile + BT - 1) / BT, NO = kBlurTile * kBlurTile / BT;
  __shared__ float tile[TS][TS + 1];
  __shared__ float tmp[TS][kBlurTile + 1];
  __shared__ float wn[K];
  blur_weights_publish<R>(sigma, wn);
  const int c = blockIdx.z;
  const int oy0 = blockIdx.y * kBlurTile, ox0 = blockIdx.x * kBlurTile;
  const float* src = in + (size_t)c * H * W;
  float v[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int e = (int)threadIdx.x + j * BT;
    v[j] = 0.0f;
    if (e < NE) {
      const int ty = e / TS, tx = e - ty * TS;
      v[j] = src[(size_t)reflect_index(oy0 - R + ty, H) * W + reflect_index(ox0 - R + tx, W)];
    }
  }
  // the epilogue's x0 (four pixels per thread) travels while the tile is transformed and filtered
  float xv[NO];
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int e = (int)threadIdx.x + j * BT;
    const int oy = oy0 + e / kBlurTile, ox = ox0 + (e & (kBlurTile - 1));
    xv[j] = (x0 != nullptr && oy < H && ox < W) ? x0[((size_t)c * H + oy) * W + ox] : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int e = (int)threadIdx.x + j * BT;
    if (e < NE) {
      const int ty = e / TS, tx = e - ty * TS;
      float t = v[j];
      if (IN == 1) t = scalar * tanhf(t);
      tile[ty][tx] = t;
    }
  }
  __syncthreads();
  float w[K];
  blur_weights_fetch<R>(wn, w);
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int e = (int)threadIdx.x + j * BT;
    if (e < TS * kBlurTile) {
      const int ty = e / kBlurTile, x = e & (kBlurTile - 1);
      float a = 0.0f;
#pragma unroll
      for (int t = 0; t < K; ++t) a += w[t] * tile[ty][x + t];
      tmp[ty][x] = a;
    }
  }
  __syncthreads();
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int e = (int)threadIdx.x + j * BT;
    const int y = e / kBlurTile, x = e & (kBlurTile - 1);
    const int oy = oy0 + y, ox = ox0 + x;
    if (oy < H && ox < W) {
      float a = 0.0f;
#pragma unroll
      for (int t = 0; t < K; ++t) a += w[t] * tmp[y + t][x];
      const size_t o = ((size_t)c * H + oy) * W + ox;
      if (x0 != nullptr) {
        const float s = xv[j] + a;
        out[o] = s;
        stat_accumulate(s, a, acc);
      } else {
        out[o] = a;
      }
    }
  }
  if (partials != nullptr) {
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_sum_store<kStatSlots, BT>(acc, partials + blk * kStatSlots);
  }
}

// ----------------------------------------------------------------------------- backward
// The image-level backward behind the plans' transposed resizes, in ONE launch:
//   + imgfit'(s)  ->  blur adjoint = zero-padded correlation on the extended domain (k_blur<1, 2>) and fold of
//   the reflected border (blur_fold)  ->
//   tanh', [accumulate], and UPDATE: mask, ||g|| partial, optimiser (k_bwd_update<1>) | else the unmasked
//   gradient (k_tanh_bwd<true>, data parallelism: the all-reduce follows).
// One block = one 32x32 tile of the IMAGE.  It computes the correlation on the window of the extended
// domain its pixels fold from: its own rows, rows -R..-1 if it holds rows 1..R, rows up to H-1+R if it holds
// rows of [H-1-R, H-2] (same for columns) - 32^2 outputs from (32+2R)^2 inputs inside the image, at most
// (32+3R)^2 from (32+5R)^2 where both borders fold into one tile; zeros outside the image.  No extended-domain
// buffer in memory, no second pass over the image.  (A crop window's transposed resize stays a launch of its
// own, k_crop_bwd_t: gathered while the tiles load it cost 36 us instead of 10.6 + 14.)
// Body of k_blur_bwd_fused.  INTERIOR: the tile and its 2R halo lie strictly inside the image and none of its pixels
// folds (77 % of the tiles at 512^2, k = 9): the window is the tile itself, 32 + 2R inputs per side with a compile-time
// row length, no bounds tests, seven loads per thread instead of eleven slots, one c2 term per pixel.  Same operations
// per element as the general form.
template <int R, bool UPDATE, bool INTERIOR, int BT>
__device__ inline void blur_bwd_body(const float* __restrict__ gsrc, const float* __restrict__ s, int H, int W, float eps,
                                     float c_fit, int accumulate, float* __restrict__ p, float* __restrict__ m,
                                     float* __restrict__ v, float* __restrict__ grad, const float* __restrict__ mask,
                                     const OptScalars& o, double* __restrict__ partials, const float* __restrict__ wn,
                                     float (*tile)[kBlurTile + 5 * R + 1], float (*tmp)[kBlurTile + 3 * R + 1],
                                     float (*c2t)[kBlurTile + 3 * R + 1]) {
  constexpr int K = 2 * R + 1, TI = kBlurTile + 5 * R;
  constexpr int TC = kBlurTile + 2 * R;                       // interior: inputs per side
  constexpr int NL = INTERIOR ? (TC * TC + BT - 1) / BT : (TI * TI + BT - 1) / BT;
  constexpr int NO = kBlurTile * kBlurTile / BT;
  const int c = blockIdx.z;
  const int oy0 = blockIdx.y * kBlurTile, ox0 = blockIdx.x * kBlurTile;
  const int ty_hi = min(oy0 + kBlurTile, H) - 1, tx_hi = min(ox0 + kBlurTile, W) - 1;
  // output window of the extended domain (inclusive bounds)
  const int wy0 = INTERIOR ? oy0 : ((oy0 == 0) ? -R : oy0);
  const int wx0 = INTERIOR ? ox0 : ((ox0 == 0) ? -R : ox0);
  const int wy1 = INTERIOR ? oy0 + kBlurTile - 1
                           : ((ty_hi >= H - 1 - R && oy0 <= H - 2) ? 2 * (H - 1) - max(oy0, H - 1 - R) : ty_hi);
  const int wx1 = INTERIOR ? ox0 + kBlurTile - 1
                           : ((tx_hi >= W - 1 - R && ox0 <= W - 2) ? 2 * (W - 1) - max(ox0, W - 1 - R) : tx_hi);
  const int wh = INTERIOR ? kBlurTile : wy1 - wy0 + 1, ww = INTERIOR ? kBlurTile : wx1 - wx0 + 1;
  const int ih = INTERIOR ? TC : wh + 2 * R, iw = INTERIOR ? TC : ww + 2 * R;
  const size_t plane = (size_t)H * W;
  // this thread's pixels: state fetched first, used last
  float pp[NO], mk[NO], mm[NO], vv[NO], g0[NO];
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int e = (int)threadIdx.x + j * BT;
    const int y = oy0 + e / kBlurTile, x = ox0 + (e & (kBlurTile - 1));
    pp[j] = mk[j] = mm[j] = vv[j] = g0[j] = 0.0f;
    if (INTERIOR || (y < H && x < W)) {
      const size_t i = (size_t)c * plane + (size_t)y * W + x;
      pp[j] = p[i];
      if (accumulate) g0[j] = grad[i];
      if (UPDATE) {
        mk[j] = mask[i];
        if (o.apply && o.kind == 0) {
          mm[j] = m[i];
          vv[j] = v[i];
        }
      }
    }
  }
  // input tile: (gradient w.r.t. s) + imgfit'(s) inside the image, zero outside
  const unsigned magic = INTERIOR ? 0u : (unsigned)((0x100000000ULL + (unsigned)iw - 1) / (unsigned)iw);   // e / iw, e < 2^16
  const int ne = ih * iw;
  float gv[NL], sv[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int e = (int)threadIdx.x + j * BT;
    gv[j] = 0.0f;
    sv[j] = 0.0f;
    if (e < ne) {
      const int iy = INTERIOR ? e / TC : (int)__umulhi((unsigned)e, magic), ix = e - iy * iw;
      const int gy = wy0 - R + iy, gx = wx0 - R + ix;
      if (INTERIOR || (gy >= 0 && gy < H && gx >= 0 && gx < W)) {
        const size_t oidx = (size_t)c * plane + (size_t)gy * W + gx;
        gv[j] = gsrc[oidx];
        sv[j] = s[oidx];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int e = (int)threadIdx.x + j * BT;
    if (e < ne) {
      const int iy = INTERIOR ? e / TC : (int)__umulhi((unsigned)e, magic), ix = e - iy * iw;
      const int gy = wy0 - R + iy, gx = wx0 - R + ix;
      float t = 0.0f;
      if (INTERIOR || (gy >= 0 && gy < H && gx >= 0 && gx < W)) t = gv[j] + imgfit_grad(sv[j], c_fit);
      tile[iy][ix] = t;
    }
  }
  __syncthreads();
  float w[K];
  blur_weights_fetch<R>(wn, w);
  {
    // e / ww by multiplication (exact for e < 2^16); a one-column window divides by one
    const unsigned mw = (!INTERIOR && ww > 1) ? (unsigned)((0x100000000ULL + (unsigned)ww - 1) / (unsigned)ww) : 0u;
    for (int e = threadIdx.x; e < ih * ww; e += BT) {
      const int iy = INTERIOR ? e / kBlurTile : ((ww > 1) ? (int)__umulhi((unsigned)e, mw) : e), x = e - iy * ww;
      float a = 0.0f;
#pragma unroll
      for (int t = 0; t < K; ++t) a += w[t] * tile[iy][x + t];
      tmp[iy][x] = a;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < wh * ww; e += BT) {
      const int y = INTERIOR ? e / kBlurTile : ((ww > 1) ? (int)__umulhi((unsigned)e, mw) : e), x = e - y * ww;
      float a = 0.0f;
#pragma unroll
      for (int t = 0; t < K; ++t) a += w[t] * tmp[y + t][x];
      c2t[y][x] = a;
    }
  }
  __syncthreads();
  double nacc[1] = {0.0};
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int e = (int)threadIdx.x + j * BT;
    const int y = oy0 + e / kBlurTile, x = ox0 + (e & (kBlurTile - 1));
    if (INTERIOR || (y < H && x < W)) {
      float gx = 0.0f;
      if (INTERIOR) {
        gx += c2t[y - wy0][x - wx0];              // 0 + c2: what the one-term fold loop forms
      } else {
        // blur_fold: own position, the reflection about the first row / column, about the last
        int yy[3], xx[3], ny = 0, nx = 0;
        yy[ny++] = y;
        if (y >= 1 && y <= R) yy[ny++] = -y;
        if (y <= H - 2 && y >= H - 1 - R) yy[ny++] = 2 * (H - 1) - y;
        xx[nx++] = x;
        if (x >= 1 && x <= R) xx[nx++] = -x;
        if (x <= W - 2 && x >= W - 1 - R) xx[nx++] = 2 * (W - 1) - x;
        for (int a = 0; a < ny; ++a)
          for (int b = 0; b < nx; ++b) gx += c2t[yy[a] - wy0][xx[b] - wx0];
      }
      const size_t i = (size_t)c * plane + (size_t)y * W + x;
      float pv = pp[j];
      const float t = tanhf(pv);
      float g = (gx * eps) * (1.0f - t * t);
      if (accumulate) g = g0[j] + g;
      if (UPDATE) {
        g = g * mk[j];                          // attack_model.py:336
        grad[i] = g;
        nacc[0] += (double)g * (double)g;
        if (o.apply) {
          if (o.kind == 0) {
            float m1 = mm[j], v1 = vv[j];
            adamw_element(pv, m1, v1, g, o);
            p[i] = pv; m[i] = m1; v[i] = v1;
          } else {
            float sg = sign_direction(g);
            p[i] = pv - o.lr * sg;
          }
        }
      } else {
        grad[i] = g;
      }
    }
  }
  if (UPDATE) {
    const int blk = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_sum_store<1, BT>(nacc, partials + blk);
    if (blk == 0 && threadIdx.x == 0) partials[kNormCountSlot] = (double)(gridDim.x * gridDim.y * gridDim.z);
  }
}

// BT: threads per tile.  512 (eight waves: half the serial work per thread, 72 instead of 90 VGPRs) is the host's default for
// radius <= 4: 14.5 -> 13.3-14.1 us at radius 4, 10.5 -> 9.0-9.5 at radius 2 (round 4)
template <int R, bool UPDATE, int BT = kBlock>
__global__ void __launch_bounds__(BT) k_blur_bwd_fused(const float* __restrict__ gsrc, const float* __restrict__ s,
                                                           int H, int W, float sigma,
                                                           float eps, float c_fit, int accumulate, float* __restrict__ p,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ grad, const float* __restrict__ mask,
                                                           OptScalars o, double* __restrict__ partials) {
  // worst case: tile 0 of an image of 32 < H <= 32 + R rows folds from both borders: H + 2R <= 32 + 3R output rows
  constexpr int K = 2 * R + 1, TI = kBlurTile + 5 * R, TO = kBlurTile + 3 * R;
  __shared__ float tile[TI][TI + 1];
  __shared__ float tmp[TI][TO + 1];
  __shared__ float c2t[TO][TO + 1];
  __shared__ float wn[K];
  blur_weights_publish<R>(sigma, wn);
  const int oy0 = blockIdx.y * kBlurTile, ox0 = blockIdx.x * kBlurTile;
  // interior: the tile plus its 2R halo strictly inside the image, no pixel of it within R of the last row / column
  const bool interior = oy0 >= kBlurTile && ox0 >= kBlurTile && oy0 + kBlurTile - 1 + 2 * R < H - 1 &&
                        ox0 + kBlurTile - 1 + 2 * R < W - 1;
  if (interior)
    blur_bwd_body<R, UPDATE, true, BT>(gsrc, s, H, W, eps, c_fit, accumulate, p, m, v, grad, mask, o, partials, wn, tile, tmp, c2t);
  else
    blur_bwd_body<R, UPDATE, false, BT>(gsrc, s, H, W, eps, c_fit, accumulate, p, m, v, grad, mask, o, partials, wn, tile, tmp, c2t);
}

// ------------------------------------------------------- backward of step t + forward blur of step t+1
// k_blur_step = k_blur_bwd_fused<R, true>(t) and k_blur_fwd_r<1, R>(t+1) in ONE launch, by halo recompute (round 4).
// The forward blur of a 32x32 tile reads x_{t+1} = eps*tanh(p_{t+1}) on the tile and its R-halo (reflected at the image
// border, i.e. always pixels of the image within R of the tile).  So every block computes the gradient and the optimiser
// update for the pixel range P = tile +- R (clipped to the image) - from the correlation on the window of the extended
// domain P's pixels fold from, i.e. inputs on P +- R (+ the fold extension) - and only the OWNER of a pixel (the tile that
// contains it) stores p, m, v, grad and counts ||g||.  The redundant arithmetic is the owner's arithmetic on the same
// inputs in the same order (the two LDS passes sum taps in ascending order wherever a position is computed), so a halo
// pixel's updated p - and eps*tanh of it - carries the owner's bits.  Then the block blurs x_{t+1} for its own tile,
// adds x0, stores s_{t+1} (into ANOTHER buffer than s_t: neighbours still read s_t for their halos) and leaves the
// statistics partials of s_{t+1} exactly as k_blur_fwd_r would (same thread -> pixel map, same reduction).
// Owner pixels sit in the first four slots of a thread with k_blur_bwd_fused's thread -> pixel map, so the ||g|| partial
// of a block is summed in the same order: the launch leaves the bits of the two launches it replaces.
// p, m, v are read from the step-t buffers and written to OTHER buffers (the caller ping-pongs): a neighbour that
// recomputes a halo pixel must read the owner's OLD p, whenever the owner's block happens to run.
// Requires min(H, W) >= 32 + 3R + 2 (one reflected border per pixel range: bounds the LDS window), no accumulation
// window, one rank.  blockIdx.z == 3: tap-table builder blocks of the NEXT step's crop window (as in k_blur_fwd_r).
template <int R, bool INTERIOR>
__device__ inline void blur_step_body(const float* __restrict__ gsrc, const float* __restrict__ s, int H, int W, float eps,
                                      float c_fit, const float* __restrict__ p, const float* __restrict__ m,
                                      const float* __restrict__ v, float* __restrict__ p_out, float* __restrict__ m_out,
                                      float* __restrict__ v_out, float* __restrict__ grad, const float* __restrict__ mask,
                                      const OptScalars& o,
                                      double* __restrict__ norm_partials, const float* __restrict__ x0,
                                      float* __restrict__ s_next, double* __restrict__ img_partials,
                                      const float* __restrict__ wn_b, const float* __restrict__ wn_f,
                                      float (*tile)[kBlurTile + 6 * R + 1], float (*tmp)[kBlurTile + 4 * R + 1],
                                      float (*c2t)[kBlurTile + 4 * R + 1]) {
  constexpr int K = 2 * R + 1;
  constexpr int TP = kBlurTile + 2 * R;                 // pixel range per side (interior: exactly)
  constexpr int TI = kBlurTile + 6 * R;                 // general: inputs per side at most (window <= TP + 2R, + 2R)
  constexpr int TC = TP + 2 * R;                        // interior: inputs per side
  constexpr int NL = INTERIOR ? (TC * TC + kBlock - 1) / kBlock : (TI * TI + kBlock - 1) / kBlock;
  constexpr int NO = kBlurTile * kBlurTile / kBlock;    // owner slots
  constexpr int NRING = TP * TP - kBlurTile * kBlurTile, NH = (NRING + kBlock - 1) / kBlock;
  constexpr int NS = NO + NH;
  constexpr int TS = TP;                                // forward tile: 32 + 2R extended positions per side
  const int c = blockIdx.z;
  const int oy0 = blockIdx.y * kBlurTile, ox0 = blockIdx.x * kBlurTile;
  const size_t plane = (size_t)H * W;
  // pixel range (inclusive), window of the extended domain its pixels fold from (inclusive), inputs
  const int py0 = INTERIOR ? oy0 - R : max(oy0 - R, 0), px0 = INTERIOR ? ox0 - R : max(ox0 - R, 0);
  const int py1 = INTERIOR ? oy0 + kBlurTile - 1 + R : min(oy0 + kBlurTile - 1 + R, H - 1);
  const int px1 = INTERIOR ? ox0 + kBlurTile - 1 + R : min(ox0 + kBlurTile - 1 + R, W - 1);
  const int wy0 = INTERIOR ? py0 : ((py0 == 0) ? -R : py0);
  const int wx0 = INTERIOR ? px0 : ((px0 == 0) ? -R : px0);
  const int wy1 = INTERIOR ? py1 : ((py1 >= H - 1 - R && py0 <= H - 2) ? 2 * (H - 1) - max(py0, H - 1 - R) : py1);
  const int wx1 = INTERIOR ? px1 : ((px1 >= W - 1 - R && px0 <= W - 2) ? 2 * (W - 1) - max(px0, W - 1 - R) : px1);
  const int wh = INTERIOR ? TP : wy1 - wy0 + 1, ww = INTERIOR ? TP : wx1 - wx0 + 1;
  const int ih = INTERIOR ? TC : wh + 2 * R, iw = INTERIOR ? TC : ww + 2 * R;
  // slot -> pixel.  Slots 0..NO-1: the tile's own pixels, k_blur_bwd_fused's map; slots NO..: the ring P \ tile
  // (rows above, rows below, columns left, columns right of the tile), skipped where it leaves the image.
  int sy[NS], sx[NS];
  bool live[NS];
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int e = (int)threadIdx.x + j * kBlock;
    sy[j] = oy0 + e / kBlurTile;
    sx[j] = ox0 + (e & (kBlurTile - 1));
    live[j] = INTERIOR || (sy[j] < H && sx[j] < W);
  }
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int e = (int)threadIdx.x + j * kBlock;
    int y, x;
    if (e < 2 * R * TP) {                       // R rows above, R rows below: TP columns each
      const int row = e / TP;
      x = ox0 - R + (e - row * TP);
      y = (row < R) ? oy0 - R + row : oy0 + kBlurTile + (row - R);
    } else {                                    // 32 rows: R columns left, R columns right
      const int f = e - 2 * R * TP, row = f / (2 * R), col = f - row * (2 * R);
      y = oy0 + row;
      x = (col < R) ? ox0 - R + col : ox0 + kBlurTile + (col - R);
    }
    sy[NO + j] = y;
    sx[NO + j] = x;
    live[NO + j] = (e < NRING) && (INTERIOR || (y >= 0 && y < H && x >= 0 && x < W));
  }
  // per-pixel state: fetched first, used last
  float pp[NS], mk[NS], mm[NS], vv[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    pp[j] = mk[j] = mm[j] = vv[j] = 0.0f;
    if (live[j]) {
      const size_t i = (size_t)c * plane + (size_t)sy[j] * W + sx[j];
      pp[j] = p[i];
      mk[j] = mask[i];
      if (o.kind == 0) {
        mm[j] = m[i];
        vv[j] = v[i];
      }
    }
  }
  // the forward epilogue's x0 (own pixels)
  float xv[NO];
#pragma unroll
  for (int j = 0; j < NO; ++j) xv[j] = live[j] ? x0[(size_t)c * plane + (size_t)sy[j] * W + sx[j]] : 0.0f;
  // input tile: (gradient w.r.t. s) + imgfit'(s) inside the image, zero outside
  const unsigned magic = INTERIOR ? 0u : (unsigned)((0x100000000ULL + (unsigned)iw - 1) / (unsigned)iw);   // e / iw, e < 2^16
  const int ne = ih * iw;
  float gv[NL], sv[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int e = (int)threadIdx.x + j * kBlock;
    gv[j] = 0.0f;
    sv[j] = 0.0f;
    if (e < ne) {
      const int iy = INTERIOR ? e / TC : (int)__umulhi((unsigned)e, magic), ix = e - iy * iw;
      const int gy = wy0 - R + iy, gx = wx0 - R + ix;
      if (INTERIOR || (gy >= 0 && gy < H && gx >= 0 && gx < W)) {
        const size_t oidx = (size_t)c * plane + (size_t)gy * W + gx;
        gv[j] = gsrc[oidx];
        sv[j] = s[oidx];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int e = (int)threadIdx.x + j * kBlock;
    if (e < ne) {
      const int iy = INTERIOR ? e / TC : (int)__umulhi((unsigned)e, magic), ix = e - iy * iw;
      const int gy = wy0 - R + iy, gx = wx0 - R + ix;
      float t = 0.0f;
      if (INTERIOR || (gy >= 0 && gy < H && gx >= 0 && gx < W)) t = gv[j] + imgfit_grad(sv[j], c_fit);
      tile[iy][ix] = t;
    }
  }
  __syncthreads();
  float w[K];
  blur_weights_fetch<R>(wn_b, w);
  {
    const unsigned mw = (!INTERIOR && ww > 1) ? (unsigned)((0x100000000ULL + (unsigned)ww - 1) / (unsigned)ww) : 0u;
    for (int e = threadIdx.x; e < ih * ww; e += kBlock) {
      const int iy = INTERIOR ? e / TP : ((ww > 1) ? (int)__umulhi((unsigned)e, mw) : e), x = e - iy * ww;
      float a = 0.0f;
#pragma unroll
      for (int t = 0; t < K; ++t) a += w[t] * tile[iy][x + t];
      tmp[iy][x] = a;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < wh * ww; e += kBlock) {
      const int y = INTERIOR ? e / TP : ((ww > 1) ? (int)__umulhi((unsigned)e, mw) : e), x = e - y * ww;
      float a = 0.0f;
#pragma unroll
      for (int t = 0; t < K; ++t) a += w[t] * tmp[y + t][x];
      c2t[y][x] = a;
    }
  }
  __syncthreads();
  // gradient, optimiser; the owner stores.  x_{t+1} goes to the forward tile (aliases `tile`: last read two barriers ago) at
  // every extended position that reflects onto the pixel
  float (*xt)[kBlurTile + 6 * R + 1] = tile;
  const int ey0 = oy0 - R, ex0 = ox0 - R;            // extended coordinate of xt[0][0]
  double nacc[1] = {0.0};
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    if (!live[j]) continue;
    const int y = sy[j], x = sx[j];
    float gx = 0.0f;
    // blur_fold: own position, the reflection about the first row / column, about the last - three candidates per axis with
    // a flag each (compile-time indices: an array filled by count went to scratch), visited in blur_fold's order
    const int yy[3] = {y, -y, 2 * (H - 1) - y}, xx[3] = {x, -x, 2 * (W - 1) - x};
    const bool yv[3] = {true, !INTERIOR && y >= 1 && y <= R, !INTERIOR && y <= H - 2 && y >= H - 1 - R};
    const bool xw[3] = {true, !INTERIOR && x >= 1 && x <= R, !INTERIOR && x <= W - 2 && x >= W - 1 - R};
    if (INTERIOR) {
      gx += c2t[y - wy0][x - wx0];
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (!yv[a]) continue;
#pragma unroll
        for (int b = 0; b < 3; ++b)
          if (xw[b]) gx += c2t[yy[a] - wy0][xx[b] - wx0];
      }
    }
    float pv = pp[j];
    const float t = tanhf(pv);
    float g = (gx * eps) * (1.0f - t * t);
    g = g * mk[j];                              // attack_model.py:336
    const bool own = j < NO;
    const size_t i = (size_t)c * plane + (size_t)y * W + x;
    if (own) {
      grad[i] = g;
      nacc[0] += (double)g * (double)g;
    }
    // o.apply is 1 (host): this launch always steps
    if (o.kind == 0) {
      float m1 = mm[j], v1 = vv[j];
      adamw_element(pv, m1, v1, g, o);
      if (own) { p_out[i] = pv; m_out[i] = m1; v_out[i] = v1; }
    } else {
      float sg = sign_direction(g);
      pv = pv - o.lr * sg;
      if (own) p_out[i] = pv;
    }
    const float xn = eps * tanhf(pv);           // k_blur_fwd_r<1>: scalar * tanhf(in)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int ty = yy[a] - ey0;
      if (!yv[a] || ty < 0 || ty >= TS) continue;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int tx = xx[b] - ex0;
        if (xw[b] && tx >= 0 && tx < TS) xt[ty][tx] = xn;
      }
    }
  }
  {
    const int blk = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_sum_store<1>(nacc, norm_partials + blk);       // ends with a barrier-separated LDS reduction of its own
    if (blk == 0 && threadIdx.x == 0) norm_partials[kNormCountSlot] = (double)(gridDim.x * gridDim.y * 3);
  }
  __syncthreads();
  // forward blur of x_{t+1} (k_blur_fwd_r from its first barrier on): rows, columns, s = x0 + blur, statistics partials
  constexpr int NR = (TS * kBlurTile + kBlock - 1) / kBlock;
  float wf[K];
  blur_weights_fetch<R>(wn_f, wf);
  float (*tmpf)[kBlurTile + 4 * R + 1] = tmp;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int e = (int)threadIdx.x + j * kBlock;
    if (e < TS * kBlurTile) {
      const int ty = e / kBlurTile, x = e & (kBlurTile - 1);
      float a = 0.0f;
#pragma unroll
      for (int t = 0; t < K; ++t) a += wf[t] * xt[ty][x + t];
      tmpf[ty][x] = a;
    }
  }
  __syncthreads();
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int e = (int)threadIdx.x + j * kBlock;
    const int y = e / kBlurTile, x = e & (kBlurTile - 1);
    if (live[j]) {
      float a = 0.0f;
#pragma unroll
      for (int t = 0; t < K; ++t) a += wf[t] * tmpf[y + t][x];
      const float sn = xv[j] + a;
      s_next[(size_t)c * plane + (size_t)sy[j] * W + sx[j]] = sn;
      stat_accumulate(sn, a, acc);
    }
  }
  const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  block_sum_store<kStatSlots>(acc, img_partials + blk * kStatSlots);
}

template <int R>
__global__ void __launch_bounds__(kBlock) k_blur_step(const float* __restrict__ gsrc, const float* __restrict__ s, int H, int W,
                                                      float sigma_b, float eps, float c_fit, const float* __restrict__ p,
                                                      const float* __restrict__ m, const float* __restrict__ v,
                                                      float* __restrict__ p_out, float* __restrict__ m_out,
                                                      float* __restrict__ v_out, float* __restrict__ grad,
                                                      const float* __restrict__ mask, OptScalars o,
                                                      double* __restrict__ norm_partials, const float* __restrict__ x0,
                                                      float sigma_f, float* __restrict__ s_next,
                                                      double* __restrict__ img_partials, TapBuild taps0, TapBuild taps1,
                                                      int tap_blocks) {
  if (blockIdx.z == 3) {
    const int tb = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    if (tb < 2 * tap_blocks) {
      const int axis = tb / tap_blocks;
      if (axis == 0) build_taps_row_c(taps0, tb * (int)blockDim.x + (int)threadIdx.x);
      else build_taps_row_c(taps1, (tb - tap_blocks) * (int)blockDim.x + (int)threadIdx.x);
    }
    return;
  }
  constexpr int K = 2 * R + 1, TI = kBlurTile + 6 * R, TO = kBlurTile + 4 * R;
  __shared__ float tile[TI][TI + 1];
  __shared__ float tmp[TI][TO + 1];
  __shared__ float c2t[TO][TO + 1];
  __shared__ float wn_b[K];
  __shared__ float wn_f[K];
  blur_weights_publish<R>(sigma_b, wn_b);
  blur_weights_publish<R>(sigma_f, wn_f);
  const int oy0 = blockIdx.y * kBlurTile, ox0 = blockIdx.x * kBlurTile;
  // interior: the pixel range tile +- R and its 2R input halo strictly inside the image, none of its pixels folds
  const bool interior = oy0 >= kBlurTile && ox0 >= kBlurTile && oy0 + kBlurTile - 1 + 3 * R < H - 1 &&
                        ox0 + kBlurTile - 1 + 3 * R < W - 1;
  if (interior)
    blur_step_body<R, true>(gsrc, s, H, W, eps, c_fit, p, m, v, p_out, m_out, v_out, grad, mask, o, norm_partials, x0, s_next, img_partials, wn_b, wn_f,
                            tile, tmp, c2t);
  else
    blur_step_body<R, false>(gsrc, s, H, W, eps, c_fit, p, m, v, p_out, m_out, v_out, grad, mask, o, norm_partials, x0, s_next, img_partials, wn_b,
                             wn_f, tile, tmp, c2t);
}

}  // namespace advx
