// advx_kernels.h - the HIP kernels of the pixel-space hot path (gfx950, wave64).
//
// All kernels are HBM/L2-bound elementwise, stencil, gather or reduction work: no MFMA.
// The two kernels that move B*P_out floats (k_emit, k_batch_reduce / the fused pair) are
// the ones the roofline refers to; everything else touches ~1-15 MB and is bounded by
// launch latency.  Compiled with -ffp-contract=off so that a*b+c rounds like torch's
// separate mul / add unless fmaf is written explicitly.
#pragma once
#include "advx_device.h"
#include "advx_comm.h"

namespace advx {

// =========================================================================== image fwd
// per-element statistics of s = x0 + x accumulated in double (attack_model.py:86-106,
// 366-373, 386-391): [0] sum d, [1] sum d^2 with d = |q - s|, q = trunc(clamp(s)*255)/255;
// [2] sum relu(-s)^2 + relu(s-0.9)^2 ; [3] sum x ; [4] sum x^2
__device__ inline void stat_accumulate(float s, float x, double (&acc)[kStatSlots]) {
  float cl = fminf(fmaxf(s, 0.0f), 1.0f);
  float q = (float)(uint32_t)(cl * 255.0f) / 255.0f;  // C truncation, like astype(uint8)
  float d = fabsf(q - s);
  float lo = fmaxf(0.9f * 0.0f - s, 0.0f);
  float up = fmaxf(s - 0.9f * 1.0f, 0.0f);
  acc[0] += (double)d;
  acc[1] += (double)d * (double)d;
  acc[2] += (double)(lo * lo + up * up);
  acc[3] += (double)x;
  acc[4] += (double)x * (double)x;
}

// x = eps*tanh(p); WRITE_S: s = x0 + x and statistics partials, else write x (blur follows)
// the grid-stride loop of k_prep / k_prep_taps with the loads of FOUR of a thread's elements in flight (a 512 x 512 image on
// the 1024 workgroups these launches get is three elements per thread: three memory round trips in a row with the plain
// loop); the elements are processed in the loop's order, so the statistics partials keep their bits
template <bool WRITE_S>
__device__ inline void prep_elements(const float* __restrict__ p, const float* __restrict__ x0, float eps, long long n,
                                     float* __restrict__ out, double (&acc)[kStatSlots], long long first, long long stride) {
  constexpr int U = 4;
  for (long long i0 = first; i0 < n; i0 += U * stride) {
    float pv[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      pv[u] = (i < n) ? p[i] : 0.0f;
      xv[u] = (WRITE_S && i < n) ? x0[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      if (i < n) {
        const float x = eps * tanhf(pv[u]);
        if (WRITE_S) {
          const float s = xv[u] + x;
          out[i] = s;
          stat_accumulate(s, x, acc);
        } else {
          out[i] = x;
        }
      }
    }
  }
}
template <bool WRITE_S>
__global__ void __launch_bounds__(kBlock) k_prep(const float* __restrict__ p, const float* __restrict__ x0,
                                                 float eps, long long n, float* __restrict__ out,
                                                 double* __restrict__ partials) {
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  prep_elements<WRITE_S>(p, x0, eps, n, out, acc, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
  if (WRITE_S) block_sum_store<kStatSlots>(acc, partials + (size_t)blockIdx.x * kStatSlots);
}

#ifndef ADVX_FINALIZE_U
#define ADVX_FINALIZE_U 8      // rows in flight per thread in the ||g|| reductions that ride in block 0 of the image-sized launches (1: rounds 1-3)
#endif
#ifndef ADVX_FINALIZE_IMG_U
#define ADVX_FINALIZE_IMG_U 4  // ... and in the statistics reductions there (six doubles per row)
#endif
constexpr int kFinU = ADVX_FINALIZE_U;
constexpr int kFinImgU = ADVX_FINALIZE_IMG_U;

// one whole block: reduce the per-block partials, [rotate SIGMA <- QERR_STD], write stats
// The plain loop (U = 1) waits for memory once per row - and the rows were written by the launch before, on other XCDs: a
// thread of the ONE block that reduces 768 rows of a 512 x 512 blur (3072 of the prepared chain, twice that at 128 threads)
// waits 6 (12, 24) full memory latencies in a row while the launch's other workgroups have long finished: this block was a
// large part of the "6-12 us whatever they move" of the image-sized launches (round 4: k_stage_fwd_t 10.9 -> 7.2 us,
// Qwen2-VL's k_stage0_fwd_multi 8.7 -> 7.8; 6.7 without any reduction).  U > 1: U rows in flight per thread in registers
// (from a clamped row where the thread has none), added in the loop's order: the same bits.  12 U VGPRs for the statistics'
// six doubles per row: U = 4 keeps every launch at <= 64 VGPRs (8 waves per SIMD); at U = 8 the heavy launches lost more
// occupancy than the reduction gave back (k_plan_tail 18.0 -> 22.2 us, Phi-3.5's k_stage0_fwd_multi 8.9 -> 9.9).
template <bool ROTATE, int U = 1>
__device__ inline void finalize_image_block(const double* __restrict__ partials, int nblk, long long n,
                                            float* __restrict__ stats) {
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  if (U > 1) {
    const int step = (int)blockDim.x;
    for (int b0 = threadIdx.x; b0 < nblk; b0 += U * step) {
      double r[U][kStatSlots];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int b = min(b0 + u * step, nblk - 1);
#pragma unroll
        for (int k = 0; k < kStatSlots; ++k) r[u][k] = partials[(size_t)b * kStatSlots + k];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (b0 + u * step < nblk) {
#pragma unroll
          for (int k = 0; k < kStatSlots; ++k) acc[k] += r[u][k];
        }
      }
    }
  } else {
    for (int b = threadIdx.x; b < nblk; b += blockDim.x)
      for (int k = 0; k < kStatSlots; ++k) acc[k] += partials[(size_t)b * kStatSlots + k];
  }
  __shared__ double tot[kStatSlots];
  block_sum_store<kStatSlots>(acc, tot);
  if (threadIdx.x == 0) {
    double N = (double)n;
    double mean_d = tot[0] / N;
    double var_d = (n > 1) ? (tot[1] - tot[0] * tot[0] / N) / (N - 1.0) : 0.0;
    double mean_x = tot[3] / N;
    double var_x = (n > 1) ? (tot[4] - tot[3] * tot[3] / N) / (N - 1.0) : 0.0;
    if (ROTATE) stats[0] = stats[1];  // ADVX_STAT_SIGMA <- previous step's QERR_STD
    stats[1] = (float)sqrt(var_d > 0.0 ? var_d : 0.0);
    stats[2] = (float)mean_d;
    stats[3] = (float)tot[0];
    stats[4] = (float)(tot[2] / N);
    stats[5] = (float)mean_x;
    stats[6] = (float)sqrt(var_x > 0.0 ? var_x : 0.0);
  }
}
__global__ void __launch_bounds__(kBlock) k_finalize_image(const double* __restrict__ partials, int nblk,
                                                           long long n, float* __restrict__ stats) {
  finalize_image_block<true, kFinImgU>(partials, nblk, n, stats);
}

// one whole block: ||g||_2 from per-block sums of squares -> stats[GRAD_NORM]
template <int U = 1>   // as finalize_image_block's (one double per row: U = 8 is 16 VGPRs)
__device__ inline void finalize_norm_block(const double* __restrict__ partials, int nblk, float* __restrict__ stats) {
  double acc[1] = {0.0};
  if (U == 1) {
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) acc[0] += partials[b];
  } else {
    const int step = (int)blockDim.x;
    for (int b0 = threadIdx.x; b0 < nblk; b0 += U * step) {
      double r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = partials[min(b0 + u * step, nblk - 1)];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (b0 + u * step < nblk) acc[0] += r[u];
    }
  }
  __shared__ double tot1[1];
  block_sum_store<1>(acc, tot1);
  if (threadIdx.x == 0) stats[7] = (float)sqrt(tot1[0]);
}

// Header of the fused pair's scratch: which partial sets still wait for their reduction.
// The reductions ride in block (0,0) of the next k_fused_fwd (kernel boundaries give the
// ordering and visibility), so a fused step is two launches instead of four.
struct FusedHeader {
  int norm_blocks;   // > 0: k_fused_bwd<true> left that many sum-of-squares partials
  int image_blocks;  // > 0: k_fused_fwd left that many statistics partial rows
  int pad[14];
};

// explicit flush (host reads stats / a non-fused kernel follows)
__global__ void __launch_bounds__(kBlock) k_fused_flush(FusedHeader* __restrict__ hdr, const double* __restrict__ img_partials,
                                                        const double* __restrict__ norm_partials, long long n,
                                                        int image_too, float* __restrict__ stats) {
  int ib = image_too ? hdr->image_blocks : 0, nb = hdr->norm_blocks;
  __syncthreads();
  if (ib > 0) finalize_image_block<true, kFinImgU>(img_partials, ib, n, stats);
  if (nb > 0) finalize_norm_block<kFinU>(norm_partials, nb, stats);
  if (threadIdx.x == 0) {
    if (image_too) hdr->image_blocks = 0;
    hdr->norm_blocks = 0;
  }
}

// ===================================================================== tap-table builder
// Device-side construction of the tables of a dynamic resize (the random-resized-crop
// window changes every step).  blockIdx.y = axis; rows [0,out) forward, [out,out+in) transposed.
struct TapBuild {
  int mode, in_size, out_size, stride, tstride;
  int row_lo, row_hi;   // the rows of [0, out_size + in_size) this launch builds (the transposed rows may ride in a later one)
  int* start;
  int* count;
  float* w;
  int* tstart;
  int* tcount;
  float* tw;
  // compose != 0: the table of TWO resizes in a row - A: the crop window (in_size rows starting at image row `offset`)
  // antialiased-bilinear to mid_size rows (the image's size, attack_model.py:309-310), then B: the plan's stage-0 resize
  // mid_size -> out_size (mode_b).  Rows [0, out_size) forward (taps into IMAGE rows), rows [out_size, out_size + mid_size)
  // transposed, one per IMAGE row (no taps outside the window).  See build_composed_row.
  int compose, mid_size, mode_b, offset;
  const int* b_start;    // B's forward rows as the plan keeps them on the device (per out row: start, count, b_stride weights)
  const int* b_count;
  const float* b_w;
  int b_stride;
};
__device__ inline void build_taps_row(const TapBuild& a, int i);
__device__ inline void build_taps_row_c(const TapBuild& a, int i);
__device__ inline void build_composed_row(const TapBuild& a, int i);
__global__ void __launch_bounds__(kBlock) k_build_taps(TapBuild a0, TapBuild a1) {
  // a branch per axis, not a select between the two argument structs: selecting makes the compiler copy both to scratch
  if (blockIdx.y == 0) build_taps_row_c(a0, blockIdx.x * blockDim.x + threadIdx.x);
  else build_taps_row_c(a1, blockIdx.x * blockDim.x + threadIdx.x);
}
// build_taps_row_c: either kind of table.  Only the launches that may carry COMPOSED rows call it (k_build_taps, the image
// kernels, k_stage0_fwd_multi): inlined into k_emit's rider it took that kernel from 30 to 57 VGPRs and from 76 to 106 SGPRs.
__device__ inline void build_taps_row_c(const TapBuild& a, int i) {
  if (i < a.row_lo || i >= a.row_hi) return;
  if (a.compose) build_composed_row(a, i);
  else build_taps_row(a, i);
}
__device__ inline void build_taps_row(const TapBuild& a, int i) {
  if (i < a.row_lo || i >= a.row_hi) return;
  if (i < a.out_size) {
    float* w = a.w + (size_t)i * a.stride;
    TapRow r = tap_row(a.mode, a.in_size, a.out_size, i, a.stride, w);
    a.start[i] = r.start;
    a.count[i] = r.count;
  } else if (i < a.out_size + a.in_size) {
    int j = i - a.out_size;
    TapRow t = tap_bounds_transposed(a.mode, a.in_size, a.out_size, j);
    if (t.count > a.tstride) t.count = a.tstride;  // guarded by the host-side bound
    a.tstart[j] = t.start;
    a.tcount[j] = t.count;
    float* tw = a.tw + (size_t)j * a.tstride;
    int st = a.stride < 64 ? a.stride : 64;
    // no row buffer indexed at run time: a kernel that carries these rows needs no scratch memory
    for (int q = 0; q < a.tstride; ++q) tw[q] = (q < t.count) ? tap_weight(a.mode, a.in_size, a.out_size, t.start + q, j, st) : 0.0f;
  }
}
// Composed table C = B o A of a crop window's resize (A) followed by the plan's stage-0 resize (B): one gather from the
// image to the canvas instead of two launches with an image-sized intermediate (the reference evaluates them one after the
// other, attack_model.py:307-314; C drops the float32 rounding of that intermediate, so this path is held to the oracle at
// 1e-4, not to bit-identity with the two-launch kernels).  w_C[y][j] = sum_k w_B[y][k] * w_A[k][j], k ascending.
//   forward rows   : B's row comes from the plan's own device table (b_*), A's rows are formed here - the window is never
//                    larger than the image, so A up-samples: support 1, at most three taps, kept in scalars - and the
//                    products accumulate in sixteen registers picked by unrolled selects (no row buffer indexed at run
//                    time: a kernel that carries these rows needs no scratch memory; no read-modify-write chain through L2);
//   transposed rows: built by a LATER launch from the finished forward table - the canvas rows that read image row r are a
//                    contiguous range found by bisection on start / start + count (both non-decreasing), and the weights are
//                    the forward table's own floats: the backward is the exact adjoint of the forward.
// one row of the antialiased-bilinear table for scale <= 1 (tap_row's arithmetic, three taps at most, in scalars)
__device__ inline void aa_row3(int in_size, int out_size, int i, int& start, int& count, float& w0, float& w1, float& w2) {
  const float scale = tap_scale(in_size, out_size);
  const float support = (scale >= 1.0f) ? (float)(1.0 * (double)scale) : 1.0f;
  const float invscale = (scale >= 1.0f) ? (float)(1.0 / (double)scale) : 1.0f;
  const float center = (float)((double)scale * ((double)i + 0.5));
  long xmin = (long)((double)(center - support) + 0.5);
  if (xmin < 0) xmin = 0;
  long xmax = (long)((double)(center + support) + 0.5);
  if (xmax > in_size) xmax = in_size;
  long xsize = xmax - xmin;
  if (xsize < 0) xsize = 0;
  if (xsize > 3) xsize = 3;
  start = (int)xmin;
  count = (int)xsize;
  float w[3] = {0.0f, 0.0f, 0.0f};
  float total = 0.0f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < count) {
      const float d = (float)(j + start) - center;
      const float arg = (float)(((double)d + 0.5) * (double)invscale);
      const float a = fabsf(arg);
      w[j] = (a < 1.0f) ? (1.0f - a) : 0.0f;
      total += w[j];
    }
  }
  if (total != 0.0f) {
    const float norm = (float)(1.0 / (double)total);
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j < count) w[j] *= norm;
  }
  w0 = w[0]; w1 = w[1]; w2 = w[2];
}
constexpr int kComposedRow = 16;      // kMaxComposedStride of the host
__device__ inline void build_composed_row(const TapBuild& a, int i) {
  if (i < a.out_size) {
    float acc[kComposedRow];
#pragma unroll
    for (int q = 0; q < kComposedRow; ++q) acc[q] = 0.0f;
    const int bs = a.b_start[i], bc = a.b_count[i];
    const float* bw = a.b_w + (size_t)i * a.b_stride;
    int lo = 0, hi = 0;
    for (int kk = 0; kk < bc; ++kk) {
      int as, ac;
      float w[3];
      aa_row3(a.in_size, a.mid_size, bs + kk, as, ac, w[0], w[1], w[2]);
      if (kk == 0) { lo = as; hi = as; }                    // starts are non-decreasing in k
      const float wb = bw[kk];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int slot = as + jj - lo;
        const float v = wb * w[jj];
#pragma unroll
        for (int q = 0; q < kComposedRow; ++q) acc[q] = (jj < ac && q == slot) ? acc[q] + v : acc[q];
      }
      if (as + ac > hi) hi = as + ac;
    }
    float* wr = a.w + (size_t)i * a.stride;
#pragma unroll
    for (int q = 0; q < kComposedRow; ++q)
      if (q < a.stride) wr[q] = acc[q];
    a.start[i] = a.offset + lo;
    a.count[i] = (hi - lo < a.stride) ? (hi - lo) : a.stride;
  } else if (i < a.out_size + a.mid_size) {
    const int r = i - a.out_size;          // image row (the forward table's starts are image rows too)
    float* tw = a.tw + (size_t)r * a.tstride;
    int ylo = 0, n = 0;
    if (r >= a.offset && r < a.offset + a.in_size) {
      int lo = 0, hi = a.out_size;         // first y with start + count > r
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a.start[mid] + a.count[mid] > r) hi = mid; else lo = mid + 1;
      }
      ylo = lo;
      lo = 0;
      hi = a.out_size;                     // first y with start > r
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a.start[mid] > r) hi = mid; else lo = mid + 1;
      }
      n = lo - ylo;
      if (n < 0) n = 0;
      if (n > a.tstride) n = a.tstride;    // guarded by the host-side bound
    }
    if (n == 0) ylo = 0;
    a.tstart[r] = ylo;
    a.tcount[r] = n;
    for (int q = 0; q < a.tstride; ++q) {
      float v = 0.0f;
      if (q < n) {
        const int y = ylo + q;
        const int slot = r - a.start[y];
        if (slot >= 0 && slot < a.count[y]) v = a.w[(size_t)y * a.stride + slot];
      }
      tw[q] = v;
    }
  }
}
// the rows of a TapRider are built by the first workgroups of the launch that carries it (row r of axis k by thread
// r - row_lo of its axis' blocks)
struct TapRider {
  TapBuild t[2];
  int blocks;          // per axis; 0: nothing rides
};
__device__ inline void ride_taps(const TapRider& tr, unsigned block) {
  if (block < 2u * (unsigned)tr.blocks) {
    const int axis = block >= (unsigned)tr.blocks;
    const TapBuild& a = axis ? tr.t[1] : tr.t[0];
    build_taps_row(a, a.row_lo + (int)(block - (unsigned)axis * tr.blocks) * (int)blockDim.x + (int)threadIdx.x);
  }
}

// ================================================================================ blur
// Separable Gaussian, LDS-staged 32x32 output tile with halo r on every side; one block =
// one tile of one channel.  REFLECT: torchvision GaussianBlur forward (reflect pad).
// ZERO + extended domain: the full correlation over [-r, n-1+r]^2 that the backward folds.
constexpr int kBlurTile = 32;
constexpr int kBlurMaxR = 15;

__device__ inline int reflect_index(int u, int n) {
  if (u < 0) u = -u;
  if (u > n - 1) u = 2 * (n - 1) - u;
  if (u < 0) u = 0;  // only for positions no valid output reads
  if (u > n - 1) u = n - 1;
  return u;
}

// MODE 0: reflect, output HxW, epilogue s = x0 + blur(x) with statistics (x0 may be null: plain store)
// MODE 1: zero padding, output (H+2r)x(W+2r)
// IN: what a tile element is made of while it is loaded - 0: in[i]; 1: eps*tanh(in[i]) (in = p: the
// tanh reparameterisation without a launch and a buffer of its own); 2: in[i] + imgfit'(aux[i])
// (in = gradient w.r.t. s, aux = s: the input of the blur adjoint).  Same expressions as k_prep /
// k_add_imgfit, so the same bits; halo elements are evaluated once per tile that needs them.
// IN == 1 launches may carry extra blocks (blockIdx.z == 3) that build the crop window's tap tables.
__device__ inline float imgfit_grad(float s, float c);
template <int MODE, int IN = 0>
__global__ void __launch_bounds__(kBlock) k_blur(const float* __restrict__ in, int H, int W, int r, float sigma,
                                                 const float* __restrict__ x0, float* __restrict__ out,
                                                 double* __restrict__ partials, const float* __restrict__ aux = nullptr,
                                                 float scalar = 0.0f, TapBuild taps0 = TapBuild(),
                                                 TapBuild taps1 = TapBuild(), int tap_blocks = 0) {
  if (IN == 1 && blockIdx.z == 3) {
    // tap-table builder blocks riding in this launch (k_build_taps' work)
    const int tb = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    if (tb < 2 * tap_blocks) {
      const int axis = tb / tap_blocks;
      if (axis == 0) build_taps_row_c(taps0, tb * (int)blockDim.x + (int)threadIdx.x);
      else build_taps_row_c(taps1, (tb - tap_blocks) * (int)blockDim.x + (int)threadIdx.x);
    }
    return;
  }
  __shared__ float wgt[2 * kBlurMaxR + 1];
  __shared__ float tile[kBlurTile + 2 * kBlurMaxR][kBlurTile + 2 * kBlurMaxR + 1];
  __shared__ float tmp[kBlurTile + 2 * kBlurMaxR][kBlurTile + 1];
  const int k = 2 * r + 1;
  const int ext = (MODE == 1) ? r : 0;
  const int OH = H + 2 * ext, OW = W + 2 * ext;
  const int c = blockIdx.z;
  const int oy0 = blockIdx.y * kBlurTile, ox0 = blockIdx.x * kBlurTile;
  // weights: exp(-0.5 (t/sigma)^2) / sum  (torchvision _get_gaussian_kernel1d)
  if ((int)threadIdx.x < k) {
    float t = (float)((int)threadIdx.x - r);
    float q = t / sigma;
    wgt[threadIdx.x] = expf(-0.5f * (q * q));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sum = 0.0f;
    for (int i = 0; i < k; ++i) sum += wgt[i];
    for (int i = 0; i < k; ++i) wgt[i] = wgt[i] / sum;
  }
  const float* src = in + (size_t)c * H * W;
  const int th = kBlurTile + 2 * r, tw = kBlurTile + 2 * r;
  for (int e = threadIdx.x; e < th * tw; e += blockDim.x) {
    int ty = e / tw, tx = e - ty * tw;
    int gy = oy0 - ext - r + ty, gx = ox0 - ext - r + tx;  // image coordinates
    float v = 0.0f;
    const bool on = (MODE == 0) || (gy >= 0 && gy < H && gx >= 0 && gx < W);
    if (on) {
      const size_t o = (MODE == 0) ? (size_t)reflect_index(gy, H) * W + reflect_index(gx, W) : (size_t)gy * W + gx;
      v = src[o];
      if (IN == 1) v = scalar * tanhf(v);
      if (IN == 2) v = v + imgfit_grad(aux[(size_t)c * H * W + o], scalar);
    }
    tile[ty][tx] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < th * kBlurTile; e += blockDim.x) {
    int ty = e / kBlurTile, x = e - ty * kBlurTile;
    float a = 0.0f;
    for (int t = 0; t < k; ++t) a += wgt[t] * tile[ty][x + t];
    tmp[ty][x] = a;
  }
  __syncthreads();
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  for (int e = threadIdx.x; e < kBlurTile * kBlurTile; e += blockDim.x) {
    int y = e / kBlurTile, x = e - y * kBlurTile;
    int oy = oy0 + y, ox = ox0 + x;
    if (oy < OH && ox < OW) {
      float a = 0.0f;
      for (int t = 0; t < k; ++t) a += wgt[t] * tmp[y + t][x];
      size_t o = ((size_t)c * OH + oy) * OW + ox;
      if (MODE == 0 && x0 != nullptr) {
        float s = x0[o] + a;
        out[o] = s;
        stat_accumulate(s, a, acc);
      } else {
        out[o] = a;
      }
    }
  }
  if (MODE == 0 && partials != nullptr) {
    size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_sum_store<kStatSlots>(acc, partials + blk * kStatSlots);
  }
}

// adjoint of reflect-pad + correlation = fold of the zero-padded full correlation `c2`
// (extended domain (H+2r)x(W+2r)): g[i] = c[i] + [1<=i<=r] c[-i] + [n-1-r<=i<=n-2] c[2(n-1)-i]
__device__ inline float blur_fold(const float* __restrict__ c2, int H, int W, int r, int y, int x) {
  const int OW = W + 2 * r;
  // an index folds from the left side, the right side, or (tiny images) both
  int yy[3], xx[3], ny = 0, nx = 0;
  yy[ny++] = y;
  if (y >= 1 && y <= r) yy[ny++] = -y;
  if (y <= H - 2 && y >= H - 1 - r) yy[ny++] = 2 * (H - 1) - y;
  xx[nx++] = x;
  if (x >= 1 && x <= r) xx[nx++] = -x;
  if (x <= W - 2 && x >= W - 1 - r) xx[nx++] = 2 * (W - 1) - x;
  float a = 0.0f;
  for (int i = 0; i < ny; ++i)
    for (int j = 0; j < nx; ++j) a += c2[(size_t)(yy[i] + r) * OW + (xx[j] + r)];
  return a;
}

__global__ void __launch_bounds__(kBlock) k_blur_fold(const float* __restrict__ c2, int H, int W, int r,
                                                      float* __restrict__ gx) {
  long long n = 3LL * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    int ch = (int)(i / ((long long)H * W));
    int rem = (int)(i - (long long)ch * H * W);
    int y = rem / W, x = rem - y * W;
    gx[i] = blur_fold(c2 + (size_t)ch * (H + 2 * r) * (W + 2 * r), H, W, r, y, x);
  }
}

// ============================================================================ image bwd
// d(image_fit_loss)/ds * scale, scale = imgfit_scale / N  (autograd of attack_model.py:94-104)
__device__ inline float imgfit_grad(float s, float c) {
  float lo = fmaxf(0.9f * 0.0f - s, 0.0f);
  float up = fmaxf(s - 0.9f * 1.0f, 0.0f);
  return c * (2.0f * up) - c * (2.0f * lo);
}

// grad_p (+)= (g_x * eps) * (1 - tanh(p)^2) ; g_x = folded blur adjoint or gs + imgfit'
template <bool BLUR>
__global__ void __launch_bounds__(kBlock) k_tanh_bwd(const float* __restrict__ p, const float* __restrict__ s,
                                                     const float* __restrict__ gs, const float* __restrict__ c2,
                                                     int H, int W, int r, float eps, float c_fit, int accumulate,
                                                     float* __restrict__ grad_p) {
  long long n = 3LL * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float gx;
    if (BLUR) {
      int ch = (int)(i / ((long long)H * W));
      int rem = (int)(i - (long long)ch * H * W);
      int y = rem / W, x = rem - y * W;
      gx = blur_fold(c2 + (size_t)ch * (H + 2 * r) * (W + 2 * r), H, W, r, y, x);
    } else {
      gx = gs[i] + imgfit_grad(s[i], c_fit);
    }
    float t = tanhf(p[i]);
    float g = (gx * eps) * (1.0f - t * t);
    grad_p[i] = accumulate ? (grad_p[i] + g) : g;
  }
}

// the image as it comes back from the lossless PNG round trip of attack_model.py:368-371:
// q = float(uint8(clamp(s,0,1)*255))/255 with C truncation (tensor2pil / pil_to_tensor,
// llavaprocessor.py:151-161) - the arithmetic stat_accumulate applies, as an image
__global__ void __launch_bounds__(kBlock) k_quantise(const float* __restrict__ s, long long n, float* __restrict__ q) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float cl = fminf(fmaxf(s[i], 0.0f), 1.0f);
    q[i] = (float)(uint32_t)(cl * 255.0f) / 255.0f;
  }
}

// plain tanh ops (unit tests / plugin-level autograd)
__global__ void __launch_bounds__(kBlock) k_tanh_fwd(const float* __restrict__ p, float eps, long long n,
                                                     float* __restrict__ x) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    x[i] = eps * tanhf(p[i]);
}
__global__ void __launch_bounds__(kBlock) k_tanh_bwd_plain(const float* __restrict__ p, const float* __restrict__ gx,
                                                           float eps, long long n, float* __restrict__ gp) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float t = tanhf(p[i]);
    gp[i] = (gx[i] * eps) * (1.0f - t * t);
  }
}

// k_prep with the crop window's tap tables built beside it: blocks [0, nprep) are k_prep's, the
// rest k_build_taps' (axis = which half of them) - one launch instead of two at the head of a step
// that crops; both only feed the resize of the window that follows.
template <bool WRITE_S>
__global__ void __launch_bounds__(kBlock) k_prep_taps(const float* __restrict__ p, const float* __restrict__ x0, float eps,
                                                      long long n, float* __restrict__ out, double* __restrict__ partials,
                                                      int nprep, TapBuild a0, TapBuild a1, int tap_blocks_per_axis) {
  // the builders come FIRST in the grid: dealt last by the dispatcher their microseconds of serial work trail the launch
  const int ntap = 2 * tap_blocks_per_axis;
  if ((int)blockIdx.x < ntap) {
    const int tb = (int)blockIdx.x;
    const int axis = tb / tap_blocks_per_axis;
    if (axis == 0) build_taps_row_c(a0, tb * blockDim.x + threadIdx.x);
    else build_taps_row_c(a1, (tb - tap_blocks_per_axis) * blockDim.x + threadIdx.x);
    return;
  }
  const int bid = (int)blockIdx.x - ntap;
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  prep_elements<WRITE_S>(p, x0, eps, n, out, acc, (long long)bid * blockDim.x + threadIdx.x, (long long)nprep * blockDim.x);
  if (WRITE_S) block_sum_store<kStatSlots>(acc, partials + (size_t)bid * kStatSlots);
}

// ========================================================================== stage fwd
// canvas[c,y,x] = normalise(pad | sum_a wa sum_b wb src[...])  - one thread per canvas element
__device__ inline float stage_fwd_value(const DStage& st, const float* __restrict__ src, long long src_cstride,
                                        int src_rstride, int c, int y, int x) {
  int ry = y - st.off_y, rx = x - st.off_x;
  float v;
  if (ry >= 0 && ry < st.res_h && rx >= 0 && rx < st.res_w) {
    const float* sp = src + (size_t)c * src_cstride;
    int ys = st.th.start[ry], yc = st.th.count[ry];
    int xs = st.tw.start[rx], xc = st.tw.count[rx];
    const float* wy = st.th.w + (size_t)ry * st.th.stride;
    const float* wx = st.tw.w + (size_t)rx * st.tw.stride;
    v = 0.0f;
    if (!st.inner_axis_h) {
      for (int a = 0; a < yc; ++a) {
        const float* rowp = sp + (size_t)(ys + a) * src_rstride + xs;
        float h = 0.0f;
        for (int b = 0; b < xc; ++b) h += wx[b] * rowp[b];
        v += wy[a] * h;
      }
    } else {
      for (int b = 0; b < xc; ++b) {
        float h = 0.0f;
        for (int a = 0; a < yc; ++a) h += wy[a] * sp[(size_t)(ys + a) * src_rstride + xs + b];
        v += wx[b] * h;
      }
    }
  } else {
    v = st.pad_value;
  }
  if (st.normalise) v = (v - st.mean[c]) / st.stdv[c];
  return v;
}

__global__ void __launch_bounds__(kBlock) k_stage_fwd(DStage st, const float* __restrict__ src, long long src_cstride,
                                                      int src_rstride, float* __restrict__ canvas) {
  long long n = 3LL * st.can_h * st.can_w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    int c = (int)((unsigned)i / ((unsigned)st.can_h * (unsigned)st.can_w));
    int rem = (int)((unsigned)i - (unsigned)c * (unsigned)st.can_h * (unsigned)st.can_w);
    int y = rem / st.can_w, x = rem - y * st.can_w;
    canvas[i] = stage_fwd_value(st, src, src_cstride, src_rstride, c, y, x);
  }
}

// k_stage_fwd whose block 0 first reduces the statistics partials of the image (the crop's resize
// follows, in the same call, the kernels that left them): no one-block launch in between
__global__ void __launch_bounds__(kBlock) k_stage_fwd_img(DStage st, const float* __restrict__ src, long long src_cstride,
                                                          int src_rstride, float* __restrict__ canvas,
                                                          const double* __restrict__ img_partials, int nblk, long long n_img,
                                                          float* __restrict__ stats) {
  if (blockIdx.x == 0 && nblk > 0) finalize_image_block<true, kFinImgU>(img_partials, nblk, n_img, stats);
  long long n = 3LL * st.can_h * st.can_w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    int c = (int)((unsigned)i / ((unsigned)st.can_h * (unsigned)st.can_w));
    int rem = (int)((unsigned)i - (unsigned)c * (unsigned)st.can_h * (unsigned)st.can_w);
    int y = rem / st.can_w, x = rem - y * st.can_w;
    canvas[i] = stage_fwd_value(st, src, src_cstride, src_rstride, c, y, x);
  }
}

// ========================================================================== stage bwd
// Gradient of one stage's canvas as the transposed resizes read it: `copies` images in canvas order left by
// k_batch_reduce (DPlan::gcan_off; QWEN publishes every canvas element `temporal` times) plus the gradient a
// later stage propagated into this canvas (dgrad).
struct CanvasGrad {
  const float* g;        // [copies][3][can_h][can_w], null when the canvas is not emitted
  int copies;
  long long copy_stride; // 3 * can_h * can_w
  const float* dgrad;    // [3][can_h][can_w] or null
};
__device__ __host__ inline CanvasGrad canvas_grad_of(const DPlan& pl, int stage, int can_h, int can_w, const float* ws,
                                                     const float* dgrad) {
  CanvasGrad cg;
  cg.g = (pl.gcan_off[stage] >= 0) ? ws + pl.gcan_off[stage] : nullptr;
  cg.copies = (pl.gcan_off[stage] >= 0) ? pl.gcan_copies[stage] : 0;
  cg.copy_stride = 3LL * can_h * can_w;
  cg.dgrad = dgrad;
  return cg;
}

// gradient of canvas element at offset o = (c*can_h + y)*can_w + x: the copies in order, then dgrad
__device__ inline float canvas_grad_at(const CanvasGrad& cg, size_t o) {
  float g = 0.0f;
  for (int t = 0; t < cg.copies; ++t) g += cg.g[(size_t)t * cg.copy_stride + o];
  if (cg.dgrad != nullptr) g += cg.dgrad[o];
  return g;
}

// gradient of SOURCE element (c, ys, xs) of a stage: transposed-tap gather over the canvas
// gradient (no atomics), divided by std where the stage normalises
__device__ inline float stage_bwd_value(const DStage& st, const CanvasGrad& cg, int c, int ys, int xs) {
  const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
  const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
  const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
  const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
  float v = 0.0f;
  for (int a = 0; a < oyc; ++a) {
    const size_t row = ((size_t)c * st.can_h + (st.off_y + oy + a)) * st.can_w + st.off_x + ox;
    float h = 0.0f;
    for (int b = 0; b < oxc; ++b) h += wx[b] * canvas_grad_at(cg, row + b);
    v += wy[a] * h;
  }
  if (st.normalise) v = v / st.stdv[c];
  return v;
}

// stage_bwd_value with a T x T window (T >= the transposed tables' row length): every load of the gather is issued before
// the first use, taps beyond a row's count come from the last valid tap's address and never enter the sum; the in-range
// taps are accumulated by stage_bwd_value's operations in its order (bit-identical).  MODE fixes canvas_grad_at's shape:
// 1 = one copy, 2 = one copy + dgrad, 3 = two copies.
template <int T, int MODE>
__device__ inline float stage_bwd_value_w(const DStage& st, const CanvasGrad& cg, int c, int ys, int xs) {
  constexpr int COPIES = (MODE == 3) ? 2 : 1;
  constexpr bool DG = MODE == 2;
  const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
  const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
  const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
  const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
  const int ly = max(oyc - 1, 0), lx = max(oxc - 1, 0);
  float wyv[T], wxv[T], r[T][T][COPIES + 1];
#pragma unroll
  for (int a = 0; a < T; ++a) {
    wyv[a] = wy[min(a, ly)];
    wxv[a] = wx[min(a, lx)];
  }
#pragma unroll
  for (int a = 0; a < T; ++a) {
    const size_t row = ((size_t)c * st.can_h + (st.off_y + oy + min(a, ly))) * st.can_w + st.off_x + ox;
#pragma unroll
    for (int b = 0; b < T; ++b) {
      const size_t o = row + min(b, lx);
#pragma unroll
      for (int t = 0; t < COPIES; ++t) r[a][b][t] = cg.g[(size_t)t * cg.copy_stride + o];
      if (DG) r[a][b][COPIES] = cg.dgrad[o];
    }
  }
  float v = 0.0f;
#pragma unroll
  for (int a = 0; a < T; ++a) {
    float h = 0.0f;
#pragma unroll
    for (int b = 0; b < T; ++b) {
      float g = 0.0f;                                  // canvas_grad_at: the copies in order, then dgrad
#pragma unroll
      for (int t = 0; t < COPIES; ++t) g += r[a][b][t];
      if (DG) g += r[a][b][COPIES];
      h = (b < oxc) ? h + wxv[b] * g : h;
    }
    v = (a < oyc) ? v + wyv[a] * h : v;
  }
  if (st.normalise) v = v / st.stdv[c];
  return v;
}

// stage_bwd_value_w with the window size a UNIFORM run-time value T <= 4 (several plans in one launch, each with its own
// tables: one code path per plan position instead of one per size): loads of rows / columns beyond T are skipped by scalar
// branches, the rest is in flight together as in the compiled windows; same operations, same order.
template <int MODE>
__device__ inline float stage_bwd_value_wu(const DStage& st, const CanvasGrad& cg, int T, int c, int ys, int xs) {
  constexpr int COPIES = (MODE == 3) ? 2 : 1;
  constexpr bool DG = MODE == 2;
  constexpr int TM = 4;
  const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
  const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
  const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
  const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
  const int ly = max(oyc - 1, 0), lx = max(oxc - 1, 0);
  float wyv[TM], wxv[TM], r[TM][TM][COPIES + 1];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    wyv[a] = wxv[a] = 0.0f;
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int t = 0; t <= COPIES; ++t) r[a][b][t] = 0.0f;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    if (a < T) {
      wyv[a] = wy[min(a, ly)];
      wxv[a] = wx[min(a, lx)];
      const size_t row = ((size_t)c * st.can_h + (st.off_y + oy + min(a, ly))) * st.can_w + st.off_x + ox;
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        if (b < T) {
          const size_t o = row + min(b, lx);
#pragma unroll
          for (int t = 0; t < COPIES; ++t) r[a][b][t] = cg.g[(size_t)t * cg.copy_stride + o];
          if (DG) r[a][b][COPIES] = cg.dgrad[o];
        }
      }
    }
  }
  float v = 0.0f;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    float h = 0.0f;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      float g = 0.0f;
#pragma unroll
      for (int t = 0; t < COPIES; ++t) g += r[a][b][t];
      if (DG) g += r[a][b][COPIES];
      h = (b < oxc) ? h + wxv[b] * g : h;
    }
    v = (a < oyc) ? v + wyv[a] * h : v;
  }
  if (st.normalise) v = v / st.stdv[c];
  return v;
}

// one thread per SOURCE element.  Grid = (column chunks, source rows, channels): the row of a
// workgroup is uniform, so its taps and weights are scalar work, and no thread divides to find its pixel.
template <int T = 0, int MODE = 0>   // T > 0: the gather as a compiled window (stage_bwd_value_w)
__global__ void __launch_bounds__(kBlock) k_stage_bwd(DStage st, CanvasGrad cg, float* __restrict__ gsrc,
                                                      long long gsrc_cstride, int gsrc_rstride, int accumulate) {
  const int c = blockIdx.z, ys = blockIdx.y;
  const int xs = blockIdx.x * blockDim.x + threadIdx.x;
  if (xs < st.src_w) {
    const size_t o = (size_t)c * gsrc_cstride + (size_t)ys * gsrc_rstride + xs;
    const float before = accumulate ? gsrc[o] : 0.0f;     // in flight while the taps are gathered
    const float v = (T > 0) ? stage_bwd_value_w<(T > 0 ? T : 1), (MODE > 0 ? MODE : 1)>(st, cg, c, ys, xs) : stage_bwd_value(st, cg, c, ys, xs);
    gsrc[o] = accumulate ? (before + v) : v;
  }
}

// k_stage_bwd with the three channels of a position in one thread: taps and weights looked up once, three gathers in
// flight.  For sources large enough that a third of the threads still fills the device (Phi-3.5's 672 x 672 hd canvas:
// 12.5 -> see DESIGN.md 5); at 113 k positions the same form was slower (two waves per SIMD, k_stage0_bwd_multi below).
__global__ void __launch_bounds__(kBlock) k_stage_bwd3(DStage st, CanvasGrad cg, float* __restrict__ gsrc,
                                                       long long gsrc_cstride, int gsrc_rstride, int accumulate, ImgGrid ig) {
  BlockXYZ blk;
  if (!xcd_band_block(ig, blk)) return;
  const int ys = blk.y;
  const int xs = blk.x * blockDim.x + threadIdx.x;
  if (xs >= st.src_w) return;
  const size_t o = (size_t)ys * gsrc_rstride + xs;
  float before[3] = {0.0f, 0.0f, 0.0f};
  if (accumulate) {
#pragma unroll
    for (int c = 0; c < 3; ++c) before[c] = gsrc[(size_t)c * gsrc_cstride + o];     // in flight while the taps are gathered
  }
  const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
  const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
  const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
  const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
  const size_t plane = (size_t)st.can_h * st.can_w;
  float v[3] = {0.0f, 0.0f, 0.0f};
  for (int a = 0; a < oyc; ++a) {
    const size_t row = (size_t)(st.off_y + oy + a) * st.can_w + st.off_x + ox;
    const float wa = wy[a];
    float h[3] = {0.0f, 0.0f, 0.0f};
    for (int b = 0; b < oxc; ++b) {
      const float wb = wx[b];
#pragma unroll
      for (int c = 0; c < 3; ++c) h[c] += wb * canvas_grad_at(cg, (size_t)c * plane + row + b);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] += wa * h[c];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r = st.normalise ? v[c] / st.stdv[c] : v[c];
    gsrc[(size_t)c * gsrc_cstride + o] = accumulate ? (before[c] + r) : r;
  }
}

// k_stage_bwd3 with the taps of one canvas ROW loaded together (round 4, ADVX_TUNE_ROW_BATCH).  The run-time loops of
// k_stage_bwd3 wait for memory once per TAP: the inner loop issues one weight and three gradient loads, `s_waitcnt vmcnt(0)`,
// three multiply-adds - a window of 5 x 5 taps is 25 round trips behind one another in a launch that has a few waves per SIMD
// to hide them.  Here the column weights are loaded once per thread, the taps of one row of the window (TB >= the table's row
// length, a compile-time bound) together for the three channels, and only the rows are walked at run time (their count is
// uniform: the row of a workgroup is): oyc round trips, and no load of a row the window does not have (what the T x T
// windows of advx_resize.h pay from 5 x 5 on).  Taps beyond a column's count are read from the last valid tap's address and
// never enter the sum; the in-range taps are accumulated by the same operations in the same order: bit-identical.
// COPIES / DG fix canvas_grad_at's shape at compile time so that nothing sits between the loads.
// Measured (profiles/r04): Qwen2-VL 512 10.3 -> 6.3 us, Phi-3.5 10.2 -> 8.6, cross 9.5 -> 7.8, the composed crop window's
// 9-tap rows 11.8 -> 9.4.  The same form of the FORWARD gather (k_stage0_fwd_multi) was level at <= 4 taps and slower on
// the composed crop's 8-tap rows (12.8 -> 13.6): not kept.
template <int TB, int COPIES, bool DG>
__device__ inline void stage_bwd3_rows(const DStage& st, const CanvasGrad& cg, int ys, int xs, float (&v)[3]) {
  const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
  const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
  const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
  const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
  const size_t plane = (size_t)st.can_h * st.can_w;
  float wv[TB];
  int off[TB];
  const int last = max(oxc - 1, 0);
#pragma unroll
  for (int k = 0; k < TB; ++k) {
    off[k] = min(k, last);
    wv[k] = wx[off[k]];
  }
  v[0] = v[1] = v[2] = 0.0f;
  for (int a = 0; a < oyc; ++a) {
    const size_t row = (size_t)(st.off_y + oy + a) * st.can_w + st.off_x + ox;
    const float wa = wy[a];
    float r[3][TB][COPIES + 1];     // [COPIES]: dgrad (DG)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < TB; ++k) {
        const size_t o = (size_t)c * plane + row + off[k];
#pragma unroll
        for (int t = 0; t < COPIES; ++t) r[c][k][t] = cg.g[(size_t)t * cg.copy_stride + o];
        if (DG) r[c][k][COPIES] = cg.dgrad[o];
      }
    float h[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < TB; ++k) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float g = 0.0f;                                  // canvas_grad_at: the copies in order, then dgrad
#pragma unroll
        for (int t = 0; t < COPIES; ++t) g += r[c][k][t];
        if (DG) g += r[c][k][COPIES];
        h[c] = (k < oxc) ? h[c] + wv[k] * g : h[c];
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] += wa * h[c];
  }
}
template <int COPIES, bool DG>
__device__ inline bool stage_bwd3_rows_any(const DStage& st, const CanvasGrad& cg, int ys, int xs, float (&v)[3]) {
  const int T = st.ttw.stride;
  if (T <= 2) stage_bwd3_rows<2, COPIES, DG>(st, cg, ys, xs, v);
  else if (T <= 4) stage_bwd3_rows<4, COPIES, DG>(st, cg, ys, xs, v);
  else if (T <= 6) stage_bwd3_rows<6, COPIES, DG>(st, cg, ys, xs, v);
  else if (T <= 8) stage_bwd3_rows<8, COPIES, DG>(st, cg, ys, xs, v);
  else if (T <= 10) stage_bwd3_rows<10, COPIES, DG>(st, cg, ys, xs, v);
  else return false;
  return true;
}
// ... and with the whole T x T window in flight (tables of <= 4 taps per row): as stage_bwd_value_w, three channels
template <int T, int MODE>
__global__ void __launch_bounds__(kBlock) k_stage_bwd3_w(DStage st, CanvasGrad cg, float* __restrict__ gsrc, long long gsrc_cstride,
                                                         int gsrc_rstride, int accumulate, ImgGrid ig) {
  constexpr int COPIES = (MODE == 3) ? 2 : 1;
  constexpr bool DG = MODE == 2;
  BlockXYZ blk;
  if (!xcd_band_block(ig, blk)) return;
  const int ys = blk.y;
  const int xs = blk.x * blockDim.x + threadIdx.x;
  if (xs >= st.src_w) return;
  const size_t o = (size_t)ys * gsrc_rstride + xs;
  float before[3] = {0.0f, 0.0f, 0.0f};
  if (accumulate) {
#pragma unroll
    for (int c = 0; c < 3; ++c) before[c] = gsrc[(size_t)c * gsrc_cstride + o];
  }
  const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
  const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
  const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
  const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
  const size_t plane = (size_t)st.can_h * st.can_w;
  const int ly = max(oyc - 1, 0), lx = max(oxc - 1, 0);
  float wyv[T], wxv[T], r[3][T][T][COPIES + 1];
#pragma unroll
  for (int a = 0; a < T; ++a) {
    wyv[a] = wy[min(a, ly)];
    wxv[a] = wx[min(a, lx)];
  }
#pragma unroll
  for (int a = 0; a < T; ++a) {
    const size_t row = (size_t)(st.off_y + oy + min(a, ly)) * st.can_w + st.off_x + ox;
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const size_t q = (size_t)c * plane + row + min(b, lx);
#pragma unroll
        for (int t = 0; t < COPIES; ++t) r[c][a][b][t] = cg.g[(size_t)t * cg.copy_stride + q];
        if (DG) r[c][a][b][COPIES] = cg.dgrad[q];
      }
  }
  float v[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int a = 0; a < T; ++a) {
    float h[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float g = 0.0f;                                  // canvas_grad_at: the copies in order, then dgrad
#pragma unroll
        for (int t = 0; t < COPIES; ++t) g += r[c][a][b][t];
        if (DG) g += r[c][a][b][COPIES];
        h[c] = (b < oxc) ? h[c] + wxv[b] * g : h[c];
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (a < oyc) ? v[c] + wyv[a] * h[c] : v[c];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float rr = st.normalise ? v[c] / st.stdv[c] : v[c];
    gsrc[(size_t)c * gsrc_cstride + o] = accumulate ? (before[c] + rr) : rr;
  }
}

// MODE: 1 = one copy, 2 = one copy + dgrad, 3 = two copies (Qwen2-VL's temporal pair)
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_stage_bwd3_rb(DStage st, CanvasGrad cg, float* __restrict__ gsrc, long long gsrc_cstride,
                                                          int gsrc_rstride, int accumulate, ImgGrid ig) {
  BlockXYZ blk;
  if (!xcd_band_block(ig, blk)) return;
  const int ys = blk.y;
  const int xs = blk.x * blockDim.x + threadIdx.x;
  if (xs >= st.src_w) return;
  const size_t o = (size_t)ys * gsrc_rstride + xs;
  float before[3] = {0.0f, 0.0f, 0.0f};
  if (accumulate) {
#pragma unroll
    for (int c = 0; c < 3; ++c) before[c] = gsrc[(size_t)c * gsrc_cstride + o];
  }
  float v[3];
  if (MODE == 1) stage_bwd3_rows_any<1, false>(st, cg, ys, xs, v);
  else if (MODE == 2) stage_bwd3_rows_any<1, true>(st, cg, ys, xs, v);
  else stage_bwd3_rows_any<2, false>(st, cg, ys, xs, v);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r = st.normalise ? v[c] / st.stdv[c] : v[c];
    gsrc[(size_t)c * gsrc_cstride + o] = accumulate ? (before[c] + r) : r;
  }
}

// Several plans over ONE image (cross-model runs, crossattack_models.py:352-391).  Their
// stage-0 kernels are each too small to fill the device and a dependent launch costs ~5 us, so
// the stage-0 work of all plans goes into one launch:
//   forward : blockIdx.y = plan; canvas_k = resize_k(image), pad, normalise
//   backward: one thread per image element sums the plans' transposed resizes left to right -
//             the value n accumulating launches of k_stage_bwd leave, without their n-1
//             read-modify-write passes over the image gradient.
constexpr int kMaxMulti = 4;
struct MultiFwd {
  int n;
  DStage st[kMaxMulti];
  float* canvas[kMaxMulti];
};
struct MultiBwd {
  int n;
  DStage st[kMaxMulti];
  CanvasGrad cg[kMaxMulti];
  int win[kMaxMulti];     // per plan: 4 * T + MODE of the windowed gather (stage_bwd_value_wu), 0 = run-time loops
};

// canvas[c,y,x] for c = 0..2 of one position: taps and weights looked up once (stage_fwd_value per channel)
__device__ inline void stage_fwd_value3(const DStage& st, const float* __restrict__ src, long long src_cstride, int src_rstride,
                                        int y, int x, float (&out)[3]) {
  const int ry = y - st.off_y, rx = x - st.off_x;
  float v[3];
  if (ry >= 0 && ry < st.res_h && rx >= 0 && rx < st.res_w) {
    const int ys = st.th.start[ry], yc = st.th.count[ry];
    const int xs = st.tw.start[rx], xc = st.tw.count[rx];
    const float* wy = st.th.w + (size_t)ry * st.th.stride;
    const float* wx = st.tw.w + (size_t)rx * st.tw.stride;
    v[0] = v[1] = v[2] = 0.0f;
    if (!st.inner_axis_h) {
      for (int a = 0; a < yc; ++a) {
        const float* rowp = src + (size_t)(ys + a) * src_rstride + xs;
        const float wa = wy[a];
        float h[3] = {0.0f, 0.0f, 0.0f};
        for (int b = 0; b < xc; ++b) {
          const float wb = wx[b];
#pragma unroll
          for (int c = 0; c < 3; ++c) h[c] += wb * rowp[(size_t)c * src_cstride + b];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] += wa * h[c];
      }
    } else {
      for (int b = 0; b < xc; ++b) {
        const float wb = wx[b];
        float h[3] = {0.0f, 0.0f, 0.0f};
        for (int a = 0; a < yc; ++a) {
          const float wa = wy[a];
#pragma unroll
          for (int c = 0; c < 3; ++c) h[c] += wa * src[(size_t)c * src_cstride + (size_t)(ys + a) * src_rstride + xs + b];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] += wb * h[c];
      }
    }
  } else {
    v[0] = v[1] = v[2] = st.pad_value;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[c] = st.normalise ? (v[c] - st.mean[c]) / st.stdv[c] : v[c];
}

// stage_fwd_value3 with a T x T window (T >= the tables' row lengths): every load issued before the first use, taps beyond a
// row's count read from the last valid tap's address and never summed; the same operations in the same order (bit-identical).
template <int T>
__device__ inline void stage_fwd_value3_w(const DStage& st, const float* __restrict__ src, long long src_cstride, int src_rstride,
                                          int y, int x, float (&out)[3]) {
  const int ry = y - st.off_y, rx = x - st.off_x;
  float v[3];
  if (ry >= 0 && ry < st.res_h && rx >= 0 && rx < st.res_w) {
    const int ys = st.th.start[ry], yc = st.th.count[ry];
    const int xs = st.tw.start[rx], xc = st.tw.count[rx];
    const float* wy = st.th.w + (size_t)ry * st.th.stride;
    const float* wx = st.tw.w + (size_t)rx * st.tw.stride;
    const int ly = max(yc - 1, 0), lx = max(xc - 1, 0);
    float wyv[T], wxv[T], r[3][T][T];
#pragma unroll
    for (int a = 0; a < T; ++a) {
      wyv[a] = wy[min(a, ly)];
      wxv[a] = wx[min(a, lx)];
    }
#pragma unroll
    for (int a = 0; a < T; ++a) {
      const float* rowp = src + (size_t)(ys + min(a, ly)) * src_rstride + xs;
#pragma unroll
      for (int b = 0; b < T; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c][a][b] = rowp[(size_t)c * src_cstride + min(b, lx)];
    }
    v[0] = v[1] = v[2] = 0.0f;
    if (!st.inner_axis_h) {
#pragma unroll
      for (int a = 0; a < T; ++a) {
        float h[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int b = 0; b < T; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) h[c] = (b < xc) ? h[c] + wxv[b] * r[c][a][b] : h[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (a < yc) ? v[c] + wyv[a] * h[c] : v[c];
      }
    } else {
#pragma unroll
      for (int b = 0; b < T; ++b) {
        float h[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) h[c] = (a < yc) ? h[c] + wyv[a] * r[c][a][b] : h[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (b < xc) ? v[c] + wxv[b] * h[c] : v[c];
      }
    }
  } else {
    v[0] = v[1] = v[2] = st.pad_value;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[c] = st.normalise ? (v[c] - st.mean[c]) / st.stdv[c] : v[c];
}

// grid = (column chunks of the widest canvas, rows of the tallest, plans)
// MAXW: the largest compiled window of this instantiation - 4 for the plans' own tables (58 VGPRs), 6 for the composed crop
// window's rows (146 VGPRs: a launch of its own, so that every other use keeps its occupancy)
template <int MAXW>
__global__ void __launch_bounds__(kBlock) k_stage0_fwd_multi(MultiFwd mf, const float* __restrict__ src, long long src_cstride,
                                                             int src_rstride, const double* __restrict__ img_partials, int nblk,
                                                             long long n_img, float* __restrict__ stats,
                                                             const double* __restrict__ norm_rows, int norm_count,
                                                             TapBuild tr0, TapBuild tr1, int tr_blocks, ImgGrid ig, int windowed) {
  BlockXYZ blk;
  if (!xcd_band_block(ig, blk)) return;
  // nblk > 0: the image kernels of the same call left statistics partials; block (0,0,0) reduces them
  // here (k_emit, the consumer of sigma, is a later launch); norm_count > 0: the ||g|| partials of the prepared chain's tail
  if (blk.x == 0 && blk.y == 0 && blk.z == 0) {
    if (nblk > 0) finalize_image_block<true, kFinImgU>(img_partials, nblk, n_img, stats);
    if (norm_count > 0) finalize_norm_block<kFinU>(norm_rows, norm_count, stats);
  }
  // tr_blocks > 0 (composed crop): the z == 0 layer of the grid - dispatched first - builds the TRANSPOSED rows of the
  // composed tables from the forward rows the image kernel before this launch finished (read only by the backward);
  // the plans follow at z - 1.  A latency-bound launch with a few waves per SIMD: the riders cost it nothing measurable.
  int zplan = (int)blk.z;
  if (tr_blocks > 0) {
    // round-robin grid: the z == 0 layer; XCD-aware grid: 2 * tr_blocks blocks appended behind the plans' (a whole layer of
    // mostly idle blocks in front would leave the first XCDs without work)
    if (blk.z == (ig.banded ? ig.gz : 0u)) {
      const int tb = ig.banded ? (int)blk.x : (int)(blk.y * ig.gx + blk.x);
      if (tb < tr_blocks) build_taps_row_c(tr0, tr0.row_lo + tb * (int)blockDim.x + (int)threadIdx.x);
      else if (tb < 2 * tr_blocks) build_taps_row_c(tr1, tr1.row_lo + (tb - tr_blocks) * (int)blockDim.x + (int)threadIdx.x);
      return;
    }
    if (!ig.banded) zplan -= 1;
  }
  const int k = zplan;
  const DStage& st = mf.st[k];
  const int y = blk.y;
  const int x = blk.x * blockDim.x + threadIdx.x;
  if (y < st.can_h && x < st.can_w) {
    float v[3];
    // windowed: 0 = loops, 1 = by the tables' row lengths, > 1 = the rows' real length (host: compose_exact).  Uniform: one
    // path per workgroup
    const int tm = !windowed ? 99 : (windowed > 1 ? windowed : max(st.th.stride, st.tw.stride));
    if (tm <= 2) stage_fwd_value3_w<2>(st, src, src_cstride, src_rstride, y, x, v);
    else if (tm <= 3) stage_fwd_value3_w<3>(st, src, src_cstride, src_rstride, y, x, v);
    else if (tm <= 4) stage_fwd_value3_w<4>(st, src, src_cstride, src_rstride, y, x, v);
    else if (MAXW >= 6 && tm <= 5) stage_fwd_value3_w<(MAXW >= 6 ? 5 : 1)>(st, src, src_cstride, src_rstride, y, x, v);
    else if (MAXW >= 6 && tm <= 6) stage_fwd_value3_w<(MAXW >= 6 ? 6 : 1)>(st, src, src_cstride, src_rstride, y, x, v);
    else stage_fwd_value3(st, src, src_cstride, src_rstride, y, x, v);
    float* __restrict__ canvas = mf.canvas[k];
    const size_t plane = (size_t)st.can_h * st.can_w;
#pragma unroll
    for (int c = 0; c < 3; ++c) canvas[(size_t)c * plane + (size_t)y * st.can_w + x] = v[c];
  }
}

// backward: one thread per (c, y, x) - at 113 k positions a thread per position leaves the SIMDs two waves each
// and the three-channel form (stage_bwd_value3) is latency-bound: measured 47 us against 27
__global__ void __launch_bounds__(kBlock) k_stage0_bwd_multi(MultiBwd mb, float* __restrict__ gsrc, long long gsrc_cstride,
                                                             int gsrc_rstride, int accumulate) {
  const int c = blockIdx.z, ys = blockIdx.y;
  const int xs = blockIdx.x * blockDim.x + threadIdx.x;
  if (xs < mb.st[0].src_w) {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < kMaxMulti; ++k) {
      if (k < mb.n) {
        const int w = mb.win[k];     // uniform
        float t;
        if ((w & 3) == 1) t = stage_bwd_value_wu<1>(mb.st[k], mb.cg[k], w >> 2, c, ys, xs);
        else if ((w & 3) == 2) t = stage_bwd_value_wu<2>(mb.st[k], mb.cg[k], w >> 2, c, ys, xs);
        else if ((w & 3) == 3) t = stage_bwd_value_wu<3>(mb.st[k], mb.cg[k], w >> 2, c, ys, xs);
        else t = stage_bwd_value(mb.st[k], mb.cg[k], c, ys, xs);
        v = (k == 0) ? t : v + t;
      }
    }
    size_t o = (size_t)c * gsrc_cstride + (size_t)ys * gsrc_rstride + xs;
    gsrc[o] = accumulate ? (gsrc[o] + v) : v;
  }
}

// adjoint of the random-resized crop (attack_model.py:307-310): the gradient arrives in the
// canvas order of the resized window (H x W); one thread per pixel of the WHOLE image writes the
// transposed resize inside the window (ci, cj, src_h, src_w) and exact zeros outside it
__device__ inline float crop_bwd_value(const DStage& st, const float* __restrict__ gcan, int ci, int cj, int c, int y, int x) {
  const int ys = y - ci, xs = x - cj;
  float v = 0.0f;
  if (ys >= 0 && ys < st.src_h && xs >= 0 && xs < st.src_w) {
    int oy = st.tth.start[ys], oyc = st.tth.count[ys];
    int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
    const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
    const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
    const float* gp = gcan + (size_t)c * st.can_h * st.can_w;
    for (int a = 0; a < oyc; ++a) {
      float h = 0.0f;
      for (int b = 0; b < oxc; ++b) h += wx[b] * gp[(size_t)(oy + a) * st.can_w + ox + b];
      v += wy[a] * h;
    }
  }
  return v;
}
__global__ void __launch_bounds__(kBlock) k_crop_bwd(DStage st, const float* __restrict__ gcan, float* __restrict__ gimg,
                                                     int H, int W, int ci, int cj) {
  const unsigned plane = (unsigned)H * (unsigned)W;
  const long long n = 3LL * plane;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((unsigned)i / plane);
    const unsigned rem = (unsigned)i - (unsigned)c * plane;
    const int y = (int)(rem / (unsigned)W), x = (int)(rem - (unsigned)y * (unsigned)W);
    gimg[i] = crop_bwd_value(st, gcan, ci, cj, c, y, x);
  }
}

// Element type of the two B*P_out tensors at the VLM boundary: 0 = float32 (what the reference
// hands to / gets from the model), 1 = float16, 2 = bfloat16.  A half-precision model casts
// pixel_values on entry and autograd casts the gradient back, both exactly representable
// steps (round-to-nearest-even on the way in, widening on the way out), so emitting / reading
// the model's dtype directly gives bit-identical numerics with half the traffic.
// 3 = float32 read with non-temporal loads: a gradient tensor larger than the 256 MiB Infinity
// Cache is read once and only evicts what the next launches need (measured: Phi-3.5 512 -18 us,
// Qwen2-VL 512 -12 us per step; the 86.7 MB headline tensor is better off cached: +1.6 us).
template <int IO>
__device__ inline float4 io_load4(const void* __restrict__ base, size_t idx) {
  if (IO == 0) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
  if (IO == 3) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v r = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(base) + idx));
    return make_float4(r.x, r.y, r.z, r.w);
  }
  // 1 / 2: float16 / bfloat16, cached; 4 / 5: the same read with non-temporal loads
  uint2 raw;
  if (IO >= 4) {
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    u2v r = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(reinterpret_cast<const unsigned short*>(base) + idx));
    raw = make_uint2(r.x, r.y);
  } else {
    raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
  }
  if (IO == 1 || IO == 4) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 a = __builtin_bit_cast(h2, raw.x), b = __builtin_bit_cast(h2, raw.y);
    return make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
  }
  return make_float4(__builtin_bit_cast(float, raw.x << 16), __builtin_bit_cast(float, raw.x & 0xffff0000u),
                     __builtin_bit_cast(float, raw.y << 16), __builtin_bit_cast(float, raw.y & 0xffff0000u));
}

// streaming (non-temporal) store of four consecutive elements
template <int IO>
__device__ inline void io_store4(void* __restrict__ base, size_t idx, float4 v) {
  if (IO == 0) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v ov = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(ov, reinterpret_cast<f4v*>(reinterpret_cast<float*>(base) + idx));
    return;
  }
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  u2v packed;
  if (IO == 1) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 a = {(_Float16)v.x, (_Float16)v.y}, b = {(_Float16)v.z, (_Float16)v.w};   // v_cvt_f16_f32: RNE
    packed = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
  } else {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 a = {(__bf16)v.x, (__bf16)v.y}, b = {(__bf16)v.z, (__bf16)v.w};           // v_cvt_pk_bf16_f32: RNE
    packed = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
  }
  __builtin_nontemporal_store(packed, reinterpret_cast<u2v*>(reinterpret_cast<unsigned short*>(base) + idx));
}

// XCD-aware column blocks (ADVX_TUNE_XCD_MAP = 2).  Workgroups are dealt round-robin over the 8 XCDs by linear id; with gridDim.x
// a multiple of 8, physical block bx runs on XCD bx % 8 whatever blockIdx.y is.  Mode 2 hands XCD k the CONTIGUOUS range of
// column blocks [k*per, (k+1)*per): all batch slices of a column block still share one XCD (one L2 fetches the shared v /
// canvas), and every XCD's stores sweep all address residues - where "column block = bx" (mode 1) pins XCD k to the 4 KiB
// stripes = k mod 8 of every row, measured 0.9 us slower on k_fused_fwd than no mapping at all (profiles/r04).
__device__ inline unsigned xcd_column_block(unsigned bx, unsigned gx_padded, int xmap) {
  if (xmap != 2) return bx;
  const unsigned per = gx_padded >> 3;
  return (bx & 7u) * per + (bx >> 3);
}
// the same for a launch whose grid serves several block counts (k_batch_reduce_multi: one count per plan): the contiguous ranges
// are cut from THIS count; -> column block, or -1 for a workgroup with nothing to do (grid.x >= 8 * ceil(nblocks / 8))
__device__ inline int xcd_block_of(unsigned bx, int nblocks, int xmap) {
  if (xmap != 2) return ((int)bx < nblocks) ? (int)bx : -1;
  const unsigned per = ((unsigned)nblocks + 7u) >> 3, j = bx >> 3;
  if (j >= per) return -1;
  const unsigned cb = (bx & 7u) * per + j;
  return ((int)cb < nblocks) ? (int)cb : -1;
}

// ================================================================================ emit
// out[b, idx] = canvas value of flat index idx (+ sigma * N(0,1)); one thread = 4
// consecutive flat indices (16-byte coalesced stores), blockIdx.y = slice of the batch.
__device__ inline float emit_value(const DPlan& pl, const float* __restrict__ ws, long long idx) {
  for (int k = 0; k < pl.n_emit; ++k) {
    const DEmit& e = pl.e[k];
    if (idx >= e.out_begin && idx < e.out_begin + e.out_count) {
      int c, y, x;
      emit_inverse(e, idx, c, y, x);
      return ws[pl.canvas_off[e.stage] + ((size_t)c * e.can_h + y) * e.can_w + x];
    }
  }
  return 0.0f;  // zero tiles (llama32processor.py:344-346, phi3processor.py:232-235)
}

// v[0..3] = emit_value(i0 .. i0+3).  Common case: the four indices lie in one emit and in one run of consecutive canvas
// elements (a tile row whose width is a multiple of 4, the inside of a 14-pixel patch row) - ONE inverse layout map
// (its divisions are a tenth of the generator's work per thread otherwise) and, when aligned, one 16-byte load.
__device__ inline void emit_values4(const DPlan& pl, const float* __restrict__ ws, long long i0, long long n, float (&v)[4]) {
  if (i0 + 3 < n) {
    for (int k = 0; k < pl.n_emit; ++k) {
      const DEmit& e = pl.e[k];
      if (i0 >= e.out_begin && i0 + 3 < e.out_begin + e.out_count) {
        long long off = -1;
        if (e.kind == ADVX_EMIT_PLAIN) {
          off = pl.canvas_off[e.stage] + (i0 - e.out_begin);
        } else {
          int c, y, x;
          emit_inverse(e, i0, c, y, x);
          const unsigned run = (e.kind == ADVX_EMIT_TILES) ? (unsigned)e.tile : (unsigned)e.patch;
          if ((unsigned)x % run + 3u < run) off = pl.canvas_off[e.stage] + ((long long)c * e.can_h + y) * e.can_w + x;
        }
        if (off >= 0) {
          if ((off & 3) == 0) {
            const float4 c4 = *reinterpret_cast<const float4*>(ws + off);
            v[0] = c4.x; v[1] = c4.y; v[2] = c4.z; v[3] = c4.w;
          } else {
            v[0] = ws[off]; v[1] = ws[off + 1]; v[2] = ws[off + 2]; v[3] = ws[off + 3];
          }
          return;
        }
        break;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? emit_value(pl, ws, i0 + k) : 0.0f;
}

// the batch loop of one float4 column: out[b, i0..i0+3] = v + sigma * noise, b0 <= b < b1
template <int NOISE, int IO, bool EDGE>
__device__ inline void emit_batch_loop(const float (&v)[4], bool vec, long long n, long long i0, long long q, int b0, int b1,
                                       float sigma, const float* __restrict__ unit_noise, unsigned long long seed,
                                       unsigned long long offset, void* __restrict__ out, long long live_lo, long long live_hi) {
  for (int b = b0; b < b1; ++b) {
    float o[4] = {v[0], v[1], v[2], v[3]};
    if (NOISE == 1) {
      if (vec) {
        float4 z = *reinterpret_cast<const float4*>(unit_noise + (size_t)b * n + i0);
        o[0] = v[0] + z.x * sigma; o[1] = v[1] + z.y * sigma; o[2] = v[2] + z.z * sigma; o[3] = v[3] + z.w * sigma;
      } else {
        for (int k = 0; k < 4; ++k)
          if (i0 + k < n) o[k] = v[k] + unit_noise[(size_t)b * n + i0 + k] * sigma;
      }
    } else if (NOISE == 2) {
      float4 z = philox_normal4_qb((uint32_t)q, (uint32_t)b, offset, seed);
      o[0] = v[0] + z.x * sigma; o[1] = v[1] + z.y * sigma; o[2] = v[2] + z.z * sigma; o[3] = v[3] + z.w * sigma;
    }
    if (EDGE) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i0 + k < live_lo || i0 + k >= live_hi) o[k] = 0.0f;
    }
    if (vec) {
      // write-once stream: non-temporal, like the fused forward; rounded once to the boundary dtype
      io_store4<IO>(out, (size_t)b * n + i0, make_float4(o[0], o[1], o[2], o[3]));
    } else {
      for (int k = 0; k < 4; ++k)
        if (i0 + k < n) reinterpret_cast<float*>(out)[(size_t)b * n + i0 + k] = o[k];
    }
  }
}

// NOISE: 0 none, 1 unit-noise tensor supplied, 2 in-kernel Philox
template <int NOISE, int IO>   // IO: boundary dtype of `out` (0 f32, 1 f16, 2 bf16; halves need n % 4 == 0)
__device__ inline void emit_body(const DPlan& pl, const float* __restrict__ ws, int batch, int b_per_slice,
                                 const float* __restrict__ sigma_dev, const float* __restrict__ unit_noise,
                                 unsigned long long seed, unsigned long long offset, void* __restrict__ out, long long q_lo,
                                 long long q_hi, long long live_lo, long long live_hi, unsigned block_x, unsigned block_y) {
  // Only the float4 columns [q_lo, q_hi) are written.  The default is all of them, noise
  // included on the constant padding tiles, as the reference does (attack_model.py:320 adds
  // randn_like to the WHOLE tensor).  A caller that keeps `out` across steps with its padding
  // zeroed once passes the columns the emits cover and [live_lo, live_hi) as their element
  // range: elements outside it inside a boundary column are written as exact zeros.
  const long long n = pl.out_numel;
  const long long q = q_lo + (long long)block_x * blockDim.x + threadIdx.x;
  if (q >= q_hi) return;
  const long long i0 = q << 2;
  float v[4];
  const bool vec = ((n & 3) == 0);
  if (vec && pl.n_emit == 1 && pl.e[0].kind == ADVX_EMIT_PLAIN && pl.e[0].out_begin == 0 && pl.e[0].out_count == n &&
      (pl.canvas_off[pl.e[0].stage] & 3) == 0) {
    // the output IS the canvas (LLaVA): one 16-byte load instead of four inverse layout maps
    const float4 c4 = *reinterpret_cast<const float4*>(ws + pl.canvas_off[pl.e[0].stage] + i0);
    v[0] = c4.x; v[1] = c4.y; v[2] = c4.z; v[3] = c4.w;
  } else {
    emit_values4(pl, ws, i0, n, v);
  }
  const float sigma = (NOISE != 0) ? sigma_dev[0] : 0.0f;
  const int b0 = (int)block_y * b_per_slice;
  const int b1 = min(batch, b0 + b_per_slice);
  // A column that straddles the live range zeroes its dead elements.  Two instances of the loop, not one with the test
  // inside: there the compiler turned the test into eleven selects per iteration in EVERY thread, a tenth of the
  // generator-bound loop's instructions.
  if ((i0 < live_lo) || (i0 + 4 > live_hi))
    emit_batch_loop<NOISE, IO, true>(v, vec, n, i0, q, b0, b1, sigma, unit_noise, seed, offset, out, live_lo, live_hi);
  else
    emit_batch_loop<NOISE, IO, false>(v, vec, n, i0, q, b0, b1, sigma, unit_noise, seed, offset, out, live_lo, live_hi);
}

template <int NOISE, int IO>
__global__ void __launch_bounds__(kBlock) k_emit(DPlan pl, const float* __restrict__ ws, int batch, int b_per_slice,
                                                 const float* __restrict__ sigma_dev, const float* __restrict__ unit_noise,
                                                 unsigned long long seed, unsigned long long offset,
                                                 void* __restrict__ out, long long q_lo, long long q_hi,
                                                 long long live_lo, long long live_hi, TapRider rider, int xmap) {
  // the crop window's transposed tap tables (read by the backward) built by the first workgroups of this launch, which
  // then go on with their own columns: hidden in a launch this long, where the image kernels they used to ride in
  // were extended by them (advx_forward_multi)
  if (blockIdx.y == 0) ride_taps(rider, blockIdx.x);
  emit_body<NOISE, IO>(pl, ws, batch, b_per_slice, sigma_dev, unit_noise, seed, offset, out, q_lo, q_hi, live_lo, live_hi,
                       xcd_column_block(blockIdx.x, gridDim.x, xmap), blockIdx.y);
}

// Cross-model runs: the emits of all plans in ONE launch (blockIdx.z = plan).  Each emit alone ends in a tail in which
// the last workgroups run on a mostly idle device; side by side the plans fill each other's tails.  Per element the
// work is k_emit's (same counters, same offsets): identical tensors.
struct EmitArgs {
  DPlan pl;
  const float* ws;
  const float* unit_noise;
  void* out;
  unsigned long long offset;
  long long q_lo, q_hi, live_lo, live_hi;
  int batch, b_per_slice, gx, slices, io;
};
struct MultiEmit {
  int n;
  EmitArgs a[kMaxMulti];
};
template <int NOISE>
__global__ void __launch_bounds__(kBlock) k_emit_multi(MultiEmit me, const float* __restrict__ sigma_dev, unsigned long long seed,
                                                       TapRider rider) {
  if (blockIdx.z == 0 && blockIdx.y == 0) ride_taps(rider, blockIdx.x);   // as in k_emit
#pragma unroll
  for (int k = 0; k < kMaxMulti; ++k) {
    if (k != (int)blockIdx.z) continue;          // static indices into the kernel arguments
    const EmitArgs& a = me.a[k];
    if ((int)blockIdx.x >= a.gx || (int)blockIdx.y >= a.slices) return;
    if (a.io == 0)
      emit_body<NOISE, 0>(a.pl, a.ws, a.batch, a.b_per_slice, sigma_dev, a.unit_noise, seed, a.offset, a.out, a.q_lo, a.q_hi,
                          a.live_lo, a.live_hi, blockIdx.x, blockIdx.y);
    else if (a.io == 1)
      emit_body<NOISE, 1>(a.pl, a.ws, a.batch, a.b_per_slice, sigma_dev, a.unit_noise, seed, a.offset, a.out, a.q_lo, a.q_hi,
                          a.live_lo, a.live_hi, blockIdx.x, blockIdx.y);
    else
      emit_body<NOISE, 2>(a.pl, a.ws, a.batch, a.b_per_slice, sigma_dev, a.unit_noise, seed, a.offset, a.out, a.q_lo, a.q_hi,
                          a.live_lo, a.live_hi, blockIdx.x, blockIdx.y);
  }
}

// ======================================================================== batch reduce
// out[i] = sum_b g[b, i].  Block = 64 float4 columns; the 4 waves split the batch and
// meet in LDS; every wave-level load is 1 KiB contiguous.  Summation order is fixed
// (b ascending inside a wave, waves 0..3) so the result is bitwise reproducible.
__device__ inline float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int IO = 0>
__device__ inline float4 batch_column_sum(const void* __restrict__ g, int batch, long long n, long long i0,
                                          int wid, int nw) {
  float4 a = make_float4(0, 0, 0, 0);
  int b = wid;
  // 8, then 4 independent loads in flight per lane (1 KiB per wave-instruction in float32)
  for (; b + 7 * nw < batch; b += 8 * nw) {
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = io_load4<IO>(g, (size_t)(b + k * nw) * n + i0);
#pragma unroll
    for (int k = 0; k < 8; ++k) a = f4add(a, v[k]);
  }
  for (; b + 3 * nw < batch; b += 4 * nw) {
    float4 v0 = io_load4<IO>(g, (size_t)b * n + i0);
    float4 v1 = io_load4<IO>(g, (size_t)(b + nw) * n + i0);
    float4 v2 = io_load4<IO>(g, (size_t)(b + 2 * nw) * n + i0);
    float4 v3 = io_load4<IO>(g, (size_t)(b + 3 * nw) * n + i0);
    a = f4add(f4add(f4add(f4add(a, v0), v1), v2), v3);
  }
  for (; b < batch; b += nw) a = f4add(a, io_load4<IO>(g, (size_t)b * n + i0));
  return a;
}

// Only the float4 columns [q_lo, q_hi) are reduced: the gradient of the constant padding tiles
// (llama32processor.py:344-346, phi3processor.py:232-235) is never read.
// CANVAS: the sums are stored in canvas order into the plan workspace `out` (gcan_dest) - one inverse layout
// map per column here instead of one per tap in the transposed resizes that read them; else out[i] = sum.
// -> workspace offset of flat index i0 if i0..i0+3 are four consecutive canvas elements of one emit, else -1
__device__ inline long long reduce_dest4(const DPlan& pl, long long i0) {
  // common case: the four indices lie in one emit and in one run of consecutive canvas elements (a whole PLAIN
  // canvas; a tile row whose width is a multiple of 4; the inside of a 14-pixel patch row) - ONE inverse map
  for (int k = 0; k < pl.n_emit; ++k) {
    const DEmit& e = pl.e[k];
    if (i0 >= e.out_begin && i0 + 3 < e.out_begin + e.out_count) {
      if (e.kind == ADVX_EMIT_PLAIN) return pl.gcan_off[e.stage] + (i0 - e.out_begin);
      int c, y, x, tc;
      emit_inverse_t(e, i0, c, y, x, tc);
      const unsigned run = (e.kind == ADVX_EMIT_TILES) ? (unsigned)e.tile : (unsigned)e.patch;
      if ((unsigned)x % run + 3u < run) return pl.gcan_off[e.stage] + (((long long)tc * 3 + c) * e.can_h + y) * e.can_w + x;
      return -1;
    }
  }
  return -1;
}
__device__ inline void reduce_store4(const DPlan& pl, float* __restrict__ ws, long long i0, long long d0, float4 t) {
  if (d0 >= 0) {
    if ((d0 & 3) == 0) {
      *reinterpret_cast<float4*>(ws + d0) = t;
    } else {
      ws[d0] = t.x; ws[d0 + 1] = t.y; ws[d0 + 2] = t.z; ws[d0 + 3] = t.w;
    }
    return;
  }
  const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long d = gcan_dest(pl, i0 + k);
    if (d >= 0) ws[d] = tv[k];
  }
}

template <int IO, bool CANVAS>   // io_load4 code: 0 / 1 / 2 cached f32 / f16 / bf16, 3 / 4 / 5 the same non-temporal
__device__ inline void reduce_body(const void* __restrict__ g, int batch, long long n, float* __restrict__ out, long long q_lo,
                                   long long q_hi, const DPlan& pl, unsigned block_x, unsigned grid_x) {
  __shared__ float4 part[kBlock / kWave][kWave];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const long long q = q_lo + (long long)block_x * kWave + lane;  // float4 column
  const long long n4 = n >> 2;
  // CANVAS: where this column's sums go (one inverse layout map), worked out while the stream is in flight
  long long dest = -1;
  if (CANVAS && wid == 0 && q < q_hi) dest = reduce_dest4(pl, q << 2);
  float4 a = make_float4(0, 0, 0, 0);
  if (q < q_hi) a = batch_column_sum<IO>(g, batch, n, q << 2, wid, kBlock / kWave);
  part[wid][lane] = a;
  __syncthreads();
  if (wid == 0 && q < q_hi) {
    float4 t = f4add(f4add(f4add(part[0][lane], part[1][lane]), part[2][lane]), part[3][lane]);
    if (CANVAS) reduce_store4(pl, out, q << 2, dest, t);
    else *reinterpret_cast<float4*>(out + (q << 2)) = t;
  }
  // scalar tail (n not a multiple of 4; float32 only): last block, first threads
  if ((IO == 0 || IO == 3) && q_hi == n4 && block_x == grid_x - 1) {
    long long tail0 = n4 << 2;
    long long i = tail0 + threadIdx.x;
    if (i < n) {
      float s = 0.0f;
      for (int b = 0; b < batch; ++b) s += reinterpret_cast<const float*>(g)[(size_t)b * n + i];
      if (CANVAS) {
        const long long d = gcan_dest(pl, i);
        if (d >= 0) out[d] = s;
      } else {
        out[i] = s;
      }
    }
  }
}

template <int IO, bool CANVAS>
__global__ void __launch_bounds__(kBlock) k_batch_reduce(const void* __restrict__ g, int batch, long long n,
                                                         float* __restrict__ out, long long q_lo, long long q_hi, DPlan pl,
                                                         int blocks, int xmap) {
  const int bx = xcd_block_of(blockIdx.x, blocks, xmap);      // XCD k reads a contiguous range of 64-column blocks
  if (bx < 0) return;
  reduce_body<IO, CANVAS>(g, batch, n, out, q_lo, q_hi, pl, (unsigned)bx, (unsigned)blocks);
}

// Cross-model runs: the batch reductions of all plans in ONE launch (blockIdx.y = plan), as k_emit_multi does for the
// emits: each reduction alone ends in a tail on a mostly idle device.  The plan's arguments are picked with a STATIC
// index (a kernel-argument array indexed at run time is copied to scratch memory: 850 us instead of 45).
struct ReduceArgs {
  DPlan pl;
  const void* g;
  float* out;
  long long n, q_lo, q_hi;
  int batch, blocks;
};
struct MultiReduce {
  int n, xmap;
  ReduceArgs a[kMaxMulti];
};
template <int IO>
__global__ void __launch_bounds__(kBlock) k_batch_reduce_multi(MultiReduce mr) {
#pragma unroll
  for (int k = 0; k < kMaxMulti; ++k) {
    if (k != (int)blockIdx.y) continue;
    const ReduceArgs& a = mr.a[k];
    const int bx = xcd_block_of(blockIdx.x, a.blocks, mr.xmap);
    if (bx < 0) return;
    reduce_body<IO, true>(a.g, a.batch, a.n, a.out, a.q_lo, a.q_hi, a.pl, (unsigned)bx, (unsigned)a.blocks);
  }
}

// rows not 16-byte aligned (n % 4 != 0): one thread per column, test-sized inputs only
template <bool CANVAS>
__global__ void __launch_bounds__(kBlock) k_batch_reduce_scalar(const float* __restrict__ g, int batch, long long n,
                                                                float* __restrict__ out, DPlan pl) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int b = 0; b < batch; ++b) s += g[(size_t)b * n + i];
    if (CANVAS) {
      const long long d = gcan_dest(pl, i);
      if (d >= 0) out[d] = s;
    } else {
      out[i] = s;
    }
  }
}

// ============================================================================== update
struct OptScalars {
  int kind, apply;
  float lr, decay, w1, beta2, w2, bias2_sqrt, eps, neg_step_size;
};

// The direction of the sign optimiser (ADVX_OPT_SIGN; north_star's "sign" update - the reference has AdamW only).  A gradient
// below the smallest NORMAL fp32 counts as zero.  Why: where a pixel's true gradient is of the order of 2^-149 - the far tail
// of a narrow blur kernel times the image-fit term - two correctly rounded evaluations of the same sum (this library's
// separable transposed blur, torchvision's k x k product kernel fl(w_i w_j)) land on 0 and on 1.4e-45 respectively; neither
// side flushes (tools/diag_denormal.py, profiles/r03/diag_denormal.log), it is one unit in the last place at the bottom of
// the format.  sign() turns that ulp into a whole lr step.  With the dead zone the discontinuity sits at 1.18e-38, where
// the two evaluations agree to 1e-7 relative.  oracle/pgd.py applies the same rule.
constexpr float kSignDeadZone = 1.17549435e-38f;
__device__ inline float sign_direction(float g) {
  return (g >= kSignDeadZone) ? 1.0f : ((g <= -kSignDeadZone) ? -1.0f : 0.0f);
}

// torch.optim.AdamW single-tensor arithmetic on one element
__device__ inline void adamw_element(float& p, float& m, float& v, float g, const OptScalars& o) {
  p = p * o.decay;                       // param.mul_(1 - lr*wd)
  m = m + o.w1 * (g - m);                // exp_avg.lerp_(grad, 1-beta1)   (weight < 0.5 form)
  v = v * o.beta2;                       // exp_avg_sq.mul_(beta2)
  v = v + (o.w2 * g) * g;                //   .addcmul_(grad, grad, value=1-beta2)
  float denom = sqrtf(v) / o.bias2_sqrt + o.eps;
  p = p + o.neg_step_size * (m / denom); // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ void __launch_bounds__(kBlock) k_update(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ grad, const float* __restrict__ mask,
                                                   long long n, OptScalars o, double* __restrict__ partials) {
  double acc[1] = {0.0};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float g = grad[i] * mask[i];          // attack_model.py:336
    grad[i] = g;
    acc[0] += (double)g * (double)g;
    if (o.apply) {
      if (o.kind == 0) {
        float pp = p[i], mm = m[i], vv = v[i];
        adamw_element(pp, mm, vv, g, o);
        p[i] = pp; m[i] = mm; v[i] = vv;
      } else {
        float sg = sign_direction(g);
        p[i] = p[i] - o.lr * sg;
      }
    }
  }
  block_sum_store<1>(acc, partials + blockIdx.x);
  if (blockIdx.x == 0 && threadIdx.x == 0) partials[2048] = (double)gridDim.x;
}

// nblk < 0: the producer (k_update, k_bwd_update, k_blur_bwd_fused) left its number of partials in slot kNormCountSlot
constexpr int kNormCountSlot = 2048;
__global__ void __launch_bounds__(kBlock) k_finalize_norm(const double* __restrict__ partials, int nblk,
                                                          float* __restrict__ stats) {
  if (nblk < 0) nblk = (int)partials[kNormCountSlot];
  finalize_norm_block<kFinU>(partials, nblk, stats);
}

// crop_bwd_value with a T x T window (T >= the transposed tables' REAL longest row, host: aa_transposed_rows_exact): every load
// issued before the first use, taps beyond a row's count from the last valid tap's address and never summed; crop_bwd_value's
// operations in its order (bit-identical)
template <int T>
__device__ inline float crop_bwd_value_w(const DStage& st, const float* __restrict__ gcan, int ci, int cj, int c, int y, int x) {
  const int ys = y - ci, xs = x - cj;
  float v = 0.0f;
  if (ys >= 0 && ys < st.src_h && xs >= 0 && xs < st.src_w) {
    const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
    const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
    const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
    const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
    const float* gp = gcan + (size_t)c * st.can_h * st.can_w;
    const int ly = max(oyc - 1, 0), lx = max(oxc - 1, 0);
    float wyv[T], wxv[T], r[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a) {
      wyv[a] = wy[min(a, ly)];
      wxv[a] = wx[min(a, lx)];
    }
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = 0; b < T; ++b) r[a][b] = gp[(size_t)(oy + min(a, ly)) * st.can_w + ox + min(b, lx)];
#pragma unroll
    for (int a = 0; a < T; ++a) {
      float h = 0.0f;
#pragma unroll
      for (int b = 0; b < T; ++b) h = (b < oxc) ? h + wxv[b] * r[a][b] : h;
      v = (a < oyc) ? v + wyv[a] * h : v;
    }
  }
  return v;
}

// Image-level backward tail and optimiser in ONE launch (no all-reduce in between): what
// [k_crop_bwd,] k_tanh_bwd and k_update do per element, in their order, on values kept in
// registers.  MODE 0: gradient of the image given (gs); 1: folded blur adjoint (c2); 2: gs is
// the gradient of the crop window's resize, gathered here (exact zeros outside the window).
template <int MODE, int T = 0>   // T > 0 (MODE 2): the window's transposed gather as a compiled window (crop_bwd_value_w)
__global__ void __launch_bounds__(kBlock) k_bwd_update(const float* __restrict__ s, const float* __restrict__ gs,
                                                       const float* __restrict__ c2, DStage crop_st, int ci, int cj, int H, int W,
                                                       int r, float eps, float c_fit, int accumulate, float* __restrict__ p,
                                                       float* __restrict__ m, float* __restrict__ v, float* __restrict__ grad,
                                                       const float* __restrict__ mask, OptScalars o,
                                                       double* __restrict__ partials) {
  const unsigned plane = (unsigned)H * (unsigned)W;
  const long long n = 3LL * plane;
  double acc[1] = {0.0};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float gx;
    if (MODE == 0) {
      gx = gs[i] + imgfit_grad(s[i], c_fit);
    } else {
      const int c = (int)((unsigned)i / plane);
      const unsigned rem = (unsigned)i - (unsigned)c * plane;
      const int y = (int)(rem / (unsigned)W), x = (int)(rem - (unsigned)y * (unsigned)W);
      if (MODE == 1) gx = blur_fold(c2 + (size_t)c * (H + 2 * r) * (W + 2 * r), H, W, r, y, x);
      else gx = ((T > 0) ? crop_bwd_value_w<(T > 0 ? T : 1)>(crop_st, gs, ci, cj, c, y, x) : crop_bwd_value(crop_st, gs, ci, cj, c, y, x)) +
                imgfit_grad(s[i], c_fit);
    }
    float pp = p[i];
    const float t = tanhf(pp);
    float g = (gx * eps) * (1.0f - t * t);
    if (accumulate) g = grad[i] + g;
    g = g * mask[i];                          // attack_model.py:336
    grad[i] = g;
    acc[0] += (double)g * (double)g;
    if (o.apply) {
      if (o.kind == 0) {
        float mm = m[i], vv = v[i];
        adamw_element(pp, mm, vv, g, o);
        p[i] = pp; m[i] = mm; v[i] = vv;
      } else {
        float sg = sign_direction(g);
        p[i] = pp - o.lr * sg;
      }
    }
  }
  block_sum_store<1>(acc, partials + blockIdx.x);
  if (blockIdx.x == 0 && threadIdx.x == 0) partials[2048] = (double)gridDim.x;
}

// k_stage_bwd3_w + k_bwd_update<0> in ONE launch (round 4): the transposed resize of a plan's stage 0 (or of the composed
// crop window o stage 0) gathered for the three channels of an image pixel and, on the values still in registers, image-fit',
// tanh', [accumulate], mask, ||g|| partial and the optimiser - what the two launches do per element, in their order (the
// gradient w.r.t. s is rounded to float exactly where k_stage_bwd3_w would have stored it).  One launch and one round trip of
// that gradient less per step of a single-plan chain without blur.  Grid (chunk of 256 pixels, rows_per_block image rows); at
// most 2048 workgroups (the ||g|| partials' room: the host picks rows_per_block for that).  T = 0: stage_bwd_value's loops.
template <int T, int MODE>
__global__ void __launch_bounds__(kBlock) k_collect_update3(DStage st, CanvasGrad cg, const float* __restrict__ s, float eps, float c_fit,
                                                            int accumulate, float* __restrict__ p, float* __restrict__ m,
                                                            float* __restrict__ v, float* __restrict__ grad,
                                                            const float* __restrict__ mask, OptScalars o,
                                                            double* __restrict__ partials, int rows_per_block, ImgGrid ig) {
  constexpr int TT = (T > 0) ? T : 1;
  constexpr int COPIES = (MODE == 3) ? 2 : 1;
  constexpr bool DG = MODE == 2;
  BlockXYZ blk;
  if (!xcd_band_block(ig, blk)) return;          // padding of the XCD-aware grid (the whole workgroup)
  const int xs = blk.x * blockDim.x + threadIdx.x;
  const unsigned row = blk.x + ig.gx * blk.y;
  const int y_end = min(st.src_h, (int)(blk.y + 1) * rows_per_block);
  double acc[1] = {0.0};
  if (xs < st.src_w)
  for (int ys = blk.y * rows_per_block; ys < y_end; ++ys) {
    const size_t plane_s = (size_t)st.src_h * st.src_w, o0 = (size_t)ys * st.src_w + xs;
    const bool adam = o.apply && o.kind == 0;
    float pp[3], sv[3], mk[3], g0[3], mm[3], vv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = (size_t)c * plane_s + o0;
      pp[c] = p[i]; sv[c] = s[i]; mk[c] = mask[i];
      g0[c] = accumulate ? grad[i] : 0.0f;
      mm[c] = adam ? m[i] : 0.0f;
      vv[c] = adam ? v[i] : 0.0f;
    }
    float gsum[3];
    if (T > 0) {
      const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
      const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
      const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
      const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
      const size_t plane = (size_t)st.can_h * st.can_w;
      const int ly = max(oyc - 1, 0), lx = max(oxc - 1, 0);
      float wyv[TT], wxv[TT], r[3][TT][TT][COPIES + 1];
#pragma unroll
      for (int a = 0; a < TT; ++a) { wyv[a] = wy[min(a, ly)]; wxv[a] = wx[min(a, lx)]; }
#pragma unroll
      for (int a = 0; a < TT; ++a) {
        const size_t rowc = (size_t)(st.off_y + oy + min(a, ly)) * st.can_w + st.off_x + ox;
#pragma unroll
        for (int b = 0; b < TT; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const size_t q = (size_t)c * plane + rowc + min(b, lx);
#pragma unroll
            for (int t = 0; t < COPIES; ++t) r[c][a][b][t] = cg.g[(size_t)t * cg.copy_stride + q];
            if (DG) r[c][a][b][COPIES] = cg.dgrad[q];
          }
      }
      gsum[0] = gsum[1] = gsum[2] = 0.0f;
#pragma unroll
      for (int a = 0; a < TT; ++a) {
        float h[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int b = 0; b < TT; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float g = 0.0f;
#pragma unroll
            for (int t = 0; t < COPIES; ++t) g += r[c][a][b][t];
            if (DG) g += r[c][a][b][COPIES];
            h[c] = (b < oxc) ? h[c] + wxv[b] * g : h[c];
          }
#pragma unroll
        for (int c = 0; c < 3; ++c) gsum[c] = (a < oyc) ? gsum[c] + wyv[a] * h[c] : gsum[c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) gsum[c] = st.normalise ? gsum[c] / st.stdv[c] : gsum[c];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) gsum[c] = stage_bwd_value(st, cg, c, ys, xs);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = (size_t)c * plane_s + o0;
      const float gx = gsum[c] + imgfit_grad(sv[c], c_fit);      // k_bwd_update<0>: gs[i] + imgfit'(s[i])
      float pv = pp[c];
      const float t = tanhf(pv);
      float g = (gx * eps) * (1.0f - t * t);
      if (accumulate) g = g0[c] + g;
      g = g * mk[c];                                             // attack_model.py:336
      grad[i] = g;
      acc[0] += (double)g * (double)g;
      if (o.apply) {
        if (o.kind == 0) {
          float m1 = mm[c], v1 = vv[c];
          adamw_element(pv, m1, v1, g, o);
          p[i] = pv; m[i] = m1; v[i] = v1;
        } else {
          float sg = sign_direction(g);
          p[i] = pv - o.lr * sg;
        }
      }
    }
  }
  block_sum_store<1>(acc, partials + row);
  if (row == 0 && threadIdx.x == 0) partials[2048] = (double)(ig.gx * ig.gy);
}

// ================================================================= fused (identity plan)
// LLaVA at native resolution: process() is (s - mean)/std with s = x0 + eps*tanh(p).
// The step is software-pipelined over two launches:
//   k_fused_fwd(t) : stream out[b,i] = v_t[i] + sigma_t * N(0,1) from the PREPARED v_t
//                    (blockIdx.y >= 1: one slice of the batch each).  The short blocks with
//                    blockIdx.y == 0 read s_t, x0 and leave the statistics partials of s_t;
//                    block (0,0) first reduces ||g|| of step t-1 (slot 7).
//                    sigma_t is slot QERR_STD, i.e. the quantise error of s_{t-1}.
//   k_fused_bwd(t) : reads the B gradients once, /std, image-fit term, tanh', mask, ||g||,
//                    optimiser, and - same thread, same registers - prepares step t+1 from
//                    the UPDATED p: s_{t+1}, v_{t+1}.  Block 0 reduces the statistics of s_t
//                    (rotating SIGMA <- old QERR_STD first), beside the other blocks' stream.
// k_fused_prep does the preparation alone (first step, or after p changed elsewhere).
// Per-step scalars in DEVICE memory, for hipGraph replay of the pair (advx_fused_*_sched): a captured launch
// cannot receive a new Philox offset or new optimiser scalars as kernel arguments, so the forward reads its step
// index from `fwd_step` (offset = offset_base + fwd_step) and the backward takes opt[bwd_step - first_step].  The
// hand-over needs no atomics: every block of the forward reads fwd_step while its block (0,0) publishes it as
// bwd_step; every block of the backward reads bwd_step while its block 0 writes fwd_step = bwd_step + 1; kernel
// boundaries order the rest.
struct SchedDev {
  unsigned long long fwd_step;
  unsigned long long bwd_step;
  unsigned long long first_step;   // step index of opt[0]
  int n_opt;                       // entries of opt[]; a backward beyond them reuses the last one
  int pad;
  // followed by OptScalars opt[n_opt]
};

struct FusedGeom {
  int plane;        // H*W, multiple of 4
  float mean[3];
  float stdv[3];
};

__global__ void __launch_bounds__(kBlock) k_fused_prep(const float* __restrict__ p, const float* __restrict__ x0,
                                                       float eps, FusedGeom geo, float* __restrict__ s_buf,
                                                       float* __restrict__ v_buf) {
  const long long n = 3LL * geo.plane;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int c = (int)(i / geo.plane);
    float s = x0[i] + eps * tanhf(p[i]);
    s_buf[i] = s;
    v_buf[i] = (s - geo.mean[c]) / geo.stdv[c];
  }
}

// NOISE: 0 none, 1 unit-noise tensor supplied, 2 in-kernel Philox, 3 in-kernel Philox addressed as k_fused_step_wave
// addresses it (one block = four consecutive batch rows of ONE pixel; the host launches four rows per thread): the
// forward the one-launch chain runs on its own - first step, first step after a resume - draws the noise the chain's
// own emission would have drawn, so a resumed run continues bit for bit
// LEAN (experiment, ADVX_TUNE_PAIR_LEAN): v_buf IS p; s and v are re-derived here from p and x0 with the expressions of
// k_fused_prep / k_fused_bwd (same bits), so that the backward need not store s_next / v_buf
template <int NOISE, int IO, bool SCHED = false, bool LEAN = false>   // SCHED: step scalars from device memory (graph replay), else arguments
__global__ void __launch_bounds__(kBlock) k_fused_fwd(const float* __restrict__ v_buf, const float* __restrict__ s_buf,
                                                      const float* __restrict__ x0, long long n, int batch,
                                                      int b_per_slice, float* stats, const float* __restrict__ unit_noise,
                                                      unsigned long long seed, unsigned long long offset,
                                                      void* __restrict__ out, FusedHeader* __restrict__ hdr,
                                                      double* __restrict__ img_partials,
                                                      const double* __restrict__ norm_partials, SchedDev* sched,
                                                      float eps, FusedGeom geo, int xmap) {
  const long long n4 = n >> 2;
  const unsigned cb = xcd_column_block(blockIdx.x, gridDim.x, xmap);      // column block of this workgroup
  const long long q = (long long)cb * blockDim.x + threadIdx.x;
  // gridDim.x may be padded to a multiple of 8 (host: pad_xcd); the column blocks that exist:
  const int col_blocks = (int)((n4 + blockDim.x - 1) / blockDim.x);
  if ((int)cb >= col_blocks) return;
  if (SCHED) {
    const unsigned long long t = sched->fwd_step;       // uniform: a scalar load
    offset += t;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) sched->bwd_step = t;
  }
  // sigma of THIS step = quantise error of the previous image = slot QERR_STD (1); it is
  // only rewritten by the next k_fused_bwd.  Block (0,0) below writes slot 7 only.
  const float sigma = (NOISE != 0) ? stats[1] : 0.0f;
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    int nb = hdr->norm_blocks;
    __syncthreads();  // every thread has read the header before thread 0 rewrites it
    if (nb > 0) finalize_norm_block(norm_partials, nb, stats);
    if (threadIdx.x == 0) {
      hdr->norm_blocks = 0;
      hdr->image_blocks = col_blocks;
    }
  }
  const long long i0 = q << 2;
  if (blockIdx.y == 0) {
    // slice 0 = statistics only: short blocks, dispatched first, so that no emission block
    // carries the double-precision reduction on its tail
    double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
    if (q < n4) {
      float4 sv;
      float4 xv = *reinterpret_cast<const float4*>(x0 + i0);
      if (LEAN) {
        const float4 pv = *reinterpret_cast<const float4*>(v_buf + i0);
        sv = make_float4(xv.x + eps * tanhf(pv.x), xv.y + eps * tanhf(pv.y), xv.z + eps * tanhf(pv.z), xv.w + eps * tanhf(pv.w));
      } else {
        sv = *reinterpret_cast<const float4*>(s_buf + i0);
      }
      stat_accumulate(sv.x, sv.x - xv.x, acc);
      stat_accumulate(sv.y, sv.y - xv.y, acc);
      stat_accumulate(sv.z, sv.z - xv.z, acc);
      stat_accumulate(sv.w, sv.w - xv.w, acc);
    }
    block_sum_store<kStatSlots>(acc, img_partials + (size_t)cb * kStatSlots);
    return;
  }
  if (q >= n4) return;
  float4 v = *reinterpret_cast<const float4*>(v_buf + i0);
  if (LEAN) {
    const float4 xv = *reinterpret_cast<const float4*>(x0 + i0);
    const int c = (int)(i0 / geo.plane);          // plane % 4 == 0: one channel per float4
    const float mu = geo.mean[c], sd = geo.stdv[c];
    v = make_float4(((xv.x + eps * tanhf(v.x)) - mu) / sd, ((xv.y + eps * tanhf(v.y)) - mu) / sd,
                    ((xv.z + eps * tanhf(v.z)) - mu) / sd, ((xv.w + eps * tanhf(v.w)) - mu) / sd);
  }
  const int b0 = ((int)blockIdx.y - 1) * b_per_slice;
  const int b1 = min(batch, b0 + b_per_slice);
  if (NOISE == 3) {
    // b_per_slice == 4 and b0 % 4 == 0 (host): rows b0..b0+3 of pixel i0+k come from ONE block
    const float4 z0 = philox_normal4_pixel((unsigned long long)i0, (unsigned)(b0 >> 2), offset, seed);
    const float4 z1 = philox_normal4_pixel((unsigned long long)i0 + 1, (unsigned)(b0 >> 2), offset, seed);
    const float4 z2 = philox_normal4_pixel((unsigned long long)i0 + 2, (unsigned)(b0 >> 2), offset, seed);
    const float4 z3 = philox_normal4_pixel((unsigned long long)i0 + 3, (unsigned)(b0 >> 2), offset, seed);
    const float zr[4][4] = {{z0.x, z1.x, z2.x, z3.x}, {z0.y, z1.y, z2.y, z3.y}, {z0.z, z1.z, z2.z, z3.z}, {z0.w, z1.w, z2.w, z3.w}};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (b0 + k < b1)
        io_store4<IO>(out, (size_t)(b0 + k) * n + i0, make_float4(v.x + zr[k][0] * sigma, v.y + zr[k][1] * sigma,
                                                                  v.z + zr[k][2] * sigma, v.w + zr[k][3] * sigma));
    }
    return;
  }
  for (int b = b0; b < b1; ++b) {
    float4 o = v;
    if (NOISE == 1) {
      float4 z = *reinterpret_cast<const float4*>(unit_noise + (size_t)b * n + i0);
      o = make_float4(v.x + z.x * sigma, v.y + z.y * sigma, v.z + z.z * sigma, v.w + z.w * sigma);
    } else if (NOISE == 2) {
      float4 z = philox_normal4_qb((uint32_t)q, (uint32_t)b, offset, seed);
      o = make_float4(v.x + z.x * sigma, v.y + z.y * sigma, v.z + z.z * sigma, v.w + z.w * sigma);
    }
    // write-once stream: non-temporal stores keep it out of the way of grad_out in the
    // Infinity Cache (measured -1 us on this kernel)
    io_store4<IO>(out, (size_t)b * n + i0, o);
  }
}

// backward (+ update + preparation of the next forward) in one pass over grad_out:
// block = 64 float4 columns (256 pixels); the 4 waves split the batch and meet in LDS; then
// thread t owns pixel t of the block.  Its per-pixel state is prefetched before the batch
// loop so that no dependent load sits on the tail.
template <bool UPDATE, int IO, bool SCHED = false, bool LEAN = false>   // LEAN: no grad_p / s_next / v_buf stores (experiment)
__global__ void __launch_bounds__(kBlock) k_fused_bwd(const void* __restrict__ g, int batch, float* __restrict__ p,
                                                      const float* __restrict__ x0, float eps, FusedGeom geo,
                                                      float c_fit, const float* __restrict__ mask,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      float* __restrict__ grad_p, OptScalars o,
                                                      float* __restrict__ s_next, float* __restrict__ v_buf,
                                                      double* __restrict__ norm_partials, float* __restrict__ stats,
                                                      FusedHeader* __restrict__ hdr,
                                                      const double* __restrict__ img_partials, SchedDev* sched, int xmap) {
  __shared__ float4 part4[kBlock / kWave][kWave];
  // XCD-aware block -> column mapping (xcd_column_block; the grid may be padded to a multiple of 8): XCD k works on a
  // contiguous range of 64-column blocks instead of every eighth one
  const unsigned bx = xcd_column_block(blockIdx.x, gridDim.x, xmap);
  const int col_blocks = (int)(((3LL * geo.plane >> 2) + kWave - 1) / kWave);
  if ((int)bx >= col_blocks) return;
  if (SCHED) {
    const unsigned long long t = sched->bwd_step;
    long long k = (long long)(t - sched->first_step);
    if (k < 0) k = 0;
    if (k > sched->n_opt - 1) k = sched->n_opt - 1;
    o = reinterpret_cast<const OptScalars*>(sched + 1)[k];
    if (bx == 0 && threadIdx.x == 0) sched->fwd_step = t + 1;
  }
  const long long n = 3LL * geo.plane;
  const long long n4 = n >> 2;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const long long q = (long long)bx * kWave + lane;
  const long long i = (long long)bx * (kWave * 4) + threadIdx.x;
  // prefetch this thread's pixel state (latency hides under the batch stream)
  float pp = 0.f, xv = 0.f, mk = 0.f, mm = 0.f, vv = 0.f;
  if (i < n) {
    pp = p[i];
    xv = x0[i];
    if (UPDATE) {
      mk = mask[i];
      if (o.apply && o.kind == 0) {
        mm = m[i];
        vv = v[i];
      }
    }
  }
  // everything that does not need the gradient sum is computed while the stream is in flight
  const int c = (i < n) ? (int)(i / geo.plane) : 0;
  const float sd = geo.stdv[c];
  const float t = tanhf(pp);
  const float s = xv + eps * t;
  const float fit = imgfit_grad(s, c_fit);
  const float dtanh = 1.0f - t * t;
  float4 a = make_float4(0, 0, 0, 0);
  if (q < n4) a = batch_column_sum<IO>(g, batch, n, q << 2, wid, kBlock / kWave);
  if (bx == 0) {
    // statistics partials left by this step's forward: reduce them here, beside the other
    // blocks' streaming work (rotates SIGMA <- old QERR_STD, then QERR_STD <- new)
    int ib = hdr->image_blocks;
    __syncthreads();
    if (ib > 0) finalize_image_block<true>(img_partials, ib, n, stats);
    if (threadIdx.x == 0) {
      hdr->image_blocks = 0;
      if (UPDATE) hdr->norm_blocks = col_blocks;
    }
  }
  part4[wid][lane] = a;
  __syncthreads();
  const float(*part)[kWave * 4] = reinterpret_cast<const float(*)[kWave * 4]>(&part4[0][0]);
  double nacc[1] = {0.0};
  if (i < n) {
    float gs = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
    float gx = gs / sd + fit;
    float gp = (gx * eps) * dtanh;
    if (UPDATE) {
      gp = gp * mk;
      nacc[0] = (double)gp * (double)gp;
      if (!LEAN) grad_p[i] = gp;
      if (o.apply) {
        if (o.kind == 0) {
          adamw_element(pp, mm, vv, gp, o);
          p[i] = pp; m[i] = mm; v[i] = vv;
        } else {
          float sg = sign_direction(gp);
          pp = pp - o.lr * sg;
          p[i] = pp;
        }
      }
      if (!LEAN) {
        // prepare the next forward from the UPDATED p
        float sn = xv + eps * tanhf(pp);
        s_next[i] = sn;
        v_buf[i] = (sn - geo.mean[c]) / sd;
      }
    } else {
      grad_p[i] = gp;
    }
  }
  if (UPDATE) block_sum_store<1>(nacc, norm_partials + bx);
}

// data-parallel tail of the pair: after the all-reduce of grad_p every rank runs this ONE
// launch - mask, ||g|| partial, optimiser, and the preparation of the next forward (s, v) -
// so that a DP step is fwd + bwd(grad only) + all-reduce + this (the norm reduction rides in
// the next forward like in the single-GPU pair).  COMM: `grad` is the recv buffer of the peer
// exchange; every block first waits until all peers have posted their slices (advx_comm.h).
template <bool COMM>
__global__ void __launch_bounds__(kBlock) k_fused_update(float* __restrict__ p, float* __restrict__ m,
                                                         float* __restrict__ v, float* __restrict__ grad,
                                                         const float* __restrict__ mask, const float* __restrict__ x0,
                                                         float eps, FusedGeom geo, OptScalars o,
                                                         float* __restrict__ s_next, float* __restrict__ v_buf,
                                                         double* __restrict__ norm_partials,
                                                         FusedHeader* __restrict__ hdr, CommDev comm) {
  const long long n = 3LL * geo.plane;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  double nacc[1] = {0.0};
  // state that does not depend on the exchange is fetched before the rendezvous
  float mk = 0.f, pp0 = 0.f, xv0 = 0.f;
  if (i < n) {
    mk = mask[i];
    pp0 = p[i];
    xv0 = x0[i];
  }
  if (COMM) comm_wait_b(comm);
  if (i < n) {
    // COMM: recv is rewritten by the peers every step - read it past every cache
    float g = (COMM ? comm_load1(grad + i) : grad[i]) * mk;
    grad[i] = g;
    nacc[0] = (double)g * (double)g;
    float pp = pp0;
    if (o.kind == 0) {
      float mm = m[i], vv = v[i];
      adamw_element(pp, mm, vv, g, o);
      p[i] = pp; m[i] = mm; v[i] = vv;
    } else {
      float sg = sign_direction(g);
      pp = pp - o.lr * sg;
      p[i] = pp;
    }
    const int c = (int)(i / geo.plane);
    float sn = xv0 + eps * tanhf(pp);
    s_next[i] = sn;
    v_buf[i] = (sn - geo.mean[c]) / geo.stdv[c];
  }
  block_sum_store<1>(nacc, norm_partials + blockIdx.x);
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr->norm_blocks = gridDim.x;
}

// ================================================== prepared chain (any plan, no blur/crop)
// The fused pair's software pipelining for plans whose process() DOES resample (LLaVA from a
// non-native image, Mllama, Qwen2-VL): the backward of step t leaves the canvas of step t+1 in
// the plan workspace, so a step is four launches instead of nine:
//   k_emit          out = canvas_t (+ sigma_t N(0,1)), sigma_t = slot QERR_STD (previous image)
//   k_batch_reduce  gsum = sum_b grad_out (only the columns an image reaches)
//   k_plan_tail     per SOURCE pixel: resize^T (/std), image-fit', tanh', mask, ||g|| partial,
//                   optimiser -> p_{t+1}; s_{t+1} = x0 + eps*tanh(p_{t+1}) and its statistics
//                   partials; block 0 reduces the statistics of s_t (rotating SIGMA <- QERR_STD)
//   k_plan_head     canvas_{t+1} = resize(s_{t+1}), pad, normalise; block 0 reduces ||g||
// A plan with a second stage fed by the first canvas (Phi-3.5: bicubic global view of the HD
// canvas) adds k_stage_bwd of that stage before the tail (its gradient reaches the tail through
// `dgrad`) and k_stage_fwd of it after the head.
// T, MODE: the transposed gather as a compiled window (stage_bwd_value_w); T = 0: stage_bwd_value's run-time loops
template <int T, int MODE>
__global__ void __launch_bounds__(kBlock) k_plan_tail(DStage st, CanvasGrad cg, float* __restrict__ p, const float* __restrict__ x0, float eps,
                                                      float c_fit, const float* __restrict__ mask, float* __restrict__ m,
                                                      float* __restrict__ v, float* __restrict__ grad_p, OptScalars o,
                                                      float* __restrict__ s_next, double* __restrict__ img_rows_out,
                                                      double* __restrict__ norm_rows, const double* __restrict__ img_rows_in,
                                                      int img_rows_in_count, float* __restrict__ stats) {
  const unsigned plane = (unsigned)st.src_h * (unsigned)st.src_w;
  const long long n = 3LL * plane;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && img_rows_in_count > 0) finalize_image_block<true, kFinImgU>(img_rows_in, img_rows_in_count, n, stats);
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  double nacc[1] = {0.0};
  if (i < n) {
    float pp = p[i];
    const float xv = x0[i], mk = mask[i];
    float mm = 0.f, vv = 0.f;
    if (o.kind == 0) {
      mm = m[i];
      vv = v[i];
    }
    const int c = (int)((unsigned)i / plane);
    const unsigned rem = (unsigned)i - (unsigned)c * plane;
    const int ys = (int)(rem / (unsigned)st.src_w), xs = (int)(rem - (unsigned)ys * (unsigned)st.src_w);
    const float t = tanhf(pp);
    const float s = xv + eps * t;
    const float gs = (T > 0) ? stage_bwd_value_w<(T > 0 ? T : 1), (MODE > 0 ? MODE : 1)>(st, cg, c, ys, xs) : stage_bwd_value(st, cg, c, ys, xs);
    float gp = ((gs + imgfit_grad(s, c_fit)) * eps) * (1.0f - t * t);
    gp = gp * mk;
    nacc[0] = (double)gp * (double)gp;
    grad_p[i] = gp;
    if (o.kind == 0) {
      adamw_element(pp, mm, vv, gp, o);
      p[i] = pp; m[i] = mm; v[i] = vv;
    } else {
      float sg = sign_direction(gp);
      pp = pp - o.lr * sg;
      p[i] = pp;
    }
    const float xn = eps * tanhf(pp);
    const float sn = xv + xn;
    s_next[i] = sn;
    stat_accumulate(sn, xn, acc);
  }
  block_sum_store2<kStatSlots, 1>(acc, img_rows_out + (size_t)blockIdx.x * kStatSlots, nacc, norm_rows + blockIdx.x);
}

// Data parallelism splits the tail around the exchange of the image gradient:
//   k_plan_tail_grad : resize^T (/std), image-fit', tanh' -> this rank's UNMASKED gradient
//                      (block 0 still reduces the statistics of the current image);
//   <all-reduce>
//   k_plan_update    : mask, ||g|| partial, optimiser, s_next and its statistics partials.
template <int T = 0, int MODE = 0>   // as k_plan_tail
__global__ void __launch_bounds__(kBlock) k_plan_tail_grad(DStage st, CanvasGrad cg, const float* __restrict__ p, const float* __restrict__ x0,
                                                           float eps, float c_fit, float* __restrict__ grad_p,
                                                           const double* __restrict__ img_rows_in, int img_rows_in_count,
                                                           float* __restrict__ stats) {
  const unsigned plane = (unsigned)st.src_h * (unsigned)st.src_w;
  const long long n = 3LL * plane;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && img_rows_in_count > 0) finalize_image_block<true, kFinImgU>(img_rows_in, img_rows_in_count, n, stats);
  if (i < n) {
    const int c = (int)((unsigned)i / plane);
    const unsigned rem = (unsigned)i - (unsigned)c * plane;
    const int ys = (int)(rem / (unsigned)st.src_w), xs = (int)(rem - (unsigned)ys * (unsigned)st.src_w);
    const float t = tanhf(p[i]);
    const float s = x0[i] + eps * t;
    const float gs = (T > 0) ? stage_bwd_value_w<(T > 0 ? T : 1), (MODE > 0 ? MODE : 1)>(st, cg, c, ys, xs) : stage_bwd_value(st, cg, c, ys, xs);
    grad_p[i] = ((gs + imgfit_grad(s, c_fit)) * eps) * (1.0f - t * t);
  }
}

template <bool COMM>   // COMM: `grad` is the recv buffer of the peer exchange (advx_comm.h)
__global__ void __launch_bounds__(kBlock) k_plan_update(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ grad, const float* __restrict__ mask,
                                                        const float* __restrict__ x0, float eps, long long n, OptScalars o,
                                                        float* __restrict__ s_next, double* __restrict__ img_rows_out,
                                                        double* __restrict__ norm_rows, CommDev comm) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  double nacc[1] = {0.0};
  float mk = 0.f, pp = 0.f, xv = 0.f;
  if (i < n) {
    mk = mask[i];
    pp = p[i];
    xv = x0[i];
  }
  if (COMM) comm_wait_b(comm);
  if (i < n) {
    const float gp = (COMM ? comm_load1(grad + i) : grad[i]) * mk;
    grad[i] = gp;
    nacc[0] = (double)gp * (double)gp;
    if (o.kind == 0) {
      float mm = m[i], vv = v[i];
      adamw_element(pp, mm, vv, gp, o);
      p[i] = pp; m[i] = mm; v[i] = vv;
    } else {
      float sg = sign_direction(gp);
      pp = pp - o.lr * sg;
      p[i] = pp;
    }
    const float xn = eps * tanhf(pp);
    const float sn = xv + xn;
    s_next[i] = sn;
    stat_accumulate(sn, xn, acc);
  }
  block_sum_store2<kStatSlots, 1>(acc, img_rows_out + (size_t)blockIdx.x * kStatSlots, nacc, norm_rows + blockIdx.x);
}

// ---- the prepared chain's image kernels for LARGE images (host: 250 k positions and more), three channels per thread
// workgroup = (chunk of kBlock pixels of one row, row), thread = the three channels of one pixel, partial row = chunk + chunks * row.
// The taps and weights of the transposed resize are looked up once for the three channels, no thread divides to find its pixel
// and a workgroup's block sums serve three elements per thread: k_plan_tail 11.4 -> 9.8 us at 512 x 512, 36 -> 26 at 1024 x 1024
// (tools/exp_tail3.hip; at 336 x 336 the flat form is faster, as for the plain transposed resizes).  k_prep_rows3 and
// k_plan_update3 run on the SAME partition, so a run that re-prepares (first step, resume) or splits the tail around an all-reduce
// sums exactly what an uninterrupted single-rank run sums.  Per element the arithmetic is the flat kernels'.
__global__ void __launch_bounds__(kBlock) k_prep_rows3(const float* __restrict__ p, const float* __restrict__ x0, float eps, int H,
                                                       int W, float* __restrict__ out, double* __restrict__ partials) {
  const int ys = blockIdx.y, xs = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned row = blockIdx.x + gridDim.x * blockIdx.y;
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  if (xs < W) {
    const size_t plane = (size_t)H * W, o0 = (size_t)ys * W + xs;
    float pv[3], xv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { pv[c] = p[(size_t)c * plane + o0]; xv[c] = x0[(size_t)c * plane + o0]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = eps * tanhf(pv[c]);
      const float sv = xv[c] + x;
      out[(size_t)c * plane + o0] = sv;
      stat_accumulate(sv, x, acc);
    }
  }
  block_sum_store<kStatSlots>(acc, partials + (size_t)row * kStatSlots);
}

// T, MODE as k_plan_tail's (T = 0: stage_bwd_value's loops, channel by channel)
template <int T, int MODE>
__global__ void __launch_bounds__(kBlock) k_plan_tail3(DStage st, CanvasGrad cg, float* __restrict__ p, const float* __restrict__ x0, float eps,
                                                       float c_fit, const float* __restrict__ mask, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ grad_p, OptScalars o,
                                                       float* __restrict__ s_next, double* __restrict__ img_rows_out,
                                                       double* __restrict__ norm_rows, const double* __restrict__ img_rows_in,
                                                       int img_rows_in_count, float* __restrict__ stats) {
  constexpr int TT = (T > 0) ? T : 1;
  constexpr int COPIES = (MODE == 3) ? 2 : 1;
  constexpr bool DG = MODE == 2;
  const int ys = blockIdx.y;
  const int xs = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned row = blockIdx.x + gridDim.x * blockIdx.y;
  if (row == 0 && img_rows_in_count > 0)
    finalize_image_block<true, kFinImgU>(img_rows_in, img_rows_in_count, 3LL * st.src_h * st.src_w, stats);
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  double nacc[1] = {0.0};
  if (xs < st.src_w) {
    const size_t plane_s = (size_t)st.src_h * st.src_w, o0 = (size_t)ys * st.src_w + xs;
    float pp[3], xv[3], mk[3], mm[3], vv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = (size_t)c * plane_s + o0;
      pp[c] = p[i]; xv[c] = x0[i]; mk[c] = mask[i];
      mm[c] = (o.kind == 0) ? m[i] : 0.0f;
      vv[c] = (o.kind == 0) ? v[i] : 0.0f;
    }
    float gsum[3];
    if (T > 0) {
      // the transposed gather of k_stage_bwd3_w: the whole window of the three channels in flight
      const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
      const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
      const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
      const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
      const size_t plane = (size_t)st.can_h * st.can_w;
      const int ly = max(oyc - 1, 0), lx = max(oxc - 1, 0);
      float wyv[TT], wxv[TT], r[3][TT][TT][COPIES + 1];
#pragma unroll
      for (int a = 0; a < TT; ++a) { wyv[a] = wy[min(a, ly)]; wxv[a] = wx[min(a, lx)]; }
#pragma unroll
      for (int a = 0; a < TT; ++a) {
        const size_t rowc = (size_t)(st.off_y + oy + min(a, ly)) * st.can_w + st.off_x + ox;
#pragma unroll
        for (int b = 0; b < TT; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const size_t q = (size_t)c * plane + rowc + min(b, lx);
#pragma unroll
            for (int t = 0; t < COPIES; ++t) r[c][a][b][t] = cg.g[(size_t)t * cg.copy_stride + q];
            if (DG) r[c][a][b][COPIES] = cg.dgrad[q];
          }
      }
      gsum[0] = gsum[1] = gsum[2] = 0.0f;
#pragma unroll
      for (int a = 0; a < TT; ++a) {
        float h[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int b = 0; b < TT; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float g = 0.0f;                                // canvas_grad_at: the copies in order, then dgrad
#pragma unroll
            for (int t = 0; t < COPIES; ++t) g += r[c][a][b][t];
            if (DG) g += r[c][a][b][COPIES];
            h[c] = (b < oxc) ? h[c] + wxv[b] * g : h[c];
          }
#pragma unroll
        for (int c = 0; c < 3; ++c) gsum[c] = (a < oyc) ? gsum[c] + wyv[a] * h[c] : gsum[c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) gsum[c] = st.normalise ? gsum[c] / st.stdv[c] : gsum[c];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) gsum[c] = stage_bwd_value(st, cg, c, ys, xs);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = (size_t)c * plane_s + o0;
      float pv = pp[c];
      const float t = tanhf(pv);
      const float sv = xv[c] + eps * t;
      float gp = ((gsum[c] + imgfit_grad(sv, c_fit)) * eps) * (1.0f - t * t);
      gp = gp * mk[c];
      nacc[0] += (double)gp * (double)gp;
      grad_p[i] = gp;
      if (o.kind == 0) {
        float m1 = mm[c], v1 = vv[c];
        adamw_element(pv, m1, v1, gp, o);
        p[i] = pv; m[i] = m1; v[i] = v1;
      } else {
        float sg = sign_direction(gp);
        pv = pv - o.lr * sg;
        p[i] = pv;
      }
      const float xn = eps * tanhf(pv);
      const float sn = xv[c] + xn;
      s_next[i] = sn;
      stat_accumulate(sn, xn, acc);
    }
  }
  block_sum_store2<kStatSlots, 1>(acc, img_rows_out + (size_t)row * kStatSlots, nacc, norm_rows + row);
}

template <bool COMM>   // k_plan_update on the three-channel partition
__global__ void __launch_bounds__(kBlock) k_plan_update3(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ grad, const float* __restrict__ mask,
                                                         const float* __restrict__ x0, float eps, int H, int W, OptScalars o,
                                                         float* __restrict__ s_next, double* __restrict__ img_rows_out,
                                                         double* __restrict__ norm_rows, CommDev comm) {
  const int ys = blockIdx.y, xs = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned row = blockIdx.x + gridDim.x * blockIdx.y;
  const size_t plane = (size_t)H * W, o0 = (size_t)ys * W + xs;
  const bool live = xs < W;
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  double nacc[1] = {0.0};
  float mk[3] = {0.f, 0.f, 0.f}, pp[3] = {0.f, 0.f, 0.f}, xv[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = (size_t)c * plane + o0;
      mk[c] = mask[i]; pp[c] = p[i]; xv[c] = x0[i];
    }
  }
  if (COMM) comm_wait_b(comm);
  if (live) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = (size_t)c * plane + o0;
      const float gp = (COMM ? comm_load1(grad + i) : grad[i]) * mk[c];
      grad[i] = gp;
      nacc[0] += (double)gp * (double)gp;
      float pv = pp[c];
      if (o.kind == 0) {
        float mm = m[i], vv = v[i];
        adamw_element(pv, mm, vv, gp, o);
        p[i] = pv; m[i] = mm; v[i] = vv;
      } else {
        float sg = sign_direction(gp);
        pv = pv - o.lr * sg;
        p[i] = pv;
      }
      const float xn = eps * tanhf(pv);
      const float sn = xv[c] + xn;
      s_next[i] = sn;
      stat_accumulate(sn, xn, acc);
    }
  }
  block_sum_store2<kStatSlots, 1>(acc, img_rows_out + (size_t)row * kStatSlots, nacc, norm_rows + row);
}

__global__ void __launch_bounds__(kBlock) k_plan_head(DStage st, const float* __restrict__ src, long long src_cstride,
                                                      int src_rstride, float* __restrict__ canvas,
                                                      const double* __restrict__ norm_rows, int norm_count,
                                                      float* __restrict__ stats) {
  if (blockIdx.x == 0 && norm_count > 0) finalize_norm_block<kFinU>(norm_rows, norm_count, stats);
  long long n = 3LL * st.can_h * st.can_w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    int c = (int)((unsigned)i / ((unsigned)st.can_h * (unsigned)st.can_w));
    int rem = (int)((unsigned)i - (unsigned)c * (unsigned)st.can_h * (unsigned)st.can_w);
    int y = rem / st.can_w, x = rem - y * st.can_w;
    canvas[i] = stage_fwd_value(st, src, src_cstride, src_rstride, c, y, x);
  }
}

// ------------------------------------------------------------------- one launch per step
// k_fused_step_wave = k_fused_bwd(t) + k_fused_fwd(t+1) for the same pixels in ONE launch:
// grad_out of step t in, pixel_values of step t+1 out.
// WAVE-INDEPENDENT: lane = one pixel; a wave owns 64 pixels (256-byte rows) and, for them,
//   A  sums grad_out over the whole batch (fixed order, 8 loads in flight per lane),
//   B  /std, image-fit term, tanh', mask, ||g||, optimiser -> p_{t+1}; s_{t+1}, v_{t+1};
//      statistics of s_{t+1} accumulate in registers,
//   C  emits out_{t+1}[b, pixel] = v_{t+1} + sigma_{t+1} N(0,1) for every b (one Philox block
//      serves four consecutive b of the lane's pixel).
// No LDS and no barrier between A, B and C, so waves drift apart and the load stream (A) of
// some waves overlaps the noise generation and store stream (C) of others on the same SIMD -
// what a workgroup-synchronous version (4 waves splitting the batch, LDS combine) could not
// do: it kept every workgroup in lockstep and was slower than two separate launches.
// sigma_{t+1} = std(|q(s_t) - s_t|) needs a global reduction over s_t: its partial rows were
// left by the PREVIOUS launch (kernel boundary = visibility) and every block re-reduces the
// two columns it needs in the same fixed order - redundant L2 reads instead of a grid barrier.
// Rows are double-buffered because early blocks of this launch write the next rows while late
// blocks still read the current ones.  Block 0 also publishes the statistics of s_t and
// ||g_{t-1}|| to `stats`.
struct StepRows {
  const double* img_in;    // [img_rows_in][kStatSlots]  statistics partials of s_t
  int img_rows_in;
  double* img_out;         // [gridDim.x][kStatSlots]    statistics partials of s_{t+1}
  const double* norm_in;   // [norm_rows_in]             sum-of-squares partials of step t-1 (0 rows: none)
  int norm_rows_in;
  double* norm_out;        // [gridDim.x]
};

template <int NOISE>
__global__ void __launch_bounds__(kBlock) k_fused_step_wave(const float* __restrict__ g, int batch, float* __restrict__ p,
                                                            const float* __restrict__ x0, float eps, FusedGeom geo,
                                                            float c_fit, const float* __restrict__ mask,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            float* __restrict__ grad_p, OptScalars o,
                                                            float* __restrict__ s_next, float* __restrict__ v_buf,
                                                            const float* __restrict__ unit_noise, unsigned long long seed,
                                                            unsigned long long offset, float* __restrict__ out,
                                                            StepRows rows, float* __restrict__ stats, long long n_groups) {
  __shared__ double sig2[2];
  const long long n = 3LL * geo.plane;
  const int lane = threadIdx.x & (kWave - 1);
  const long long wave = ((long long)blockIdx.x * kBlock + threadIdx.x) / kWave;
  const long long nwaves = (long long)gridDim.x * (kBlock / kWave);
  // ---- sigma_{t+1}: fixed-order reduction of columns 0,1 of the current partial rows
  {
    double acc[2] = {0.0, 0.0};
    for (int r = threadIdx.x; r < rows.img_rows_in; r += kBlock) {
      acc[0] += rows.img_in[(size_t)r * kStatSlots + 0];
      acc[1] += rows.img_in[(size_t)r * kStatSlots + 1];
    }
    block_sum_store<2>(acc, sig2);
  }
  const double N = (double)n;
  const double var_d = (n > 1) ? (sig2[1] - sig2[0] * sig2[0] / N) / (N - 1.0) : 0.0;
  const float sigma = (float)sqrt(var_d > 0.0 ? var_d : 0.0);
  if (blockIdx.x == 0) {
    finalize_image_block<true>(rows.img_in, rows.img_rows_in, n, stats);
    if (rows.norm_rows_in > 0) finalize_norm_block(rows.norm_in, rows.norm_rows_in, stats);
  }
  double nacc[1] = {0.0};
  double sacc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  for (long long grp = wave; grp < n_groups; grp += nwaves) {
    const long long i = grp * kWave + lane;
    if (i >= n) continue;
    // ---- A: this pixel's state and its gradient column
    float pp = p[i];
    const float xv = x0[i];
    const float mk = mask[i];
    float mm = 0.f, vv = 0.f;
    if (o.kind == 0) {
      mm = m[i];
      vv = v[i];
    }
    float gs = 0.0f;
    {
      int b = 0;
      for (; b + 8 <= batch; b += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = g[(size_t)(b + k) * n + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) gs += t[k];
      }
      for (; b < batch; ++b) gs += g[(size_t)b * n + i];
    }
    // ---- B: update and preparation
    const int c = (int)(i / geo.plane);
    const float sd = geo.stdv[c];
    float t = tanhf(pp);
    float s = xv + eps * t;
    float gx = gs / sd + imgfit_grad(s, c_fit);
    float gp = ((gx * eps) * (1.0f - t * t)) * mk;
    nacc[0] += (double)gp * (double)gp;
    grad_p[i] = gp;
    if (o.kind == 0) {
      adamw_element(pp, mm, vv, gp, o);
      p[i] = pp; m[i] = mm; v[i] = vv;
    } else {
      float sg = sign_direction(gp);
      pp = pp - o.lr * sg;
      p[i] = pp;
    }
    const float xn = eps * tanhf(pp);
    const float sn = xv + xn;
    const float vn = (sn - geo.mean[c]) / sd;
    s_next[i] = sn;
    v_buf[i] = vn;
    stat_accumulate(sn, xn, sacc);
    // ---- C: emit this pixel for every b
    for (int b = 0; b < batch; b += 4) {
      float z[4] = {0.f, 0.f, 0.f, 0.f};
      if (NOISE == 2) {
        float4 zz = philox_normal4_pixel((unsigned long long)i, (unsigned)(b >> 2), offset, seed);
        z[0] = zz.x; z[1] = zz.y; z[2] = zz.z; z[3] = zz.w;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (b + k < batch) {
          if (NOISE == 1) z[k] = unit_noise[(size_t)(b + k) * n + i];
          __builtin_nontemporal_store(vn + z[k] * sigma, out + (size_t)(b + k) * n + i);
        }
      }
    }
  }
  block_sum_store<1>(nacc, rows.norm_out + blockIdx.x);
  block_sum_store<kStatSlots>(sacc, rows.img_out + (size_t)blockIdx.x * kStatSlots);
}

// finalise pending rows on demand (host reads stats)
__global__ void __launch_bounds__(kBlock) k_step_flush(const double* __restrict__ norm_rows, int n_norm,
                                                       float* __restrict__ stats) {
  if (n_norm > 0) finalize_norm_block<kFinU>(norm_rows, n_norm, stats);
}

// the generator on its own (tests, advx_philox_normal): batch row 0 of the stream k_emit uses
__global__ void __launch_bounds__(kBlock) k_philox_normal(float* __restrict__ out, long long n,
                                                          unsigned long long seed, unsigned long long offset) {
  long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n4 = (n + 3) >> 2;
  if (q >= n4) return;
  float4 z = philox_normal4((unsigned long long)q, offset, seed);
  float zz[4] = {z.x, z.y, z.z, z.w};
  for (int k = 0; k < 4; ++k)
    if ((q << 2) + k < n) out[(q << 2) + k] = zz[k];
}

}  // namespace advx
