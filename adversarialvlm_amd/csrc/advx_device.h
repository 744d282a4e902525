// advx_device.h - device-side structs and helpers shared by the kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "advx_taps.h"

namespace advx {

constexpr int kBlock = 256;       // 4 wave64 per workgroup
constexpr int kWave = 64;
constexpr int kStatSlots = 6;     // per-block partial sums (double)

// ----------------------------------------------------------------------------- tables
struct DevTaps {
  int n;
  int stride;
  const int* start;
  const int* count;
  const float* w;
};

// one separable resize into a padded (and optionally normalised) canvas
struct DStage {
  int mode;
  int src_h, src_w;
  int res_h, res_w;
  int can_h, can_w;
  int off_y, off_x;
  float pad_value;
  int normalise;
  int inner_axis_h;
  float mean[3];
  float stdv[3];
  DevTaps th, tw;    // per output row / column: taps into the source
  DevTaps tth, ttw;  // per source row / column: taps into the output (backward gather)
};

#define ADVX_EMIT_PLAIN 0
#define ADVX_EMIT_TILES 1
#define ADVX_EMIT_QWEN 2

// how a canvas is laid out inside one sample's pixel_values (flat index space)
struct DEmit {
  int kind;
  int stage;            // which canvas
  long long out_begin;  // first flat index written by this emit
  long long out_count;  // number of flat indices
  int can_h, can_w;
  int tile, tiles_w;                   // TILES
  int patch, merge, temporal, grid_w;  // QWEN
};

struct DPlan {
  int n_emit;
  DEmit e[2];
  long long out_numel;
  int n_stage;
  long long canvas_off[2];  // float offsets into the workspace
  long long dgrad_off[2];   // gradient buffers of canvases that feed a later stage (-1: none)
  // batch-reduced gradient of the pixel_values, stored in CANVAS order per stage: [copies][3][can_h][can_w]
  // (copies = temporal duplicates of QWEN, else 1).  k_batch_reduce pays the inverse layout map once per
  // column it sums; the transposed resizes then gather from plain images instead of walking the layout map
  // per tap (that index arithmetic was most of their instructions).  -1: the stage's canvas is not emitted.
  long long gcan_off[2];
  int gcan_copies[2];
};

// ------------------------------------------------------------------ layout index maps
// All per-sample index math is 32-bit unsigned (advx_plan_create rejects plans whose sample
// has 2^31 or more elements): 64-bit integer division costs several times more on the VALU.
// flat output index (first temporal copy) of canvas element (c,y,x)
__device__ __host__ inline long long emit_index(const DEmit& e, int c, int y, int x) {
  const unsigned base = (unsigned)e.out_begin;
  if (e.kind == ADVX_EMIT_PLAIN) {
    return (long long)(base + ((unsigned)c * (unsigned)e.can_h + (unsigned)y) * (unsigned)e.can_w + (unsigned)x);
  }
  if (e.kind == ADVX_EMIT_TILES) {
    unsigned T = (unsigned)e.tile;
    unsigned tyi = (unsigned)y / T, ty = (unsigned)y - tyi * T;
    unsigned txi = (unsigned)x / T, tx = (unsigned)x - txi * T;
    unsigned t = tyi * (unsigned)e.tiles_w + txi;
    return (long long)(base + ((t * 3u + (unsigned)c) * T + ty) * T + tx);
  }
  // QWEN: row = ((by*(grid_w/merge)+bx)*merge+mh)*merge+mw ; col = ((c*T+t)*P+ph)*P+pw
  unsigned P = (unsigned)e.patch, M = (unsigned)e.merge;
  unsigned gy = (unsigned)y / P, ph = (unsigned)y - gy * P;
  unsigned gx = (unsigned)x / P, pw = (unsigned)x - gx * P;
  unsigned by = gy / M, mh = gy - by * M;
  unsigned bx = gx / M, mw = gx - bx * M;
  unsigned row = ((by * ((unsigned)e.grid_w / M) + bx) * M + mh) * M + mw;
  unsigned col = (((unsigned)c * (unsigned)e.temporal + 0u) * P + ph) * P + pw;
  unsigned row_len = 3u * (unsigned)e.temporal * P * P;
  return (long long)(base + row * row_len + col);
}

// emit_index is separable: emit_index(e,c,y,x) == out_begin + emit_rowpart(e,c,y) + emit_colpart(e,x).
// A thread that visits a window of rows and columns pays the divisions once per row / column
// instead of once per element (k_stage_bwd).
__device__ __host__ inline unsigned emit_rowpart(const DEmit& e, int c, int y) {
  if (e.kind == ADVX_EMIT_PLAIN) return ((unsigned)c * (unsigned)e.can_h + (unsigned)y) * (unsigned)e.can_w;
  if (e.kind == ADVX_EMIT_TILES) {
    unsigned T = (unsigned)e.tile;
    unsigned tyi = (unsigned)y / T, ty = (unsigned)y - tyi * T;
    return ((tyi * (unsigned)e.tiles_w * 3u + (unsigned)c) * T + ty) * T;
  }
  unsigned P = (unsigned)e.patch, M = (unsigned)e.merge;
  unsigned gy = (unsigned)y / P, ph = (unsigned)y - gy * P;
  unsigned by = gy / M, mh = gy - by * M;
  unsigned row_len = 3u * (unsigned)e.temporal * P * P;
  return ((by * ((unsigned)e.grid_w / M)) * M * M + mh * M) * row_len + ((unsigned)c * (unsigned)e.temporal * P + ph) * P;
}
__device__ __host__ inline unsigned emit_colpart(const DEmit& e, int x) {
  if (e.kind == ADVX_EMIT_PLAIN) return (unsigned)x;
  if (e.kind == ADVX_EMIT_TILES) {
    unsigned T = (unsigned)e.tile;
    unsigned txi = (unsigned)x / T, tx = (unsigned)x - txi * T;
    return txi * 3u * T * T + tx;
  }
  unsigned P = (unsigned)e.patch, M = (unsigned)e.merge;
  unsigned gx = (unsigned)x / P, pw = (unsigned)x - gx * P;
  unsigned bx = gx / M, mw = gx - bx * M;
  unsigned row_len = 3u * (unsigned)e.temporal * P * P;
  return (bx * M * M + mw) * row_len + pw;
}

// stride between the temporal copies of one canvas element (QWEN), number of copies
__device__ __host__ inline int emit_copies(const DEmit& e) { return e.kind == ADVX_EMIT_QWEN ? e.temporal : 1; }
__device__ __host__ inline long long emit_copy_stride(const DEmit& e) { return (long long)e.patch * e.patch; }

// inverse: flat index (inside this emit's range) -> canvas element (and which temporal copy of it)
__device__ __host__ inline void emit_inverse_t(const DEmit& e, long long idx, int& c, int& y, int& x, int& t);
__device__ __host__ inline void emit_inverse(const DEmit& e, long long idx, int& c, int& y, int& x) {
  int t;
  emit_inverse_t(e, idx, c, y, x, t);
}
__device__ __host__ inline void emit_inverse_t(const DEmit& e, long long idx, int& c, int& y, int& x, int& t) {
  t = 0;
  unsigned r = (unsigned)(idx - e.out_begin);
  if (e.kind == ADVX_EMIT_PLAIN) {
    unsigned W = (unsigned)e.can_w, plane = (unsigned)e.can_h * W;
    unsigned cc = r / plane;
    unsigned q = r - cc * plane;
    unsigned yy = q / W;
    c = (int)cc;
    y = (int)yy;
    x = (int)(q - yy * W);
    return;
  }
  if (e.kind == ADVX_EMIT_TILES) {
    unsigned T = (unsigned)e.tile, tt = T * T;
    unsigned ti = r / (3u * tt);
    unsigned q = r - ti * 3u * tt;
    unsigned cc = q / tt;
    q -= cc * tt;
    unsigned ty = q / T, tx = q - ty * T;
    unsigned tyi = ti / (unsigned)e.tiles_w, txi = ti - tyi * (unsigned)e.tiles_w;
    c = (int)cc;
    y = (int)(tyi * T + ty);
    x = (int)(txi * T + tx);
    return;
  }
  unsigned P = (unsigned)e.patch, M = (unsigned)e.merge, pp = P * P;
  unsigned row_len = 3u * (unsigned)e.temporal * pp;
  unsigned row = r / row_len;
  unsigned col = r - row * row_len;
  unsigned ct = col / pp;
  unsigned q = col - ct * pp;
  c = (int)(ct / (unsigned)e.temporal);
  t = (int)(ct - (unsigned)c * (unsigned)e.temporal);
  unsigned ph = q / P, pw = q - ph * P;
  unsigned mw = row % M;
  unsigned r2 = row / M;
  unsigned mh = r2 % M;
  unsigned blk = r2 / M;
  unsigned bw = (unsigned)e.grid_w / M;
  unsigned by = blk / bw, bx = blk - by * bw;
  y = (int)((by * M + mh) * P + ph);
  x = (int)((bx * M + mw) * P + pw);
}

// where the batch-reduced gradient of flat output index idx goes: float offset into the plan workspace
// (canvas order of the emit's stage), or -1 for the constant padding no emit covers
__device__ __host__ inline long long gcan_dest(const DPlan& pl, long long idx) {
  for (int k = 0; k < pl.n_emit; ++k) {
    const DEmit& e = pl.e[k];
    if (idx >= e.out_begin && idx < e.out_begin + e.out_count) {
      if (e.kind == ADVX_EMIT_PLAIN) return pl.gcan_off[e.stage] + (idx - e.out_begin);   // canvas order IS flat order
      int c, y, x, t;
      emit_inverse_t(e, idx, c, y, x, t);
      return pl.gcan_off[e.stage] + (((long long)t * 3 + c) * e.can_h + y) * e.can_w + x;
    }
  }
  return -1;
}

// ------------------------------------------------------------------------- reductions
// Wave-wide sum by DPP moves (VALU, no LDS round trips: a __shfl of a double is two
// ds_bpermute per step and the dependent chain of six steps dominated every per-block
// reduction).  The TOTAL ends up in the LAST lane; the order of the additions is fixed.
//   row_shr:1..3 -> every lane holds itself + its 3 predecessors of the 16-lane row,
//   row_shr:4 / row_shr:8 (bank-masked) -> lane 15 of each row holds the row sum,
//   row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3) -> lane 63 holds the wave sum.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ inline double dpp_or_zero(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum_to_last_lane(double v) {
  double s = v + dpp_or_zero<0x111, 0xf, 0xf>(v);
  s += dpp_or_zero<0x112, 0xf, 0xf>(v);
  s += dpp_or_zero<0x113, 0xf, 0xf>(v);
  s += dpp_or_zero<0x114, 0xf, 0xe>(s);
  s += dpp_or_zero<0x118, 0xf, 0xc>(s);
  s += dpp_or_zero<0x142, 0xa, 0xf>(s);
  s += dpp_or_zero<0x143, 0xc, 0xf>(s);
  return s;
}

// Block-wide sum of NS doubles per thread -> dst[0..NS) (written by one lane; visible to the block after the call).
// Order (fixed, so the totals are bitwise reproducible): per lane position the waves 0..nw-1 in turn, then the DPP tree
// over the 64 positions.  Rounds 1-3 ran the tree in EVERY wave and added the four wave totals last: NS x 21 VALU
// instructions in each wave of every workgroup of the image-sized launches, which are bound by the instructions their SIMDs
// have to issue (k_plan_tail: ~150 of ~700 per thread).  Now the waves above 0 store their NS values and are done; wave 0
// reads them, adds and runs the NS trees: (NS x 27 + 3 x NS) / 4 per wave on average.  (Blocks of 64..MAXT threads, a
// multiple of 64.)
template <int NS, int MAXT = kBlock>
__device__ inline void block_sum_store(const double (&acc)[NS], double* dst) {
  __shared__ double red[NS][MAXT];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave, nw = (int)(blockDim.x / kWave);
  if (wid > 0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) red[k][threadIdx.x] = acc[k];
  }
  __syncthreads();
  if (wid == 0) {
    // every value of the other waves is read before the first addition (one LDS round trip, not one per wave and slot)
    constexpr int MW = MAXT / kWave;
    if (MW <= 4) {
      double o[NS][MW - 1];
#pragma unroll
      for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int w = 1; w < MW; ++w) o[k][w - 1] = (w < nw) ? red[k][w * kWave + lane] : 0.0;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        double v = acc[k];
#pragma unroll
        for (int w = 1; w < MW; ++w)
          if (w < nw) v += o[k][w - 1];
        v = wave_sum_to_last_lane(v);
        if (lane == kWave - 1) dst[k] = v;
      }
    } else {
      // eight waves: one slot's values at a time (all slots at once would be 14 (MW - 1) NS registers)
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        double o[MW - 1];
#pragma unroll
        for (int w = 1; w < MW; ++w) o[w - 1] = (w < nw) ? red[k][w * kWave + lane] : 0.0;
        double v = acc[k];
#pragma unroll
        for (int w = 1; w < MW; ++w)
          if (w < nw) v += o[w - 1];
        v = wave_sum_to_last_lane(v);
        if (lane == kWave - 1) dst[k] = v;
      }
    }
  }
  __syncthreads();  // dst may be shared memory; `red` may be reused by a following call
}

// Two block sums in one pass (the statistics partials and the ||g|| partial of the same workgroup): one LDS exchange and
// one pair of barriers instead of two.  Per slot the additions are block_sum_store's, in its order.
template <int NA, int NB>
__device__ inline void block_sum_store2(const double (&a)[NA], double* dst_a, const double (&b)[NB], double* dst_b) {
  constexpr int NS = NA + NB, MW = kBlock / kWave;
  __shared__ double red[NS][kBlock];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave, nw = (int)(blockDim.x / kWave);
  double acc[NS];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = a[k];
#pragma unroll
  for (int k = 0; k < NB; ++k) acc[NA + k] = b[k];
  if (wid > 0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) red[k][threadIdx.x] = acc[k];
  }
  __syncthreads();
  if (wid == 0) {
    double o[NS][MW - 1];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
      for (int w = 1; w < MW; ++w) o[k][w - 1] = (w < nw) ? red[k][w * kWave + lane] : 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      double v = acc[k];
#pragma unroll
      for (int w = 1; w < MW; ++w)
        if (w < nw) v += o[k][w - 1];
      v = wave_sum_to_last_lane(v);
      if (lane == kWave - 1) {
        if (k < NA) dst_a[k] = v;
        else dst_b[k - NA] = v;
      }
    }
  }
  __syncthreads();
}

// ----------------------------------------------------------------------- Philox noise
// Philox4x32 (Salmon et al., SC'11) with the standard 10 rounds
#ifndef ADVX_PHILOX_ROUNDS
#define ADVX_PHILOX_ROUNDS 10
#endif
__device__ inline uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ADVX_PHILOX_ROUNDS; ++r) {
    // one 32x32->64 multiply (v_mad_u64_u32) per word instead of a mul_hi + mul_lo pair
    unsigned long long p0 = (unsigned long long)0xD2511F53u * (unsigned long long)c.x;
    unsigned long long p1 = (unsigned long long)0xCD9E8D57u * (unsigned long long)c.z;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    // three-input xor in one VALU op (v_bitop3_b32, truth table 0x96; new on gfx950)
    c = make_uint4(__builtin_amdgcn_bitop3_b32(hi1, c.y, k0, 0x96), lo1, __builtin_amdgcn_bitop3_b32(hi0, c.w, k1, 0x96),
                   lo0);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
// four N(0,1) draws from one Philox block: Box-Muller on uniforms built with one convert and
// one fma each (u = r * 2^-32 [+ 2^-33 to keep the logarithm's argument in (0,1]]); raw
// v_log / v_sqrt / v_sin / v_cos (the angle is in revolutions, v_log is log2).
__device__ inline float4 box_muller4(uint4 r);
__device__ inline float4 philox_normal4(unsigned long long gidx, unsigned long long offset,
                                        unsigned long long seed) {
  uint4 c = make_uint4((uint32_t)gidx, (uint32_t)(gidx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32));
  return box_muller4(philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32)));
}
// batched streams (k_emit, k_fused_fwd): counter = (quad q of the sample, batch row b, offset).  A thread
// keeps q and walks b, so only the second counter word changes inside its loop: the products of
// round 1 and one product each of rounds 2 and 3 do not depend on b and are hoisted by the
// compiler (4 of the 20 v_mad_u64_u32 per block, against 1 with a flat b*n4+q counter).
__device__ inline float4 philox_normal4_qb(uint32_t q, uint32_t b, unsigned long long offset,
                                           unsigned long long seed) {
  uint4 c = make_uint4(q, b, (uint32_t)offset, (uint32_t)(offset >> 32));
  return box_muller4(philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32)));
}
__device__ inline float4 box_muller4(uint4 r) {
  const float k32 = 2.3283064365386963e-10f;   // 2^-32
  const float k33 = 1.1641532182693481e-10f;   // 2^-33
  float u1 = __builtin_fmaf((float)r.x, k32, k33);  // (0, 1]
  float u2 = (float)r.y * k32;                      // [0, 1]  (1.0 is the same angle as 0)
  float u3 = __builtin_fmaf((float)r.z, k32, k33);
  float u4 = (float)r.w * k32;
  // -2 ln u = -2 ln2 log2 u
  float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u3));
  float4 z;
  z.x = ra * __builtin_amdgcn_cosf(u2);
  z.y = ra * __builtin_amdgcn_sinf(u2);
  z.z = rb * __builtin_amdgcn_cosf(u4);
  z.w = rb * __builtin_amdgcn_sinf(u4);
  return z;
}

// per-pixel addressing used by the one-launch step (lane = pixel): one Philox block gives the
// noise of four consecutive batch rows (4*bq .. 4*bq+3) of pixel i
__device__ inline float4 philox_normal4_pixel(unsigned long long i, unsigned bq, unsigned long long offset,
                                              unsigned long long seed) {
  // distinct from philox_normal4's counter space through the key (seed high word is perturbed)
  return philox_normal4(i, (offset << 20) | (unsigned long long)(bq & 0xFFFFFu), seed ^ 0x9E3779B97F4A7C15ULL);
}

// XCD-aware block map of the image-sized (column chunk, row, layer) grids (ADVX_TUNE_IMG_XCD, round 4).  Dealt round-robin,
// the workgroups of every row land on all 8 XCDs and each XCD's L2 fetches the WHOLE source of a resize (8 x 3.1 MB at
// 512 x 512 where 3.1 MB are needed).  Launched 1-D with the grid padded to a multiple of 8, physical block b runs on XCD
// b % 8 (guide: workgroup dispatch).  The logical blocks are cut into GROUPS of `group` consecutive ones (host: a few rows
// of column chunks) that are dealt to the XCDs in turn: XCD k's i-th block is logical block ((i / group) * 8 + k) * group +
// i % group.  One XCD then fetches the source rows of its row groups (+ the window's halo) only - and, unlike whole bands of
// rows per XCD (measured: a crop window that leaves the first and last rows of the image empty, or a multi-plan grid sized
// for its largest canvas, left some XCDs without work and cost 1-4 us), every XCD gets the same share of every region.
// logical block of physical block b (XCD b % 8's (b / 8)-th block) for groups of `group` logical blocks dealt to the XCDs in turn
__host__ __device__ inline unsigned xcd_group_logical(unsigned b, unsigned group) {
  const unsigned k = b & 7u, i = b >> 3;
  const unsigned q = i / group;
  return (q * 8u + k) * group + (i - q * group);
}
struct BlockXYZ { unsigned x, y, z; };
// the logical grid of a launch (host: img_grid).  extra: blocks appended behind the gx * gy * gz of the grid (riders: table
// builders); they come back as z == gz, x = index
struct ImgGrid { int banded; unsigned gx, gy, gz, extra, group; };
__device__ inline bool xcd_band_block(const ImgGrid& ig, BlockXYZ& b) {
  if (!ig.banded) {
    b.x = blockIdx.x; b.y = blockIdx.y; b.z = blockIdx.z;
    return true;
  }
  const unsigned gx = ig.gx, gy = ig.gy, gz = ig.gz;
  const unsigned L = xcd_group_logical(blockIdx.x, ig.group);
  const unsigned body = gx * gy * gz;
  if (L >= body) {
    b.x = L - body; b.y = 0; b.z = gz;
    return L - body < ig.extra;
  }
  const unsigned t = L / gx;
  b.x = L - t * gx;
  b.z = t / gy;
  b.y = t - b.z * gy;
  return true;
}

}  // namespace advx
