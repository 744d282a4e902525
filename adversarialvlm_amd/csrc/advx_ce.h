// advx_ce.h - suffix-only cross entropy: the loss of
// attack_model.py:324-328 / llavaprocessor.py:73-78 on the logits of the target positions only.
//
// The host asks the VLM for the last K = suffix_len + 1 positions (logits_to_keep), so the
// [B, S, V] logits tensor (2.6 GB fp16 at B=64, S=640, V=32000) and its fp32 copy never exist;
// what is left, [B, K, V] with the first T = suffix_len - shift positions of each row block
// supervised, is small (tens to hundreds of MB), and its log-softmax + NLL forward and backward are
// one read each: 16-byte accesses, fp32 arithmetic, fixed-order reductions.
#pragma once
#include "advx_device.h"

namespace advx {

__device__ inline float block_max(float v) {
  __shared__ float red_max[16];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  if (lane == 0) red_max[wid] = v;
  __syncthreads();
  float m = red_max[0];
  for (int w = 1; w < (int)(blockDim.x / kWave); ++w) m = fmaxf(m, red_max[w]);
  __syncthreads();
  return m;
}

__device__ inline double block_sum_bcast(double v) {
  __shared__ double tot;
  double acc[1] = {v};
  block_sum_store<1>(acc, &tot);   // ends with a barrier
  return tot;
}

// element v of a row in the boundary dtype (0 f32, 1 f16, 2 bf16)
template <int IO>
__device__ inline float ce_load(const void* row, long long v) {
  if (IO == 0) return reinterpret_cast<const float*>(row)[v];
  if (IO == 1) return (float)reinterpret_cast<const _Float16*>(row)[v];
  return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(row)[v] << 16);
}
template <int IO>
__device__ inline void ce_store(void* row, long long v, float x) {
  if (IO == 0) reinterpret_cast<float*>(row)[v] = x;
  else if (IO == 1) reinterpret_cast<_Float16*>(row)[v] = (_Float16)x;
  else reinterpret_cast<__bf16*>(row)[v] = (__bf16)x;
}
template <int IO>
__device__ inline const void* ce_row(const void* base, long long elem_off) {
  return reinterpret_cast<const char*>(base) + elem_off * (IO == 0 ? 4 : 2);
}

// exp(d), d <= 0 (an element minus its row's maximum or log-sum-exp).  Half-precision logits: v_exp_f32 on d * log2(e) - two
// instructions where expf's range reduction and overflow checks take thirteen, which made both kernels VALU-bound (the backward
// moved 1.2 G elements per ms whatever their width).  Relative error <= 1 ulp + |d| 2^-24 (< 3e-6 down to exp(-40)), far
// inside what a half-precision logit carries; float32 logits keep expf.
template <int IO>
__device__ inline float ce_exp(float d) {
  return IO == 0 ? expf(d) : __builtin_amdgcn_exp2f(d * 1.44269504088896340736f);
}

// ---- 16-byte accesses: 4 floats / 8 halfs per load.  A row starts at an element-aligned address; `head` elements sit in
// front of its first 16-byte boundary and fewer than VEC behind its last whole vector - those go one per thread.
template <int IO>
struct CeVec {
  static constexpr int VEC = (IO == 0) ? 4 : 8;
  static constexpr int ES = (IO == 0) ? 4 : 2;
};
struct CeRowSplit {
  int head;          // scalar elements [0, head)
  long long nvec;    // whole vectors behind them
  long long tail0;   // scalar elements [tail0, vocab)
};
template <int IO>
__device__ inline CeRowSplit ce_split(const void* row, long long vocab) {
  CeRowSplit r;
  const unsigned mis = (unsigned)(reinterpret_cast<unsigned long long>(row) & 15u);
  const long long h = ((16u - mis) & 15u) / CeVec<IO>::ES;
  r.head = (int)(h < vocab ? h : vocab);
  r.nvec = (vocab - r.head) / CeVec<IO>::VEC;
  r.tail0 = r.head + r.nvec * CeVec<IO>::VEC;
  return r;
}
template <int IO>
__device__ inline void ce_unpack(const uint4& raw, float (&x)[CeVec<IO>::VEC]) {
  const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
  if (IO == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = __builtin_bit_cast(float, w[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (IO == 1) {
        x[2 * k] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[k] & 0xffffu));
        x[2 * k + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[k] >> 16));
      } else {
        x[2 * k] = __builtin_bit_cast(float, w[k] << 16);
        x[2 * k + 1] = __builtin_bit_cast(float, w[k] & 0xffff0000u);
      }
    }
  }
}
template <int IO>
__device__ inline uint4 ce_pack(const float (&x)[CeVec<IO>::VEC]) {
  unsigned w[4];
  if (IO == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = __builtin_bit_cast(unsigned, x[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned lo, hi;
      if (IO == 1) {
        lo = __builtin_bit_cast(unsigned short, (_Float16)x[2 * k]);
        hi = __builtin_bit_cast(unsigned short, (_Float16)x[2 * k + 1]);
      } else {
        lo = __builtin_bit_cast(unsigned short, (__bf16)x[2 * k]);
        hi = __builtin_bit_cast(unsigned short, (__bf16)x[2 * k + 1]);
      }
      w[k] = lo | (hi << 16);
    }
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}
// the row's elements outside its whole vectors (fewer than 2 VEC of them): element index of scalar slot k, or -1
__device__ inline long long ce_scalar_index(const CeRowSplit& sp, long long vocab, int k) {
  if (k < sp.head) return k;
  const long long e = sp.tail0 + (k - sp.head);
  return e < vocab ? e : -1;
}

// Forward, launch 1 of 2 - grid (supervised row r = b*T + t, chunk of the row): a workgroup reads kCeFwdVecs vectors per
// thread ONCE, with 16-byte loads that are all in flight together, and runs both passes - max, sum of exp(x - max) - on the
// registers; it leaves (max, sum) of its chunk.  A row of 32 064 halfs is two chunks (1024 workgroups at 64 prompts x 8
// target positions), Qwen2-VL's 152 064 ten.  The row's few elements outside its whole vectors go with chunk 0.
// (Rounds 1-3: one workgroup per row, two passes over memory with 2-byte loads - 0.28-0.37 TB/s, tools/ce_bench.py; one
// workgroup per row holding the whole row in registers: 2.1 TB/s - each workgroup's latency, not the memory, was the bound.)
#ifndef ADVX_CE_FWD_VECS
#define ADVX_CE_FWD_VECS 8      // 16-byte vectors per thread held in registers by k_ce_fwd (4 and 16 measured: profiles/r04/w_ce_bench.log)
#endif
constexpr int kCeFwdVecs = ADVX_CE_FWD_VECS;
template <int IO>
__global__ void __launch_bounds__(kBlock) k_ce_fwd(const void* __restrict__ logits, long long sb, long long st, int T,
                                                   long long vocab, float2* __restrict__ chunk_max_sum) {
  constexpr int VEC = CeVec<IO>::VEC, NV = kCeFwdVecs;
  const long long r = blockIdx.x;
  const long long b = r / T, t = r - b * T;
  const void* row = ce_row<IO>(logits, b * sb + t * st);
  const CeRowSplit sp = ce_split<IO>(row, vocab);
  const uint4* body = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(row) + (size_t)sp.head * CeVec<IO>::ES);
  const long long v0 = (long long)blockIdx.y * NV * blockDim.x + threadIdx.x;
  uint4 raw[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const long long v = v0 + (long long)k * blockDim.x;
    raw[k] = make_uint4(0u, 0u, 0u, 0u);
    if (sp.nvec > 0) raw[k] = body[v < sp.nvec ? v : sp.nvec - 1];      // a clamped address where the thread has none
  }
  const long long se = (blockIdx.y == 0 && threadIdx.x < 2 * VEC) ? ce_scalar_index(sp, vocab, (int)threadIdx.x) : -1;
  const float xs = se >= 0 ? ce_load<IO>(row, se) : -INFINITY;
  float m = xs;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (v0 + (long long)k * blockDim.x < sp.nvec) {
      float x[VEC];
      ce_unpack<IO>(raw[k], x);
#pragma unroll
      for (int i = 0; i < VEC; ++i) m = fmaxf(m, x[i]);
    }
  }
  m = block_max(m);
  // the second pass unpacks the raw words again: keeping the first pass's floats alive would double the registers (the
  // compiler does, unless the words are made opaque here)
#pragma unroll
  for (int k = 0; k < NV; ++k) asm volatile("" : "+v"(raw[k].x), "+v"(raw[k].y), "+v"(raw[k].z), "+v"(raw[k].w));
  float s = se >= 0 ? ce_exp<IO>(xs - m) : 0.0f;
  if (m > -INFINITY) {             // a chunk of -inf only (or an empty one) contributes nothing
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (v0 + (long long)k * blockDim.x < sp.nvec) {
        float x[VEC];
        ce_unpack<IO>(raw[k], x);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s += ce_exp<IO>(x[i] - m);
      }
    }
  } else {
    s = 0.0f;
  }
  __shared__ double tot;
  double acc[1] = {(double)s};
  block_sum_store<1>(acc, &tot);
  if (threadIdx.x == 0) chunk_max_sum[r * gridDim.y + blockIdx.y] = make_float2(m, (float)tot);
}

// Forward, launch 2 of 2 - one workgroup of 64 .. 1024 threads: per row the chunks' (max, sum) -> log-sum-exp and the row's loss, then the mean
// over the supervised rows (targets outside [0, vocab) are ignored, like ignore_index)
constexpr int kCeFinishThreads = 1024;       // one row per thread up to 1024 rows: a row is three dependent loads
template <int IO>
__global__ void __launch_bounds__(kCeFinishThreads) k_ce_finish(const void* __restrict__ logits, long long sb, long long st, int T,
                                                      const long long* __restrict__ targets, long long rows, long long vocab,
                                                      const float2* __restrict__ chunk_max_sum, int chunks,
                                                      float* __restrict__ row_loss, float* __restrict__ row_lse,
                                                      float* __restrict__ out /* [loss, n_valid] */) {
  double acc[2] = {0.0, 0.0};
  constexpr int U = 8;               // chunks in flight per thread (LLaVA-1.5: 2 per row, Qwen2-VL: 10)
  for (long long r = threadIdx.x; r < rows; r += blockDim.x) {
    const float2* c = chunk_max_sum + r * chunks;
    // everything that does not depend on the chunks first: the target and its logit
    const long long b = r / T, t = r - b * T;
    const long long tg = targets[r];
    const bool valid = tg >= 0 && tg < vocab;
    const float xt = valid ? ce_load<IO>(ce_row<IO>(logits, b * sb + t * st), tg) : 0.0f;
    float m = -INFINITY;
    for (int k0 = 0; k0 < chunks; k0 += U) {
      float cm[U];
#pragma unroll
      for (int u = 0; u < U; ++u) cm[u] = c[min(k0 + u, chunks - 1)].x;
#pragma unroll
      for (int u = 0; u < U; ++u) m = fmaxf(m, cm[u]);
    }
    double total = 0.0;
    for (int k0 = 0; k0 < chunks; k0 += U) {
      float2 cv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) cv[u] = c[min(k0 + u, chunks - 1)];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (k0 + u < chunks && cv[u].x > -INFINITY) total += (double)cv[u].y * (double)expf(cv[u].x - m);
    }
    const float lse = m + logf((float)total);
    const float loss = valid ? (lse - xt) : 0.0f;
    row_lse[r] = lse;
    row_loss[r] = loss;
    if (valid) {
      acc[0] += (double)loss;
      acc[1] += 1.0;
    }
  }
  __shared__ double tot[2];
  block_sum_store<2, kCeFinishThreads>(acc, tot);
  if (threadIdx.x == 0) {
    out[0] = (tot[1] > 0.0) ? (float)(tot[0] / tot[1]) : 0.0f;
    out[1] = (float)tot[1];
  }
}

// grad[b, j, :] for all K kept positions: (softmax - onehot) * upstream / n_valid on the T
// supervised ones, zeros on the rest (same strides as logits; may alias logits).  Grid (kept row, chunk of the row): a
// workgroup takes kCeBwdVecs vectors per thread, all loaded before the first use, 16-byte loads and stores; the row's few
// elements outside its whole vectors go with chunk 0.  (Rounds 1-3: one workgroup per row, 2-byte accesses - 1.2-1.4 TB/s.)
constexpr int kCeBwdVecs = 4;
template <int IO>
__global__ void __launch_bounds__(kBlock) k_ce_bwd(const void* __restrict__ logits, long long sb, long long st, int T, int K,
                                                   const long long* __restrict__ targets, long long vocab,
                                                   const float* __restrict__ row_lse, const float* __restrict__ mean_and_n,
                                                   const float* __restrict__ upstream, void* __restrict__ grad) {
  constexpr int VEC = CeVec<IO>::VEC, U = kCeBwdVecs;
  const long long rk = blockIdx.x;
  const long long b = rk / K, j = rk - b * K;
  void* grow = const_cast<void*>(ce_row<IO>(grad, b * sb + j * st));
  const void* row = ce_row<IO>(logits, b * sb + j * st);
  const bool supervised = j < T;
  const long long r = b * T + j;
  const long long tg = supervised ? targets[r] : -1;
  const bool valid = (tg >= 0 && tg < vocab);
  const float nv = mean_and_n[1];
  const float scale = (supervised && valid && nv > 0.0f) ? upstream[0] / nv : 0.0f;
  const float lse = supervised ? row_lse[r] : 0.0f;
  // the gradient row must sit like the logits row relative to 16-byte boundaries (it does: same strides, fresh allocation or
  // the logits themselves); otherwise element by element
  const bool same_phase = ((reinterpret_cast<unsigned long long>(row) ^ reinterpret_cast<unsigned long long>(grow)) & 15u) == 0;
  if (!same_phase) {
    const long long per = (vocab + gridDim.y - 1) / gridDim.y;
    const long long e1 = min(vocab, (long long)(blockIdx.y + 1) * per);
    for (long long v = (long long)blockIdx.y * per + threadIdx.x; v < e1; v += blockDim.x) {
      float pr = 0.0f;
      if (supervised) {
        pr = ce_exp<IO>(ce_load<IO>(row, v) - lse);
        if (v == tg) pr -= 1.0f;
        pr *= scale;
      }
      ce_store<IO>(grow, v, pr);
    }
    return;
  }
  const CeRowSplit sp = ce_split<IO>(row, vocab);
  if (blockIdx.y == 0 && threadIdx.x < 2 * VEC) {
    const long long se = ce_scalar_index(sp, vocab, (int)threadIdx.x);
    if (se >= 0) {
      float pr = 0.0f;
      if (supervised) {
        pr = ce_exp<IO>(ce_load<IO>(row, se) - lse);
        if (se == tg) pr -= 1.0f;
        pr *= scale;
      }
      ce_store<IO>(grow, se, pr);
    }
  }
  const uint4* body = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(row) + (size_t)sp.head * CeVec<IO>::ES);
  uint4* gbody = reinterpret_cast<uint4*>(reinterpret_cast<char*>(grow) + (size_t)sp.head * CeVec<IO>::ES);
  const long long v0 = (long long)blockIdx.y * U * blockDim.x + threadIdx.x;
  if (!supervised) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long v = v0 + (long long)u * blockDim.x;
      if (v < sp.nvec) gbody[v] = make_uint4(0u, 0u, 0u, 0u);       // +0.0 in every format
    }
    return;
  }
  uint4 raw[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long v = v0 + (long long)u * blockDim.x;
    raw[u] = make_uint4(0u, 0u, 0u, 0u);
    if (v0 < sp.nvec) raw[u] = body[v < sp.nvec ? v : sp.nvec - 1];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long v = v0 + (long long)u * blockDim.x;
    if (v < sp.nvec) {
      float x[VEC];
      ce_unpack<IO>(raw[u], x);
      const long long e0 = sp.head + v * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float pr = ce_exp<IO>(x[i] - lse);
        if (e0 + i == tg) pr -= 1.0f;
        x[i] = pr * scale;
      }
      gbody[v] = ce_pack<IO>(x);
    }
  }
}

}  // namespace advx
