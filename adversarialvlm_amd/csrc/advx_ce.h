// advx_ce.h - suffix-only cross entropy: the loss of
// attack_model.py:324-328 / llavaprocessor.py:73-78 on the logits of the target positions only.
//
// The host asks the VLM for the last K = suffix_len + 1 positions (logits_to_keep), so the
// [B, S, V] logits tensor (2.6 GB fp16 at B=64, S=640, V=32000) and its fp32 copy never exist;
// what is left, [B, K, V] with the first T = suffix_len - shift positions of each row block
// supervised, is small (a few tens of MB), and its log-softmax + NLL forward and backward are
// one read each: one workgroup per row, fp32 arithmetic, fixed-order reductions.
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

// one workgroup per supervised row r = b*T + t of logits[b, t, :] (element strides sb, st)
template <int IO>
__global__ void __launch_bounds__(kBlock) k_ce_fwd(const void* __restrict__ logits, long long sb, long long st, int T,
                                                   const long long* __restrict__ targets, long long vocab,
                                                   float* __restrict__ row_loss, float* __restrict__ row_lse) {
  const long long r = blockIdx.x;
  const long long b = r / T, t = r - b * T;
  const void* row = ce_row<IO>(logits, b * sb + t * st);
  float m = -INFINITY;
  for (long long v = threadIdx.x; v < vocab; v += blockDim.x) m = fmaxf(m, ce_load<IO>(row, v));
  m = block_max(m);
  float s = 0.0f;
  for (long long v = threadIdx.x; v < vocab; v += blockDim.x) s += expf(ce_load<IO>(row, v) - m);
  const double total = block_sum_bcast((double)s);
  if (threadIdx.x == 0) {
    const float lse = m + logf((float)total);
    const long long tg = targets[r];
    row_lse[r] = lse;
    row_loss[r] = (tg >= 0 && tg < vocab) ? (lse - ce_load<IO>(row, tg)) : 0.0f;
  }
}

// mean over the supervised rows (targets outside [0, vocab) are ignored, like ignore_index)
__global__ void __launch_bounds__(kBlock) k_ce_mean(const float* __restrict__ row_loss, const long long* __restrict__ targets,
                                                    long long rows, long long vocab, float* __restrict__ out /* [loss, n_valid] */) {
  double acc[2] = {0.0, 0.0};
  for (long long r = threadIdx.x; r < rows; r += blockDim.x) {
    const long long tg = targets[r];
    if (tg >= 0 && tg < vocab) {
      acc[0] += (double)row_loss[r];
      acc[1] += 1.0;
    }
  }
  __shared__ double tot[2];
  block_sum_store<2>(acc, tot);
  if (threadIdx.x == 0) {
    out[0] = (tot[1] > 0.0) ? (float)(tot[0] / tot[1]) : 0.0f;
    out[1] = (float)tot[1];
  }
}

// grad[b, j, :] for all K kept positions: (softmax - onehot) * upstream / n_valid on the T
// supervised ones, zeros on the rest (same strides as logits; may alias logits)
template <int IO>
__global__ void __launch_bounds__(kBlock) k_ce_bwd(const void* __restrict__ logits, long long sb, long long st, int T, int K,
                                                   const long long* __restrict__ targets, long long vocab,
                                                   const float* __restrict__ row_lse, const float* __restrict__ mean_and_n,
                                                   const float* __restrict__ upstream, void* __restrict__ grad) {
  const long long rk = blockIdx.x;
  const long long b = rk / K, j = rk - b * K;
  void* grow = const_cast<void*>(ce_row<IO>(grad, b * sb + j * st));
  if (j >= T) {
    for (long long v = threadIdx.x; v < vocab; v += blockDim.x) ce_store<IO>(grow, v, 0.0f);
    return;
  }
  const long long r = b * T + j;
  const void* row = ce_row<IO>(logits, b * sb + j * st);
  const long long tg = targets[r];
  const bool valid = (tg >= 0 && tg < vocab);
  const float nv = mean_and_n[1];
  const float scale = (valid && nv > 0.0f) ? upstream[0] / nv : 0.0f;
  const float lse = row_lse[r];
  for (long long v = threadIdx.x; v < vocab; v += blockDim.x) {
    float pr = expf(ce_load<IO>(row, v) - lse);
    if (v == tg) pr -= 1.0f;
    ce_store<IO>(grow, v, pr * scale);
  }
}

}  // namespace advx
