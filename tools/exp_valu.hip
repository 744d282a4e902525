// exp_valu.hip - development experiment: issue cost (cycles per wave64 instruction per SIMD) of the
// VALU instructions the in-kernel noise generator is made of, on gfx950.  Each kernel runs long
// chains of one instruction on 8 independent registers per lane with every SIMD fully occupied.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAIN(NAME, TYPE, INIT, BODY)                                                                   \
  __global__ void __launch_bounds__(256) NAME(TYPE* out, int iters, TYPE seed) {                       \
    TYPE r[8];                                                                                          \
    for (int k = 0; k < 8; ++k) r[k] = INIT;                                                            \
    for (int i = 0; i < iters; ++i) {                                                                   \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) { TYPE x = r[k]; BODY; r[k] = x; }                  \
    }                                                                                                   \
    TYPE acc = r[0];                                                                                    \
    for (int k = 1; k < 8; ++k) acc = acc + r[k];                                                       \
    if (acc == (TYPE)123456789) out[0] = acc;                                                           \
  }

CHAIN(k_mad64, uint32_t, seed + threadIdx.x + k,
      { unsigned long long p = (unsigned long long)x * 0xD2511F53u; x = (uint32_t)(p >> 32) ^ (uint32_t)p; })
CHAIN(k_mulhi, uint32_t, seed + threadIdx.x + k, { x = __umulhi(x, 0xD2511F53u) + 1u; })
CHAIN(k_mullo, uint32_t, seed + threadIdx.x + k, { x = x * 0xD2511F53u + 1u; })
CHAIN(k_mul24, uint32_t, seed + threadIdx.x + k, { x = __umul24(x, 0x511F53u) + 1u; })
CHAIN(k_xor, uint32_t, seed + threadIdx.x + k, { x = x ^ (x >> 3); })
CHAIN(k_bitop3, uint32_t, seed + threadIdx.x + k, { x = __builtin_amdgcn_bitop3_b32(x, x >> 3, seed, 0x96); })
CHAIN(k_add, uint32_t, seed + threadIdx.x + k, { x = x + seed; })
CHAIN(k_fma, float, (float)(seed + threadIdx.x + k), { x = __builtin_fmaf(x, 0.999f, 0.5f); })
CHAIN(k_sin, float, (float)(seed + threadIdx.x + k) * 1e-3f, { x = __builtin_amdgcn_sinf(x); })
CHAIN(k_log, float, (float)(seed + threadIdx.x + k) + 2.0f, { x = __builtin_amdgcn_logf(x) + 3.0f; })
CHAIN(k_sqrt, float, (float)(seed + threadIdx.x + k) + 2.0f, { x = __builtin_amdgcn_sqrtf(x) + 3.0f; })
CHAIN(k_rcp, float, (float)(seed + threadIdx.x + k) + 2.0f, { x = __builtin_amdgcn_rcpf(x) + 3.0f; })
CHAIN(k_cvt, uint32_t, seed + threadIdx.x + k, { x = (uint32_t)__builtin_bit_cast(uint32_t, (float)x); })

template <typename K, typename T>
void run(const char* name, K kern, T seed, int ops_per_body, const char* note) {
  T* out;
  (void)hipMalloc(&out, 64);
  const int iters = 2000, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, seed);
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, seed);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  // wave-instructions per SIMD = 8 waves * iters * 8 chains * ops_per_body
  double per_simd = 8.0 * iters * 8 * ops_per_body;
  printf("  %-10s %8.1f us  -> %.2f ns per wave-instruction-group per SIMD (%s)\n", name, ms * 1e3, ms * 1e6 / per_simd, note);
  (void)hipFree(out);
}

int main() {
  printf("cost per body (divide by the clock period, ~0.42 ns at 2.4 GHz, for cycles)\n");
  run("add", k_add, 7u, 1, "1 v_add");
  run("xor", k_xor, 7u, 1, "v_lshrrev + v_xor");
  run("bitop3", k_bitop3, 7u, 1, "v_lshrrev + v_bitop3");
  run("mad64", k_mad64, 7u, 1, "v_mad_u64_u32 + v_xor");
  run("mulhi", k_mulhi, 7u, 1, "v_mul_hi_u32 + v_add");
  run("mullo", k_mullo, 7u, 1, "v_mul_lo_u32 + v_add (or v_mad_u32_u24?)");
  run("mul24", k_mul24, 7u, 1, "v_mul_u32_u24 + v_add");
  run("fma", k_fma, 7.0f, 1, "v_fma_f32");
  run("sin", k_sin, 7.0f, 1, "v_sin_f32");
  run("log", k_log, 7.0f, 1, "v_log_f32 + v_add");
  run("sqrt", k_sqrt, 7.0f, 1, "v_sqrt_f32 + v_add");
  run("rcp", k_rcp, 7.0f, 1, "v_rcp_f32 + v_add");
  run("cvt", k_cvt, 7u, 1, "v_cvt_f32_u32");
  return 0;
}
