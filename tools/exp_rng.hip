// exp_rng.hip - development experiment: cost of in-kernel noise generators beside an
// 86.7 MB streaming store (the shape of k_fused_fwd).  hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ROUNDS, bool MUL64>
__device__ inline uint4 philox(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    if (MUL64) {
      unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
      hi0 = p0 >> 32; lo0 = (uint32_t)p0; hi1 = p1 >> 32; lo1 = (uint32_t)p1;
    } else {
      hi0 = __umulhi(0xD2511F53u, c.x); lo0 = 0xD2511F53u * c.x;
      hi1 = __umulhi(0xCD9E8D57u, c.z); lo1 = 0xCD9E8D57u * c.z;
    }
    c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}
__device__ inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ inline uint32_t xoshiro_next(uint4& s) {  // xoshiro128++
  uint32_t r = rotl(s.x + s.w, 7) + s.x;
  uint32_t t = s.y << 9;
  s.z ^= s.x; s.w ^= s.y; s.y ^= s.z; s.x ^= s.w; s.z ^= t; s.w = rotl(s.w, 11);
  return r;
}
__device__ inline float4 box_muller(uint4 r) {
  const float k24 = 1.0f / 16777216.0f;
  float u1 = (float)((r.x >> 8) + 1u) * k24, u2 = (float)(r.y >> 8) * k24;
  float u3 = (float)((r.z >> 8) + 1u) * k24, u4 = (float)(r.w >> 8) * k24;
  float ra = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  float rb = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u3));
  return make_float4(ra * __builtin_amdgcn_cosf(u2), ra * __builtin_amdgcn_sinf(u2), rb * __builtin_amdgcn_cosf(u4),
                     rb * __builtin_amdgcn_sinf(u4));
}

// MODE 0 none, 1 philox10 hi/lo, 2 philox10 mul64, 3 philox7 mul64, 4 xoshiro128++ (philox-seeded), 5 uniform-only philox10
template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ base, long long n, int batch, int bps, float sigma,
                                         float* __restrict__ out) {
  long long n4 = n >> 2;
  long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n4) return;
  float4 v = *reinterpret_cast<const float4*>(base + (q << 2));
  int b0 = blockIdx.y * bps, b1 = min(batch, b0 + bps);
  uint4 st;
  if (MODE == 4) st = philox<10, true>(make_uint4((uint32_t)q, (uint32_t)(q >> 32), blockIdx.y, 77), 1234, 5678);
  for (int b = b0; b < b1; ++b) {
    float4 o = v;
    if (MODE != 0) {
      uint4 r;
      unsigned long long gi = (unsigned long long)b * n4 + q;
      uint4 c = make_uint4((uint32_t)gi, (uint32_t)(gi >> 32), 7, 0);
      if (MODE == 1) r = philox<10, false>(c, 1234, 5678);
      if (MODE == 2 || MODE == 5) r = philox<10, true>(c, 1234, 5678);
      if (MODE == 3) r = philox<7, true>(c, 1234, 5678);
      if (MODE == 4) { r.x = xoshiro_next(st); r.y = xoshiro_next(st); r.z = xoshiro_next(st); r.w = xoshiro_next(st); }
      float4 z;
      if (MODE == 5) z = make_float4(r.x * 2.3e-10f, r.y * 2.3e-10f, r.z * 2.3e-10f, r.w * 2.3e-10f);
      else z = box_muller(r);
      o = make_float4(v.x + z.x * sigma, v.y + z.y * sigma, v.z + z.z * sigma, v.w + z.w * sigma);
    }
    *reinterpret_cast<float4*>(out + (size_t)b * n + (q << 2)) = o;
  }
}

template <int MODE>
float run(const float* base, long long n, int batch, int slices, float* out, int iters) {
  int bps = (batch + slices - 1) / slices;
  dim3 grid((unsigned)((n / 4 + 255) / 256), slices);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, base, n, batch, bps, 1e-3f, out);
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, base, n, batch, bps, 1e-3f, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters * 1e3f;
}

int main() {
  const long long n = 3LL * 336 * 336;
  const int batch = 64;
  float *base, *out;
  CHECK(hipMalloc(&base, n * 4));
  CHECK(hipMalloc(&out, n * 4 * batch));
  CHECK(hipMemset(base, 0, n * 4));
  for (int slices : {4, 8, 16, 32, 64}) {
    printf("slices %2d: none %.1f | philox10 hi/lo %.1f | philox10 mul64 %.1f | philox7 %.1f | xoshiro128++ %.1f | philox10 no-boxmuller %.1f  (us per 86.7MB launch)\n",
           slices, run<0>(base, n, batch, slices, out, 200), run<1>(base, n, batch, slices, out, 200),
           run<2>(base, n, batch, slices, out, 200), run<3>(base, n, batch, slices, out, 200),
           run<4>(base, n, batch, slices, out, 200), run<5>(base, n, batch, slices, out, 200));
  }
  return 0;
}
