// exp_merge.hip - development experiment: can ONE kernel overlap the gradient stream (read
// B*P floats) with the noise generation + pixel_values stream (write B*P floats)?
// Variant: wave-independent persistent waves, one pixel per lane (64-pixel groups, 256-byte
// rows), software-pipelined so that the loads of group k+1 are in flight while group k is
// emitted.  Timing only (simplified update arithmetic).  hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline uint4 philox10(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
    c = make_uint4((uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}
__device__ inline float4 normal4(uint4 r) {
  const float k32 = 2.3283064365386963e-10f, k33 = 1.1641532182693481e-10f;
  float u1 = __builtin_fmaf((float)r.x, k32, k33), u2 = (float)r.y * k32;
  float u3 = __builtin_fmaf((float)r.z, k32, k33), u4 = (float)r.w * k32;
  float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u3));
  return make_float4(ra * __builtin_amdgcn_cosf(u2), ra * __builtin_amdgcn_sinf(u2), rb * __builtin_amdgcn_cosf(u4),
                     rb * __builtin_amdgcn_sinf(u4));
}

// batch must be a multiple of 8 here
template <bool RNG, bool PIPE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ g, float* __restrict__ p, long long n, int batch,
                                         float sigma, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long ngroups = (n + 63) >> 6;
  long long grp = wave;
  if (grp >= ngroups) return;
  // prologue: reduce the first group (not overlapped)
  float acc = 0.f;
  {
    long long i = grp * 64 + lane;
    for (int b = 0; b < batch; b += 8) {
      float t[8];
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) t[k2] = g[(size_t)(b + k2) * n + i];
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) acc += t[k2];
    }
  }
  while (grp < ngroups) {
    const long long i = grp * 64 + lane;
    const long long nxt = grp + nwaves;
    const long long in_ = nxt * 64 + lane;
    const bool has_next = PIPE && (nxt < ngroups);
    // update (stand-in for /std, tanh', AdamW, prepare)
    float pp = p[i];
    pp = pp - 0.01f * tanhf(acc) * (1.0f - tanhf(pp) * tanhf(pp));
    p[i] = pp;
    float v = tanhf(pp);
    float acc_next = 0.f;
    // emit group `grp` while the loads of group `nxt` are in flight (8 per lane)
    for (int b = 0; b < batch; b += 8) {
      float t[8];
      if (has_next) {
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) t[k2] = g[(size_t)(b + k2) * n + in_];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 z = make_float4(0, 0, 0, 0);
        if (RNG) z = normal4(philox10(make_uint4((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)(b + 4 * h), 7), 1234, 5678));
        __builtin_nontemporal_store(v + sigma * z.x, out + (size_t)(b + 4 * h + 0) * n + i);
        __builtin_nontemporal_store(v + sigma * z.y, out + (size_t)(b + 4 * h + 1) * n + i);
        __builtin_nontemporal_store(v + sigma * z.z, out + (size_t)(b + 4 * h + 2) * n + i);
        __builtin_nontemporal_store(v + sigma * z.w, out + (size_t)(b + 4 * h + 3) * n + i);
      }
      if (has_next) {
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) acc_next += t[k2];
      }
    }
    if (!PIPE && nxt < ngroups) {
      for (int b = 0; b < batch; b += 8) {
        float t[8];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) t[k2] = g[(size_t)(b + k2) * n + in_];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) acc_next += t[k2];
      }
    }
    acc = acc_next;
    grp = nxt;
  }
}

template <bool RNG, bool PIPE>
float run(const float* g, float* p, long long n, int batch, float* out, int blocks, int iters) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<RNG, PIPE>), dim3(blocks), dim3(256), 0, 0, g, p, n, batch, 1e-3f, out);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<RNG, PIPE>), dim3(blocks), dim3(256), 0, 0, g, p, n, batch, 1e-3f, out);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / iters * 1e3f;
}

int main() {
  const long long n = 3LL * 336 * 336;
  const int batch = 64;
  float *g, *p, *out;
  if (hipMalloc(&g, n * 4 * batch) != hipSuccess || hipMalloc(&p, n * 4) != hipSuccess || hipMalloc(&out, n * 4 * batch) != hipSuccess) return 1;
  (void)hipMemset(g, 0, n * 4 * batch); (void)hipMemset(p, 0, n * 4);
  for (int blocks : {256, 512, 768, 1024, 1323, 2048}) {
    printf("blocks %4d (%5.2f groups/wave): pipelined+rng %.1f | pipelined no-rng %.1f | unpipelined+rng %.1f | unpipelined no-rng %.1f (us per step)\n",
           blocks, 5292.0 / (blocks * 4), run<true, true>(g, p, n, batch, out, blocks, 100), run<false, true>(g, p, n, batch, out, blocks, 100),
           run<true, false>(g, p, n, batch, out, blocks, 100), run<false, false>(g, p, n, batch, out, blocks, 100));
  }
  return 0;
}
