// exp_bwd_cold.hip - development experiment: the read stream of k_fused_bwd at the headline size (64 x 338 688 floats =
// 86.7 MB), rotating through 8 tensors so that every launch reads from DRAM.  Which split of the batch over waves /
// loads in flight streams fastest?  hipcc -O3 --offload-arch=gfx950 -o exp_bwd_cold tools/exp_bwd_cold.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ inline float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ inline float4 ld(const float4* p) {
  if (NT) {
    f4v r = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(r.x, r.y, r.z, r.w);
  }
  return *p;
}

// WAVES waves split the batch; U loads in flight per lane; COLS float4 columns per lane
template <int WAVES, int U, bool NT, int COLS>
__global__ void __launch_bounds__(WAVES * 64) k_split(const float4* __restrict__ g, int batch, long long n4, float4* __restrict__ out) {
  __shared__ float4 part[WAVES][64 * COLS];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < COLS; ++c) {
    const long long q = ((long long)blockIdx.x * COLS + c) * 64 + lane;
    float4 a = make_float4(0, 0, 0, 0);
    if (q < n4) {
      int b = wid;
      for (; b + (U - 1) * WAVES < batch; b += U * WAVES) {
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = ld<NT>(g + (size_t)(b + WAVES * k) * n4 + q);
#pragma unroll
        for (int k = 0; k < U; ++k) a = add4(a, v[k]);
      }
      for (; b < batch; b += WAVES) a = add4(a, ld<NT>(g + (size_t)b * n4 + q));
    }
    part[wid][c * 64 + lane] = a;
  }
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const long long q = ((long long)blockIdx.x * COLS + c) * 64 + lane;
      if (q < n4) {
        float4 t = part[0][c * 64 + lane];
        for (int w = 1; w < WAVES; ++w) t = add4(t, part[w][c * 64 + lane]);
        out[q] = t;
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_linear(const float4* __restrict__ g, long long total4, float4* __restrict__ sink) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  float4 a = make_float4(0, 0, 0, 0);
  for (; i + 7 * stride < total4; i += 8 * stride) {
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = g[i + k * stride];
#pragma unroll
    for (int k = 0; k < 8; ++k) a = add4(a, v[k]);
  }
  for (; i < total4; i += stride) a = add4(a, g[i]);
  if (a.x == 12345.678f) sink[0] = a;
}

int main() {
  const int batch = 64, ring = 8;
  const long long n = 3LL * 336 * 336, n4 = n / 4, total4 = n4 * batch;
  std::vector<float4*> g(ring);
  float4* out;
  for (auto& p : g) {
    if (hipMalloc(&p, total4 * 16) != hipSuccess) return 1;
    (void)hipMemset(p, 0, total4 * 16);
  }
  if (hipMalloc(&out, n4 * 16) != hipSuccess) return 1;
  const double mb = total4 * 16 / 1e6;
  auto timeit = [&](auto launch, bool cold) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < ring; ++i) launch(g[cold ? i : 0]);
    (void)hipDeviceSynchronize();
    const int iters = 80;
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) launch(g[cold ? i % ring : 0]);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / iters * 1e3f;
  };
  auto report = [&](const char* name, auto launch) {
    float c = timeit(launch, true), h = timeit(launch, false);
    printf("  %-52s cold %6.2f us %5.2f TB/s | cached %6.2f us %5.2f TB/s\n", name, c, mb / c, h, mb / h);
  };
  printf("tensor %d x %lld floats = %.1f MB, ring of %d\n", batch, n, mb, ring);
#define SPLIT(W, U, NT, C) [&](float4* p) { hipLaunchKernelGGL((k_split<W, U, NT, C>), dim3((n4 + 64 * C - 1) / (64 * C)), dim3(W * 64), 0, 0, p, batch, n4, out); }
  report("linear read, 2048 blocks (ceiling)", [&](float4* p) { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, p, total4, out); });
  report("4 waves x 2 rounds of 8 (current)", SPLIT(4, 8, false, 1));
  report("4 waves x 2 rounds of 8, non-temporal", SPLIT(4, 8, true, 1));
  report("4 waves x 16 in flight", SPLIT(4, 16, false, 1));
  report("8 waves x 8 in flight (one round)", SPLIT(8, 8, false, 1));
  report("8 waves x 8 in flight, non-temporal", SPLIT(8, 8, true, 1));
  report("16 waves x 4 in flight", SPLIT(16, 4, false, 1));
  report("2 waves x 2 rounds of 16", SPLIT(2, 16, false, 1));
  report("4 waves x 8, two columns per lane", SPLIT(4, 8, false, 2));
  report("8 waves x 8, two columns per lane", SPLIT(8, 8, false, 2));
  return 0;
}
