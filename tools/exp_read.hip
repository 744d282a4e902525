// exp_read.hip - development experiment: how fast can the batch reduction  out[i] = sum_b g[b,i]
// stream a gradient tensor that does NOT fit the 256 MiB Infinity Cache (Mllama: 64 x 3.76 M
// floats = 963 MB), and which access pattern gets there?  hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ inline float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// (1) ceiling: linear grid-stride read of the whole tensor, 8 loads in flight per lane
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

// (2) current k_batch_reduce: block = 64 float4 columns, 4 waves split the batch, LDS combine
__global__ void __launch_bounds__(256) k_cols_split(const float4* __restrict__ g, int batch, long long n4,
                                                    float4* __restrict__ out) {
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long q = (long long)blockIdx.x * 64 + lane;
  float4 a = make_float4(0, 0, 0, 0);
  if (q < n4) {
    int b = wid;
    for (; b + 28 < batch; b += 32) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = g[(size_t)(b + 4 * k) * n4 + q];
#pragma unroll
      for (int k = 0; k < 8; ++k) a = add4(a, v[k]);
    }
    for (; b < batch; b += 4) a = add4(a, g[(size_t)b * n4 + q]);
  }
  part[wid][lane] = a;
  __syncthreads();
  if (wid == 0 && q < n4) out[q] = add4(add4(add4(part[0][lane], part[1][lane]), part[2][lane]), part[3][lane]);
}

// (3) one column per lane, the lane walks the whole batch (no LDS), U loads in flight
template <int U>
__global__ void __launch_bounds__(256) k_cols_own(const float4* __restrict__ g, int batch, long long n4,
                                                  float4* __restrict__ out) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= n4) return;
  float4 a = make_float4(0, 0, 0, 0);
  int b = 0;
  for (; b + U <= batch; b += U) {
    float4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = g[(size_t)(b + k) * n4 + q];
#pragma unroll
    for (int k = 0; k < U; ++k) a = add4(a, v[k]);
  }
  for (; b < batch; ++b) a = add4(a, g[(size_t)b * n4 + q]);
  out[q] = a;
}

// (4) persistent: a fixed number of workgroups walk the column chunks (grid-stride), one column per
// lane, whole batch per lane - fewer, longer-lived waves
template <int U>
__global__ void __launch_bounds__(256) k_cols_persist(const float4* __restrict__ g, int batch, long long n4,
                                                      float4* __restrict__ out) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
    float4 a = make_float4(0, 0, 0, 0);
    int b = 0;
    for (; b + U <= batch; b += U) {
      float4 v[U];
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] = g[(size_t)(b + k) * n4 + q];
#pragma unroll
      for (int k = 0; k < U; ++k) a = add4(a, v[k]);
    }
    for (; b < batch; ++b) a = add4(a, g[(size_t)b * n4 + q]);
    out[q] = a;
  }
}

template <typename F>
float timeit(F f, int iters) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
  const int batch = 64;
  const long long n = argc > 1 ? atoll(argv[1]) : 4LL * 3 * 560 * 560;   // floats per sample
  const long long n4 = n / 4, total4 = n4 * batch;
  float4 *g, *out;
  if (hipMalloc(&g, total4 * 16) != hipSuccess || hipMalloc(&out, n4 * 16) != hipSuccess) return 1;
  (void)hipMemset(g, 0, total4 * 16);
  const double mb = total4 * 16 / 1e6;
  printf("tensor %d x %lld floats = %.1f MB\n", batch, n, mb);
  auto report = [&](const char* name, float us) { printf("  %-44s %8.1f us  %5.2f TB/s\n", name, us, mb / us / 1e6 * 1e6 / 1e6); };
  for (int blocks : {2048, 4096, 8192})
    report(blocks == 2048 ? "linear 2048 blocks" : blocks == 4096 ? "linear 4096 blocks" : "linear 8192 blocks",
           timeit([&] { hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, g, total4, out); }, 20));
  report("columns, 4 waves split batch (current)",
         timeit([&] { hipLaunchKernelGGL(k_cols_split, dim3((n4 + 63) / 64), dim3(256), 0, 0, g, batch, n4, out); }, 20));
  report("own column, 8 in flight",
         timeit([&] { hipLaunchKernelGGL(k_cols_own<8>, dim3((n4 + 255) / 256), dim3(256), 0, 0, g, batch, n4, out); }, 20));
  report("own column, 16 in flight",
         timeit([&] { hipLaunchKernelGGL(k_cols_own<16>, dim3((n4 + 255) / 256), dim3(256), 0, 0, g, batch, n4, out); }, 20));
  for (int blocks : {512, 1024, 2048}) {
    char nm[64];
    snprintf(nm, sizeof nm, "persistent own column, 16 in flight, %d blocks", blocks);
    report(nm, timeit([&] { hipLaunchKernelGGL(k_cols_persist<16>, dim3(blocks), dim3(256), 0, 0, g, batch, n4, out); }, 20));
  }
  return 0;
}
