// exp_bwd_occ.hip - development experiment (round 4): k_fused_bwd's 1323 workgroups are all resident at once and all
// finish their streams together, so the 8.1 MB of per-pixel stores of every workgroup hit the memory system in one burst at
// the very end of the launch.  Does it pay to let the workgroups run in TWO OR THREE ROUNDS - fewer resident at once, so
// that the epilogues of one round overlap the streams of the next?  Occupancy is cut from outside with dynamic LDS the
// kernel never touches (no code change), and, for comparison, the XCD maps 0 / 2.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o tools/exp_bwd_occ.bin tools/exp_bwd_occ.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../adversarialvlm_amd/csrc/advx_kernels.h"

using namespace advx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct State {
  float *p, *x0, *mask, *m, *v, *grad, *s_next, *v_buf;
  double* norm;
  FusedHeader* hdr;
  double* img;
  float* stats;
};

static float* dev_floats(size_t n, float lo, float hi, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = lo + (hi - lo) * (float)(s >> 8) / 16777216.0f;
  }
  float* d;
  CK(hipMalloc(&d, n * 4));
  CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

int main() {
  const int batch = 64, ring = 8, H = 336, W = 336;
  const long long n = 3LL * H * W, n4 = n / 4;
  const int tiles = (int)((n4 + kWave - 1) / kWave);
  std::vector<float*> g(ring);
  for (int r = 0; r < ring; ++r) g[r] = dev_floats((size_t)batch * n, -0.01f, 0.01f, 100 + r);
  FusedGeom geo;
  geo.plane = H * W;
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, sd[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  for (int c = 0; c < 3; ++c) { geo.mean[c] = mean[c]; geo.stdv[c] = sd[c]; }
  OptScalars o;
  o.kind = 0; o.apply = 1; o.lr = 1e-2f; o.decay = 1.0f - 1e-2f * 1e-2f; o.w1 = 0.1f; o.beta2 = 0.999f; o.w2 = 0.001f;
  o.bias2_sqrt = 0.0316227766f; o.eps = 1e-8f; o.neg_step_size = -0.1f;
  State s;
  s.p = dev_floats(n, -0.05f, 0.05f, 1); s.x0 = dev_floats(n, 0.f, 1.f, 2); s.mask = dev_floats(n, 1.f, 1.f, 3);
  s.m = dev_floats(n, -1e-3f, 1e-3f, 4); s.v = dev_floats(n, 0.f, 1e-5f, 5);
  CK(hipMalloc(&s.grad, n * 4)); CK(hipMalloc(&s.s_next, n * 4)); CK(hipMalloc(&s.v_buf, n * 4));
  CK(hipMalloc(&s.norm, 4096 * 8)); CK(hipMalloc(&s.hdr, sizeof(FusedHeader))); CK(hipMemset(s.hdr, 0, sizeof(FusedHeader)));
  CK(hipMalloc(&s.img, 4096 * 8 * kStatSlots)); CK(hipMalloc(&s.stats, 64 * 4)); CK(hipMemset(s.stats, 0, 64 * 4));

  auto launch = [&](float* gr, int lds_bytes, int xmap) {
    const int grid = xmap ? ((tiles + 7) & ~7) : tiles;
    hipLaunchKernelGGL((k_fused_bwd<true, 0, false>), dim3(grid), dim3(kBlock), lds_bytes, 0, (const void*)gr, batch, s.p, s.x0, 0.5f, geo,
                       2.0f / (float)n, s.mask, s.m, s.v, s.grad, o, s.s_next, s.v_buf, s.norm, s.stats, s.hdr, s.img,
                       (SchedDev*)nullptr, xmap);
  };
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_bwd<true, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  (void)hipGetLastError();
  auto timeit = [&](int lds_bytes, int xmap, bool cold) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < ring; ++i) launch(g[cold ? i : 0], lds_bytes, xmap);
    CK(hipDeviceSynchronize());
    const int iters = 200;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) launch(g[cold ? i % ring : 0], lds_bytes, xmap);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;
  };
  // workgroups per CU allowed by the dynamic LDS (the kernel's own 4 KB + 2 KB of block_sum_store on top)
  const int lds[] = {0, 20 * 1024, 26 * 1024, 34 * 1024, 46 * 1024, 60 * 1024};
  const char* what[] = {"as shipped (5.2 per CU, all resident)", "~6 per CU", "~5 per CU", "~4 per CU (1024 resident)", "~3 per CU (768 resident)",
                        "~2 per CU (512 resident)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int xmap : {2, 0})
      for (int k = 0; k < 6; ++k) {
        const float c = timeit(lds[k], xmap, true), h = timeit(lds[k], xmap, false);
        printf("xmap %d  dynamic LDS %5d B  %-40s cold %6.2f us | cached %6.2f us\n", xmap, lds[k], what[k], c, h);
      }
  return 0;
}
