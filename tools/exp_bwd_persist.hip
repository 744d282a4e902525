// exp_bwd_persist.hip - development experiment: can k_fused_bwd's cold time (20.4-21.2 us for
// 86.7 MB of gradient + 14.9 MB of per-pixel state at the headline size) be brought to <= 19.0 us by making it
// persistent over tiles with the next tile's loads issued before the current tile's epilogue?
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o tools/exp_bwd_persist.bin tools/exp_bwd_persist.hip
//
// Every variant runs the SAME arithmetic in the same summation order on the same data (outputs compared bit for bit
// with the shipped kernel) over a ring of 8 gradient tensors (cold: every byte from DRAM):
//   shipped      advx::k_fused_bwd<true, 0> as libadvx launches it (one 256-pixel tile per workgroup, 1323 workgroups,
//                all resident at once: every load of the tensor is in flight from the first microsecond)
//   persistent   G workgroups loop over the tiles; the batch loads and the state loads of tile k+1 are issued before
//                tile k's LDS combine / tanh / AdamW / stores (two register sets)
//   traffic-only the shipped kernel's loads and stores with the arithmetic removed (sum, copy): what the BYTES cost
//   stream-only  the batch reduction alone (exp_bwd_cold.hip's kernel), for the fixed cost of a launch of this size
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../adversarialvlm_amd/csrc/advx_kernels.h"

using namespace advx;

// ------------------------------------------------------------------------------------------------ persistent form
struct TileRegs {
  float4 v[16];                    // this wave's 16 batch rows of the tile's 64 columns
  float pp, xv, mk, mm, vv;        // this thread's pixel state
};

__device__ inline void tile_issue(TileRegs& r, const float* __restrict__ g, int batch, long long n, long long tile, int lane,
                                  int wid, const float* p, const float* x0, const float* mask, const float* m, const float* v) {
  const long long i0 = (tile * kWave + lane) << 2;
#pragma unroll
  for (int k = 0; k < 16; ++k) r.v[k] = *reinterpret_cast<const float4*>(g + (size_t)(wid + 4 * k) * n + i0);
  const long long i = tile * (kWave * 4) + threadIdx.x;
  r.pp = p[i]; r.xv = x0[i]; r.mk = mask[i]; r.mm = m[i]; r.vv = v[i];
}

__device__ inline void tile_finish(const TileRegs& r, long long tile, float4 (*part4)[kWave], float eps, const FusedGeom& geo,
                                   float c_fit, float* p, float* m, float* v, float* grad_p, const OptScalars& o, float* s_next,
                                   float* v_buf, double& nacc) {
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const long long i = tile * (kWave * 4) + threadIdx.x;
  const int c = (int)(i / geo.plane);
  const float sd = geo.stdv[c];
  float pp = r.pp, mm = r.mm, vv = r.vv;
  const float t = tanhf(pp);
  const float s = r.xv + eps * t;
  const float fit = imgfit_grad(s, c_fit);
  const float dtanh = 1.0f - t * t;
  // same order as batch_column_sum: two rounds of 8, each summed front to back
  float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < 16; ++k) a = f4add(a, r.v[k]);
  __syncthreads();                 // the previous tile's readers are done with part4
  part4[wid][lane] = a;
  __syncthreads();
  const float(*part)[kWave * 4] = reinterpret_cast<const float(*)[kWave * 4]>(&part4[0][0]);
  float gs = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
  float gx = gs / sd + fit;
  float gp = (gx * eps) * dtanh;
  gp = gp * r.mk;
  nacc += (double)gp * (double)gp;
  grad_p[i] = gp;
  adamw_element(pp, mm, vv, gp, o);
  p[i] = pp; m[i] = mm; v[i] = vv;
  float sn = r.xv + eps * tanhf(pp);
  s_next[i] = sn;
  v_buf[i] = (sn - geo.mean[c]) / sd;
}

// n a multiple of 1024 is NOT required: tiles = n / 1024 full tiles here (the headline size 338688 = 330.75 tiles is
// padded up by the harness' allocation; the experiment runs 1323 quarter... see main: tile = 256 pixels, 1323 tiles)
__global__ void __launch_bounds__(kBlock) k_bwd_persistent(const float* __restrict__ g, int batch, float* __restrict__ p,
                                                           const float* __restrict__ x0, float eps, FusedGeom geo, float c_fit,
                                                           const float* __restrict__ mask, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ grad_p, OptScalars o,
                                                           float* __restrict__ s_next, float* __restrict__ v_buf,
                                                           double* __restrict__ norm_partials, int tiles) {
  __shared__ float4 part4[kBlock / kWave][kWave];
  const long long n = 3LL * geo.plane;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  double nacc = 0.0;
  TileRegs A, B;
  long long t = blockIdx.x;
  if (t < tiles) tile_issue(A, g, batch, n, t, lane, wid, p, x0, mask, m, v);
  while (t < tiles) {
    const long long t1 = t + gridDim.x;
    if (t1 < tiles) tile_issue(B, g, batch, n, t1, lane, wid, p, x0, mask, m, v);
    tile_finish(A, t, part4, eps, geo, c_fit, p, m, v, grad_p, o, s_next, v_buf, nacc);
    if (t1 >= tiles) break;
    const long long t2 = t1 + gridDim.x;
    if (t2 < tiles) tile_issue(A, g, batch, n, t2, lane, wid, p, x0, mask, m, v);
    tile_finish(B, t1, part4, eps, geo, c_fit, p, m, v, grad_p, o, s_next, v_buf, nacc);
    t = t2;
  }
  double acc[1] = {nacc};
  block_sum_store<1>(acc, norm_partials + blockIdx.x);
}

// --------------------------------------------------------------------------------------------- traffic-only form
__global__ void __launch_bounds__(kBlock) k_bwd_traffic(const float* __restrict__ g, int batch, float* __restrict__ p,
                                                        const float* __restrict__ x0, FusedGeom geo, const float* __restrict__ mask,
                                                        float* __restrict__ m, float* __restrict__ v, float* __restrict__ grad_p,
                                                        float* __restrict__ s_next, float* __restrict__ v_buf) {
  __shared__ float4 part4[kBlock / kWave][kWave];
  const long long n = 3LL * geo.plane, n4 = n >> 2;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const long long q = (long long)blockIdx.x * kWave + lane;
  const long long i = (long long)blockIdx.x * (kWave * 4) + threadIdx.x;
  float pp = 0.f, xv = 0.f, mk = 0.f, mm = 0.f, vv = 0.f;
  if (i < n) { pp = p[i]; xv = x0[i]; mk = mask[i]; mm = m[i]; vv = v[i]; }
  float4 a = make_float4(0, 0, 0, 0);
  if (q < n4) a = batch_column_sum<0>(g, batch, n, q << 2, wid, kBlock / kWave);
  part4[wid][lane] = a;
  __syncthreads();
  const float(*part)[kWave * 4] = reinterpret_cast<const float(*)[kWave * 4]>(&part4[0][0]);
  if (i < n) {
    float gs = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
    grad_p[i] = gs + mk; p[i] = pp + gs; m[i] = mm + gs; v[i] = vv + gs; s_next[i] = xv + gs; v_buf[i] = xv - gs;
  }
}

__global__ void __launch_bounds__(kBlock) k_stream_only(const float* __restrict__ g, int batch, long long n, float* __restrict__ out) {
  __shared__ float4 part4[kBlock / kWave][kWave];
  const long long n4 = n >> 2;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const long long q = (long long)blockIdx.x * kWave + lane;
  float4 a = make_float4(0, 0, 0, 0);
  if (q < n4) a = batch_column_sum<0>(g, batch, n, q << 2, wid, kBlock / kWave);
  part4[wid][lane] = a;
  __syncthreads();
  if (wid == 0 && q < n4)
    *reinterpret_cast<float4*>(out + (q << 2)) = f4add(f4add(f4add(part4[0][lane], part4[1][lane]), part4[2][lane]), part4[3][lane]);
}

// ------------------------------------------------------------------------------------------------------- harness
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
  const int tiles = (int)((n4 + kWave - 1) / kWave);             // 1323, and n = 1323 * 256 exactly
  if ((long long)tiles * 256 != n) { printf("size is not a whole number of tiles\n"); return 1; }
  std::vector<float*> g(ring);
  for (int r = 0; r < ring; ++r) g[r] = dev_floats((size_t)batch * n, -0.01f, 0.01f, 100 + r);
  FusedGeom geo;
  geo.plane = H * W;
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, sd[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  for (int c = 0; c < 3; ++c) { geo.mean[c] = mean[c]; geo.stdv[c] = sd[c]; }
  OptScalars o;
  o.kind = 0; o.apply = 1; o.lr = 1e-2f; o.decay = 1.0f - 1e-2f * 1e-2f; o.w1 = 0.1f; o.beta2 = 0.999f; o.w2 = 0.001f;
  o.bias2_sqrt = 0.0316227766f; o.eps = 1e-8f; o.neg_step_size = -0.1f;
  auto make_state = [&]() {
    State s;
    s.p = dev_floats(n, -0.05f, 0.05f, 1); s.x0 = dev_floats(n, 0.f, 1.f, 2); s.mask = dev_floats(n, 1.f, 1.f, 3);
    s.m = dev_floats(n, -1e-3f, 1e-3f, 4); s.v = dev_floats(n, 0.f, 1e-5f, 5);
    CK(hipMalloc(&s.grad, n * 4)); CK(hipMalloc(&s.s_next, n * 4)); CK(hipMalloc(&s.v_buf, n * 4));
    CK(hipMalloc(&s.norm, 4096 * 8)); CK(hipMalloc(&s.hdr, sizeof(FusedHeader))); CK(hipMemset(s.hdr, 0, sizeof(FusedHeader)));
    CK(hipMalloc(&s.img, 4096 * 8 * kStatSlots)); CK(hipMalloc(&s.stats, 64 * 4)); CK(hipMemset(s.stats, 0, 64 * 4));
    return s;
  };
  auto shipped = [&](State& s, float* gr) {
    hipLaunchKernelGGL((k_fused_bwd<true, 0, false>), dim3(tiles), dim3(kBlock), 0, 0, (const void*)gr, batch, s.p, s.x0, 0.5f, geo,
                       2.0f / (float)n, s.mask, s.m, s.v, s.grad, o, s.s_next, s.v_buf, s.norm, s.stats, s.hdr, s.img, (SchedDev*)nullptr);
  };
  auto persistent = [&](int G) {
    return [&, G](State& s, float* gr) {
      hipLaunchKernelGGL(k_bwd_persistent, dim3(G), dim3(kBlock), 0, 0, gr, batch, s.p, s.x0, 0.5f, geo, 2.0f / (float)n, s.mask, s.m,
                         s.v, s.grad, o, s.s_next, s.v_buf, s.norm, tiles);
    };
  };
  auto traffic = [&](State& s, float* gr) {
    hipLaunchKernelGGL(k_bwd_traffic, dim3(tiles), dim3(kBlock), 0, 0, gr, batch, s.p, s.x0, geo, s.mask, s.m, s.v, s.grad, s.s_next, s.v_buf);
  };
  auto stream = [&](State& s, float* gr) { hipLaunchKernelGGL(k_stream_only, dim3(tiles), dim3(kBlock), 0, 0, gr, batch, n, s.grad); };

  // ---- bit-identity of the persistent form with the shipped kernel (one step from equal states)
  {
    State a = make_state(), b = make_state();
    shipped(a, g[0]);
    persistent(512)(b, g[0]);
    CK(hipDeviceSynchronize());
    std::vector<float> ha(n), hb(n);
    int bad = 0;
    float* pa[6] = {a.p, a.m, a.v, a.grad, a.s_next, a.v_buf};
    float* pb[6] = {b.p, b.m, b.v, b.grad, b.s_next, b.v_buf};
    for (int k = 0; k < 6; ++k) {
      CK(hipMemcpy(ha.data(), pa[k], n * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), pb[k], n * 4, hipMemcpyDeviceToHost));
      bad += memcmp(ha.data(), hb.data(), n * 4) != 0;
    }
    printf("persistent vs shipped after one step: %s (p, m, v, grad, s_next, v_buf)\n", bad ? "DIFFERENT" : "bit-identical");
  }
  State st = make_state();
  auto timeit = [&](auto launch, bool cold) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < ring; ++i) launch(st, g[cold ? i : 0]);
    CK(hipDeviceSynchronize());
    const int iters = 200;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) launch(st, g[cold ? i % ring : 0]);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;
  };
  const double mb_g = (double)batch * n * 4 / 1e6, mb_state = 11.0 * n * 4 / 1e6;
  auto report = [&](const char* name, auto launch, double mb) {
    float c = timeit(launch, true), h = timeit(launch, false);
    printf("  %-44s cold %6.2f us %5.2f TB/s | cached %6.2f us %5.2f TB/s\n", name, c, mb / c, h, mb / h);
  };
  printf("gradient %.1f MB + state %.1f MB (5 reads, 6 writes per pixel); back-to-back launches, ring of %d\n", mb_g, mb_state, ring);
  for (int rep = 0; rep < 2; ++rep) {
    report("shipped k_fused_bwd (1323 workgroups)", shipped, mb_g + mb_state);
    report("persistent, 256 workgroups", persistent(256), mb_g + mb_state);
    report("persistent, 512 workgroups", persistent(512), mb_g + mb_state);
    report("persistent, 768 workgroups", persistent(768), mb_g + mb_state);
    report("persistent, 1024 workgroups", persistent(1024), mb_g + mb_state);
    report("traffic only (same loads and stores)", traffic, mb_g + mb_state);
    report("stream only (batch reduction, 1.4 MB out)", stream, mb_g + n * 4 / 1e6);
  }
  return 0;
}
