// exp_tail3.hip - development experiment (round 4): the prepared chain's tail (k_plan_tail<T, MODE>: one thread per (channel, y, x),
// flat partition, two run-time divisions, seven block sums per workgroup) with the THREE CHANNELS of a pixel in one thread on a
// (column chunk, row) grid - taps and weights looked up once, no divisions, a third of the block sums per pixel - as k_stage_bwd3_w
// does for the plain transposed resize.  Per-pixel arithmetic unchanged (p, m, v, grad, s_next compared bit for bit); the
// statistics / ||g|| partials are summed over another partition, so only their totals are compared.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o tools/exp_tail3.bin tools/exp_tail3.hip
#include "../adversarialvlm_amd/csrc/advx.hip"

#include <cstdio>
#include <vector>

using namespace advx;

template <int T, int MODE>
__global__ void __launch_bounds__(kBlock) k_plan_tail3(DStage st, CanvasGrad cg, float* __restrict__ p, const float* __restrict__ x0, float eps,
                                                       float c_fit, const float* __restrict__ mask, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ grad_p, OptScalars o,
                                                       float* __restrict__ s_next, double* __restrict__ img_rows_out,
                                                       double* __restrict__ norm_rows) {
  constexpr int COPIES = (MODE == 3) ? 2 : 1;
  constexpr bool DG = MODE == 2;
  const int ys = blockIdx.y;
  const int xs = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned row = blockIdx.x + gridDim.x * blockIdx.y;
  double acc[kStatSlots] = {0, 0, 0, 0, 0, 0};
  double nacc[1] = {0.0};
  if (xs < st.src_w) {
    const size_t plane_s = (size_t)st.src_h * st.src_w, o0 = (size_t)ys * st.src_w + xs;
    float pp[3], xv[3], mk[3], mm[3], vv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = (size_t)c * plane_s + o0;
      pp[c] = p[i]; xv[c] = x0[i]; mk[c] = mask[i];
      mm[c] = (o.kind == 0) ? m[i] : 0.0f;
      vv[c] = (o.kind == 0) ? v[i] : 0.0f;
    }
    // transposed gather, three channels (k_stage_bwd3_w's body)
    const int oy = st.tth.start[ys], oyc = st.tth.count[ys];
    const int ox = st.ttw.start[xs], oxc = st.ttw.count[xs];
    const float* wy = st.tth.w + (size_t)ys * st.tth.stride;
    const float* wx = st.ttw.w + (size_t)xs * st.ttw.stride;
    const size_t plane = (size_t)st.can_h * st.can_w;
    const int ly = max(oyc - 1, 0), lx = max(oxc - 1, 0);
    float wyv[T], wxv[T], r[3][T][T][COPIES + 1];
#pragma unroll
    for (int a = 0; a < T; ++a) { wyv[a] = wy[min(a, ly)]; wxv[a] = wx[min(a, lx)]; }
#pragma unroll
    for (int a = 0; a < T; ++a) {
      const size_t rowc = (size_t)(st.off_y + oy + min(a, ly)) * st.can_w + st.off_x + ox;
#pragma unroll
      for (int b = 0; b < T; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const size_t q = (size_t)c * plane + rowc + min(b, lx);
#pragma unroll
          for (int t = 0; t < COPIES; ++t) r[c][a][b][t] = cg.g[(size_t)t * cg.copy_stride + q];
          if (DG) r[c][a][b][COPIES] = cg.dgrad[q];
        }
    }
    float gsum[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < T; ++a) {
      float h[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int b = 0; b < T; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float g = 0.0f;
#pragma unroll
          for (int t = 0; t < COPIES; ++t) g += r[c][a][b][t];
          if (DG) g += r[c][a][b][COPIES];
          h[c] = (b < oxc) ? h[c] + wxv[b] * g : h[c];
        }
#pragma unroll
      for (int c = 0; c < 3; ++c) gsum[c] = (a < oyc) ? gsum[c] + wyv[a] * h[c] : gsum[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = (size_t)c * plane_s + o0;
      const float gs = st.normalise ? gsum[c] / st.stdv[c] : gsum[c];
      float pv = pp[c];
      const float t = tanhf(pv);
      const float s = xv[c] + eps * t;
      float gp = ((gs + imgfit_grad(s, c_fit)) * eps) * (1.0f - t * t);
      gp = gp * mk[c];
      nacc[0] += (double)gp * (double)gp;
      grad_p[i] = gp;
      if (o.kind == 0) {
        float m1 = mm[c], v1 = vv[c];
        adamw_element(pv, m1, v1, gp, o);
        p[i] = pv; m[i] = m1; v[i] = v1;
      } else {
        pv = pv - o.lr * sign_direction(gp);
        p[i] = pv;
      }
      const float xn = eps * tanhf(pv);
      const float sn = xv[c] + xn;
      s_next[i] = sn;
      stat_accumulate(sn, xn, acc);
    }
  }
  block_sum_store2<kStatSlots, 1>(acc, img_rows_out + (size_t)row * kStatSlots, nacc, norm_rows + row);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, unsigned seed, float lo, float hi) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 7u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * (float)(s >> 8) / 16777216.0f; }
  float* d;
  CK(hipMalloc(&d, n * 4));
  CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

struct State { float *p, *m, *v, *grad, *s_next; double *img, *norm; };

int main() {
  struct Case { const char* name; int kind, H, W; long long a0, a1; };
  const Case cases[] = {{"llava 512 -> 336", ADVX_KIND_LLAVA, 512, 512, 336, 336}, {"llava 336 -> 224 (small)", ADVX_KIND_LLAVA, 336, 336, 224, 224},
                        {"llava 1024 -> 336", ADVX_KIND_LLAVA, 1024, 1024, 336, 336}};
  for (const Case& cs : cases) {
    advx_plan_desc d;
    memset(&d, 0, sizeof(d));
    d.kind = cs.kind; d.in_h = cs.H; d.in_w = cs.W; d.a0 = cs.a0; d.a1 = cs.a1;
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, sd[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    for (int c = 0; c < 3; ++c) { d.mean[c] = mean[c]; d.std[c] = sd[c]; }
    advx_plan* p = nullptr;
    if (advx_plan_create(&d, &p) != 0 || advx_plan_upload(p, nullptr) != 0) { printf("plan failed: %s\n", advx_last_error()); return 1; }
    const DStage& D = p->dstage[0];
    const int T = std::max(D.tth.stride, D.ttw.stride);
    const size_t n = 3ull * cs.H * cs.W, n_can = 3ull * D.can_h * D.can_w;
    float* gcan = dev_rand(n_can, 2, -0.01f, 0.01f);
    float* x0 = dev_rand(n, 3, 0.f, 1.f);
    float* mask = dev_rand(n, 4, 1.f, 1.f);
    CanvasGrad cg;
    cg.g = gcan; cg.copies = 1; cg.copy_stride = (long long)n_can; cg.dgrad = nullptr;
    OptScalars o;
    o.kind = 0; o.apply = 1; o.lr = 1e-2f; o.decay = 1.0f - 1e-2f * 1e-2f; o.w1 = 0.1f; o.beta2 = 0.999f; o.w2 = 0.001f;
    o.bias2_sqrt = 0.0316227766f; o.eps = 1e-8f; o.neg_step_size = -0.1f;
    auto make = [&]() {
      State s;
      s.p = dev_rand(n, 11, -0.05f, 0.05f); s.m = dev_rand(n, 12, -1e-3f, 1e-3f); s.v = dev_rand(n, 13, 0.f, 1e-5f);
      CK(hipMalloc(&s.grad, n * 4)); CK(hipMalloc(&s.s_next, n * 4));
      CK(hipMalloc(&s.img, 16384 * 8 * kStatSlots)); CK(hipMalloc(&s.norm, 16384 * 8));
      return s;
    };
    const int flat_blocks = (int)((n + kBlock - 1) / kBlock);
    const dim3 grid3((cs.W + kBlock - 1) / kBlock, cs.H);
    auto shipped = [&](State& s) {
      if (T <= 2) hipLaunchKernelGGL((k_plan_tail<2, 1>), dim3(flat_blocks), dim3(kBlock), 0, 0, D, cg, s.p, x0, 0.5f, 2.0f / (float)n, mask, s.m, s.v, s.grad, o, s.s_next, s.img, s.norm, (const double*)nullptr, 0, (float*)nullptr);
      else if (T == 3) hipLaunchKernelGGL((k_plan_tail<3, 1>), dim3(flat_blocks), dim3(kBlock), 0, 0, D, cg, s.p, x0, 0.5f, 2.0f / (float)n, mask, s.m, s.v, s.grad, o, s.s_next, s.img, s.norm, (const double*)nullptr, 0, (float*)nullptr);
      else hipLaunchKernelGGL((k_plan_tail<4, 1>), dim3(flat_blocks), dim3(kBlock), 0, 0, D, cg, s.p, x0, 0.5f, 2.0f / (float)n, mask, s.m, s.v, s.grad, o, s.s_next, s.img, s.norm, (const double*)nullptr, 0, (float*)nullptr);
    };
    auto three = [&](State& s, int threads) {
      const dim3 g((cs.W + threads - 1) / threads, cs.H);
      if (T <= 2) hipLaunchKernelGGL((k_plan_tail3<2, 1>), g, dim3(threads), 0, 0, D, cg, s.p, x0, 0.5f, 2.0f / (float)n, mask, s.m, s.v, s.grad, o, s.s_next, s.img, s.norm);
      else if (T == 3) hipLaunchKernelGGL((k_plan_tail3<3, 1>), g, dim3(threads), 0, 0, D, cg, s.p, x0, 0.5f, 2.0f / (float)n, mask, s.m, s.v, s.grad, o, s.s_next, s.img, s.norm);
      else hipLaunchKernelGGL((k_plan_tail3<4, 1>), g, dim3(threads), 0, 0, D, cg, s.p, x0, 0.5f, 2.0f / (float)n, mask, s.m, s.v, s.grad, o, s.s_next, s.img, s.norm);
    };
    // bit-identity of one step
    {
      State a = make(), b = make();
      shipped(a);
      three(b, 128);
      CK(hipDeviceSynchronize());
      std::vector<float> ha(n), hb(n);
      int bad = 0;
      float* pa[5] = {a.p, a.m, a.v, a.grad, a.s_next};
      float* pb[5] = {b.p, b.m, b.v, b.grad, b.s_next};
      for (int k = 0; k < 5; ++k) {
        CK(hipMemcpy(ha.data(), pa[k], n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), pb[k], n * 4, hipMemcpyDeviceToHost));
        bad += memcmp(ha.data(), hb.data(), n * 4) != 0;
      }
      printf("%s (transposed rows of %d): three-channel form after one step: %s (p, m, v, grad, s_next)\n", cs.name, T, bad ? "DIFFERENT" : "bit-identical");
    }
    State st = make();
    auto timeit = [&](auto f) {
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      for (int i = 0; i < 20; ++i) f();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < 300; ++i) f();
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      return ms / 300 * 1e3f;
    };
    for (int rep = 0; rep < 2; ++rep)
      printf("  shipped k_plan_tail<T,1> %6.2f us | three channels per thread: 128 threads %6.2f us, 256 threads %6.2f us (back-to-back launches)\n",
             timeit([&] { shipped(st); }), timeit([&] { three(st, 128); }), timeit([&] { three(st, 256); }));
    advx_plan_destroy(p);
  }
  return 0;
}
