// exp_tile_resize.hip - development experiment: the image-sized gathers (plan resize forward: k_stage0_fwd_multi, transposed:
// k_stage_bwd3) as LDS-TILED SEPARABLE kernels.  A gather thread loads rows x columns source elements per output (25 at five taps per
// axis, 49 at the seven of a composed crop); a workgroup that loads the source patch of a canvas tile ONCE into LDS and runs the
// two 1-D passes there loads ~2 elements per output, with the same operations in the same order per output (bit-identical).
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o tools/exp_tile_resize.bin tools/exp_tile_resize.hip
#include "../adversarialvlm_amd/csrc/advx.hip"

#include <cstdio>
#include <vector>

using namespace advx;

constexpr int TH = 16, TW = 64;          // output tile of one workgroup (one channel)
constexpr int PR = 56, PC = 176;         // source patch capacity in LDS (rows, columns)
constexpr int SMAX = 16;                 // taps per axis the weight rows in LDS hold

// canvas[c, y0.., x0..] = normalise(pad | sum_a wy[a] * (sum_b wx[b] * src[...]))   (stage_fwd_value, !inner_axis_h)
__global__ void __launch_bounds__(256) k_stage_fwd_tile(DStage st, const float* __restrict__ src, long long src_cstride, int src_rstride,
                                                        float* __restrict__ canvas) {
  __shared__ float P[PR][PC];
  __shared__ float Hh[PR][TW];
  __shared__ float wxL[SMAX][TW], wyL[SMAX][TH];
  __shared__ int sxL[TW], cxL[TW], syL[TH], cyL[TH];
  const int tx = threadIdx.x & (TW - 1), ty = threadIdx.x / TW;      // 64 x 4
  const int c = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const int ry0 = max(y0 - st.off_y, 0), ry1 = min(y0 + TH - st.off_y, st.res_h);
  const int rx0 = max(x0 - st.off_x, 0), rx1 = min(x0 + TW - st.off_x, st.res_w);
  const bool any = ry0 < ry1 && rx0 < rx1;
  int ys0 = 0, nrows = 0, xs0 = 0, ncols = 0;
  if (any) {
    ys0 = st.th.start[ry0];
    nrows = st.th.start[ry1 - 1] + st.th.count[ry1 - 1] - ys0;
    xs0 = st.tw.start[rx0];
    ncols = st.tw.start[rx1 - 1] + st.tw.count[rx1 - 1] - xs0;
    // tables of the tile's rows and columns
    if (threadIdx.x < TW) {
      const int rx = rx0 + (int)threadIdx.x;
      const bool on = rx < rx1;
      sxL[threadIdx.x] = on ? st.tw.start[rx] - xs0 : 0;
      cxL[threadIdx.x] = on ? st.tw.count[rx] : 0;
      for (int b = 0; b < st.tw.stride; ++b) wxL[b][threadIdx.x] = on ? st.tw.w[(size_t)rx * st.tw.stride + b] : 0.0f;
    } else if (threadIdx.x < TW + TH) {
      const int k = (int)threadIdx.x - TW, ry = ry0 + k;
      const bool on = ry < ry1;
      syL[k] = on ? st.th.start[ry] - ys0 : 0;
      cyL[k] = on ? st.th.count[ry] : 0;
      for (int a = 0; a < st.th.stride; ++a) wyL[a][k] = on ? st.th.w[(size_t)ry * st.th.stride + a] : 0.0f;
    }
    const float* sp = src + (size_t)c * src_cstride + (size_t)ys0 * src_rstride + xs0;
    for (int r = ty; r < nrows; r += 4)
      for (int cc = tx; cc < ncols; cc += TW) P[r][cc] = sp[(size_t)r * src_rstride + cc];
  }
  __syncthreads();
  if (any) {
    const int nx = rx1 - rx0;
    if (tx < nx) {
      const int s0 = sxL[tx], cnt = cxL[tx];
      for (int r = ty; r < nrows; r += 4) {
        float h = 0.0f;
        for (int b = 0; b < cnt; ++b) h += wxL[b][tx] * P[r][s0 + b];
        Hh[r][tx] = h;
      }
    }
  }
  __syncthreads();
  for (int yy = ty; yy < TH; yy += 4) {
    const int y = y0 + yy, x = x0 + tx;
    if (y >= st.can_h || x >= st.can_w) continue;
    const int ry = y - st.off_y, rx = x - st.off_x;
    float v;
    if (ry >= 0 && ry < st.res_h && rx >= 0 && rx < st.res_w) {
      const int k = ry - ry0, xi = rx - rx0;
      const int s0 = syL[k], cnt = cyL[k];
      v = 0.0f;
      for (int a = 0; a < cnt; ++a) v += wyL[a][k] * Hh[s0 + a][xi];
    } else {
      v = st.pad_value;
    }
    if (st.normalise) v = (v - st.mean[c]) / st.stdv[c];
    canvas[((size_t)c * st.can_h + y) * st.can_w + x] = v;
  }
}

// gsrc[c, ys, xs] = (sum_a wyT[a] * (sum_b wxT[b] * G[oy+a][ox+b])) / std      (stage_bwd_value; one copy, no dgrad)
__global__ void __launch_bounds__(256) k_stage_bwd_tile(DStage st, const float* __restrict__ gcan, float* __restrict__ gsrc,
                                                        long long gsrc_cstride, int gsrc_rstride) {
  __shared__ float P[PR][PC];
  __shared__ float Hh[PR][TW];
  __shared__ float wxL[SMAX][TW], wyL[SMAX][TH];
  __shared__ int sxL[TW], cxL[TW], syL[TH], cyL[TH];
  __shared__ int lim[4];
  const int tx = threadIdx.x & (TW - 1), ty = threadIdx.x / TW;
  const int c = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const int y1 = min(y0 + TH, st.src_h), x1 = min(x0 + TW, st.src_w);
  // the rows / columns of the tile that have taps form a contiguous run (rows outside a crop window have none)
  if (threadIdx.x == 0) {
    int a0 = -1, a1 = -1;
    for (int y = y0; y < y1; ++y)
      if (st.tth.count[y] > 0) { if (a0 < 0) a0 = y; a1 = y; }
    int b0 = -1, b1 = -1;
    for (int x = x0; x < x1; ++x)
      if (st.ttw.count[x] > 0) { if (b0 < 0) b0 = x; b1 = x; }
    lim[0] = a0; lim[1] = a1; lim[2] = b0; lim[3] = b1;
  }
  __syncthreads();
  const int a0 = lim[0], a1 = lim[1], b0 = lim[2], b1 = lim[3];
  const bool any = a0 >= 0 && b0 >= 0;
  int oy0 = 0, nrows = 0, ox0 = 0, ncols = 0;
  if (any) {
    oy0 = st.tth.start[a0];
    nrows = st.tth.start[a1] + st.tth.count[a1] - oy0;
    ox0 = st.ttw.start[b0];
    ncols = st.ttw.start[b1] + st.ttw.count[b1] - ox0;
    if (threadIdx.x < TW) {
      const int x = x0 + (int)threadIdx.x;
      const bool on = x < x1 && st.ttw.count[x] > 0;
      sxL[threadIdx.x] = on ? st.ttw.start[x] - ox0 : 0;
      cxL[threadIdx.x] = on ? st.ttw.count[x] : 0;
      for (int b = 0; b < st.ttw.stride; ++b) wxL[b][threadIdx.x] = on ? st.ttw.w[(size_t)x * st.ttw.stride + b] : 0.0f;
    } else if (threadIdx.x < TW + TH) {
      const int k = (int)threadIdx.x - TW, y = y0 + k;
      const bool on = y < y1 && st.tth.count[y] > 0;
      syL[k] = on ? st.tth.start[y] - oy0 : 0;
      cyL[k] = on ? st.tth.count[y] : 0;
      for (int a = 0; a < st.tth.stride; ++a) wyL[a][k] = on ? st.tth.w[(size_t)y * st.tth.stride + a] : 0.0f;
    }
    const float* gp = gcan + ((size_t)c * st.can_h + (st.off_y + oy0)) * st.can_w + st.off_x + ox0;
    for (int r = ty; r < nrows; r += 4)
      for (int cc = tx; cc < ncols; cc += TW) P[r][cc] = 0.0f + gp[(size_t)r * st.can_w + cc];     // canvas_grad_at: g = 0; g += copy
  }
  __syncthreads();
  if (any) {
    const int cnt = cxL[tx], s0 = sxL[tx];
    for (int r = ty; r < nrows; r += 4) {
      float h = 0.0f;
      for (int b = 0; b < cnt; ++b) h += wxL[b][tx] * P[r][s0 + b];
      Hh[r][tx] = h;
    }
  }
  __syncthreads();
  for (int yy = ty; yy < TH; yy += 4) {
    const int y = y0 + yy, x = x0 + tx;
    if (y >= st.src_h || x >= st.src_w) continue;
    float v = 0.0f;
    if (any) {
      const int cnt = cyL[yy], s0 = syL[yy];
      for (int a = 0; a < cnt; ++a) v += wyL[a][yy] * Hh[s0 + a][tx];
    }
    if (st.normalise) v = v / st.stdv[c];
    gsrc[(size_t)c * gsrc_cstride + (size_t)y * gsrc_rstride + x] = v;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 7u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.0f - 0.3f; }
  float* d;
  CK(hipMalloc(&d, n * 4));
  CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

template <class F>
static float timeit(F f, int iters = 200) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 20; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters * 1e3f;
}

int main() {
  const int srcs[] = {404, 512, 700, 1020};
  printf("LLaVA plans src x src -> 336 x 336, one plan; back-to-back launches (no cold-cache rotation: the tensors are a few MB)\n");
  for (int src : srcs) {
    advx_plan_desc d;
    memset(&d, 0, sizeof(d));
    d.kind = ADVX_KIND_LLAVA; d.in_h = src; d.in_w = src; d.a0 = 336; d.a1 = 336;
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, sd[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    for (int c = 0; c < 3; ++c) { d.mean[c] = mean[c]; d.std[c] = sd[c]; }
    advx_plan* p = nullptr;
    if (advx_plan_create(&d, &p) != 0 || advx_plan_upload(p, nullptr) != 0) { printf("plan failed: %s\n", advx_last_error()); return 1; }
    const DStage& D = p->dstage[0];
    const size_t n_img = 3ull * src * src, n_can = 3ull * D.can_h * D.can_w;
    float* img = dev_rand(n_img, 1);
    float* gcan = dev_rand(n_can, 2);
    float *can_a, *can_b, *gs_a, *gs_b;
    CK(hipMalloc(&can_a, n_can * 4)); CK(hipMalloc(&can_b, n_can * 4)); CK(hipMalloc(&gs_a, n_img * 4)); CK(hipMalloc(&gs_b, n_img * 4));
    // ---- forward
    MultiFwd mf;
    memset(&mf, 0, sizeof(mf));
    mf.n = 1; mf.st[0] = D; mf.canvas[0] = can_a;
    TapBuild none;
    memset(&none, 0, sizeof(none));
    auto gather_f = [&]() {
      hipLaunchKernelGGL(k_stage0_fwd_multi, dim3((D.can_w + kRowBlock - 1) / kRowBlock, D.can_h, 1), dim3(kRowBlock), 0, 0, mf, (const float*)img,
                         (long long)src * src, src, (const double*)nullptr, 0, 0LL, (float*)nullptr, (const double*)nullptr, 0, none, none, 0);
    };
    auto tile_f = [&]() {
      hipLaunchKernelGGL(k_stage_fwd_tile, dim3((D.can_w + TW - 1) / TW, (D.can_h + TH - 1) / TH, 3), dim3(256), 0, 0, D, (const float*)img,
                         (long long)src * src, src, can_b);
    };
    // ---- transposed
    CanvasGrad cg;
    cg.g = gcan; cg.copies = 1; cg.copy_stride = (long long)n_can; cg.dgrad = nullptr;
    auto gather_b = [&]() {
      hipLaunchKernelGGL(k_stage_bwd3, dim3((D.src_w + 127) / 128, D.src_h), dim3(128), 0, 0, D, cg, gs_a, (long long)src * src, src, 0);
    };
    auto tile_b = [&]() {
      hipLaunchKernelGGL(k_stage_bwd_tile, dim3((D.src_w + TW - 1) / TW, (D.src_h + TH - 1) / TH, 3), dim3(256), 0, 0, D, (const float*)gcan, gs_b,
                         (long long)src * src, src);
    };
    const int need_r = (int)(TH * (double)src / 336 + D.th.stride + 2), need_c = (int)(TW * (double)src / 336 + D.tw.stride + 2);
    const bool fits = need_r <= PR && need_c <= PC && D.th.stride <= SMAX;
    gather_f(); gather_b();
    if (fits) { tile_f(); tile_b(); }
    CK(hipDeviceSynchronize());
    int bad_f = -1, bad_b = -1;
    if (fits) {
      std::vector<float> a(n_can), b(n_can), ga(n_img), gb(n_img);
      CK(hipMemcpy(a.data(), can_a, n_can * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), can_b, n_can * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ga.data(), gs_a, n_img * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), gs_b, n_img * 4, hipMemcpyDeviceToHost));
      bad_f = memcmp(a.data(), b.data(), n_can * 4) != 0;
      bad_b = memcmp(ga.data(), gb.data(), n_img * 4) != 0;
    }
    printf("%5d -> 336: taps/axis fwd %d, T %d | forward gather %6.2f us", src, D.th.stride, D.tth.stride, timeit(gather_f));
    if (fits) printf(", tiled %6.2f us (%s)", timeit(tile_f), bad_f ? "DIFFERENT" : "bit-identical");
    printf(" | transposed gather %6.2f us", timeit(gather_b));
    if (fits) printf(", tiled %6.2f us (%s)", timeit(tile_b), bad_b ? "DIFFERENT" : "bit-identical");
    printf("%s\n", fits ? "" : "   (patch does not fit the LDS tile)");
    advx_plan_destroy(p);
  }
  return 0;
}
