/* A host that is NOT Python and NOT torch drives the hot path through the C ABI alone (include/advx.h): plain device
 * pointers from hipMalloc, sizes, a NULL stream.  Three steps of the headline pair on a 64 x 64 image, batch 8,
 * in-kernel Philox noise, AdamW; the optimised tensor, the last pixel_values and the statistics are written to a file
 * that tests/test_gpu_cabi_c_host.py compares, byte for byte, with the same steps driven from Python.
 *
 *   hipcc -x c -o pair_steps tests/cabi/pair_steps.c -Iinclude -Ladversarialvlm_amd -ladvx_hip -Wl,-rpath,$PWD/adversarialvlm_amd
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "advx.h"

#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at line %d\n", __LINE__); return 2; } } while (0)
#define AK(x) do { if ((x) != 0) { fprintf(stderr, "advx error at line %d: %s\n", __LINE__, advx_last_error()); return 3; } } while (0)

/* tests/conftest.py: lcg_tensor(shape, salt) - exact integer arithmetic, values in [-0.5, 0.5) */
static void lcg(float* out, size_t n, uint64_t salt) {
  for (size_t i = 0; i < n; ++i) {
    uint64_t v = ((uint64_t)i * 2654435761ull + salt * 40503ull) % 4294967296ull;
    v = (v * 1664525ull + 1013904223ull) % 4294967296ull;
    out[i] = (float)((double)v / 4294967296.0 - 0.5);
  }
}

static float* to_device(const float* h, size_t n) {
  float* d = NULL;
  if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
  if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: pair_steps <output file>\n"); return 1; }
  const int H = 64, W = 64, B = 8, steps = 3;
  const size_t n = 3u * H * W;
  advx_plan_desc d;
  memset(&d, 0, sizeof(d));
  d.kind = ADVX_KIND_LLAVA; d.in_h = H; d.in_w = W; d.a0 = H; d.a1 = W;
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, sd[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  for (int c = 0; c < 3; ++c) { d.mean[c] = mean[c]; d.std[c] = sd[c]; }
  advx_plan* plan = NULL;
  AK(advx_plan_create(&d, &plan));
  AK(advx_plan_upload(plan, NULL));
  if (!advx_fused_supported(plan)) { fprintf(stderr, "plan is not fused-capable\n"); return 4; }

  float* h = (float*)malloc(sizeof(float) * n * B);
  lcg(h, n, 7);
  for (size_t i = 0; i < n; ++i) h[i] += 0.5f;                         /* x0 in [0, 1) */
  float* x0 = to_device(h, n);
  for (size_t i = 0; i < n; ++i) h[i] = 1.0f;
  float* mask = to_device(h, n);
  memset(h, 0, sizeof(float) * n);
  float *p = to_device(h, n), *m = to_device(h, n), *v = to_device(h, n), *grad = to_device(h, n);
  float *s0 = to_device(h, n), *s1 = to_device(h, n), *vbuf = to_device(h, n);
  float hstats[ADVX_STATS_N];
  memset(hstats, 0, sizeof(hstats));
  hstats[ADVX_STAT_QERR_STD] = 1e-3f;                                  /* sigma of the first step (attack_model.py:261) */
  float* stats = to_device(hstats, ADVX_STATS_N);
  const int64_t sf = advx_fused_scratch_floats(plan);
  float* scratch = NULL;
  CK(hipMalloc((void**)&scratch, sizeof(float) * sf));
  CK(hipMemset(scratch, 0, sizeof(float) * sf));                       /* the fused scratch is zero-initialised once */
  float *out = NULL, *g = NULL;
  CK(hipMalloc((void**)&out, sizeof(float) * n * B));
  if (!x0 || !mask || !p || !m || !v || !grad || !s0 || !s1 || !vbuf || !stats) return 5;

  float* sbuf[2] = {s0, s1};
  int cur = 0, prepared = 0;
  double lr = 1e-2;
  for (int t = 0; t < steps; ++t) {
    AK(advx_fused_fwd(plan, p, x0, 0.5f, B, NULL, 1, 1234u, (uint64_t)t, out, sbuf[cur], vbuf, prepared, 0, stats, scratch, NULL));
    prepared = 1;
    lcg(h, n * B, 100 + t);                                             /* the "model's" gradient of this step */
    for (size_t i = 0; i < n * B; ++i) h[i] *= 0.02f;
    if (g) CK(hipFree(g));
    g = to_device(h, n * B);
    if (!g) return 5;
    /* torch.optim.AdamW's scalars of step t + 1, in double like the Python host (lr 1e-2, betas 0.9 / 0.999, wd 0.01) */
    advx_opt_scalars o;
    o.kind = ADVX_OPT_ADAMW; o.apply = 1;
    o.lr = (float)lr; o.decay = (float)(1.0 - lr * 1e-2); o.w1 = (float)(1.0 - 0.9); o.beta2 = (float)0.999; o.w2 = (float)(1.0 - 0.999);
    o.bias2_sqrt = (float)sqrt(1.0 - pow(0.999, t + 1)); o.eps = (float)1e-8; o.neg_step_size = (float)(-(lr / (1.0 - pow(0.9, t + 1))));
    AK(advx_fused_bwd(plan, g, B, p, x0, 0.5f, 1.0f, mask, m, v, grad, &o, sbuf[1 - cur], vbuf, stats, scratch, NULL));
    cur = 1 - cur;
  }
  AK(advx_fused_flush(plan, stats, scratch, 0, NULL));
  CK(hipDeviceSynchronize());

  FILE* f = fopen(argv[1], "wb");
  if (!f) return 6;
  CK(hipMemcpy(h, p, sizeof(float) * n, hipMemcpyDeviceToHost));
  fwrite(h, sizeof(float), n, f);
  CK(hipMemcpy(h, out, sizeof(float) * n * B, hipMemcpyDeviceToHost));
  fwrite(h, sizeof(float), n * B, f);
  CK(hipMemcpy(hstats, stats, sizeof(hstats), hipMemcpyDeviceToHost));
  fwrite(hstats, sizeof(float), ADVX_STATS_N, f);
  fclose(f);
  printf("ok: %d steps, ||g|| = %.6g, sigma_next = %.6g\n", steps, hstats[ADVX_STAT_GRAD_NORM], hstats[ADVX_STAT_QERR_STD]);
  advx_plan_destroy(plan);
  return 0;
}
