/* advx.h - C ABI of libadvx_hip.so: the pixel-space hot path of the universal
 * adversarial-image PGD loop as hand-written CDNA4 (gfx950) HIP kernels.
 *
 * The reference (FusionBrainLab/AdversarialVLM) has no FFI of its own: its boundary is the
 * Python plugin API (src/processors/__init__.py:49-76, abstract_processor.py:91-208) and
 * everything below it is stock torch.  This header is the boundary a replacement of that
 * torch arithmetic binds to; each entry point cites the reference lines it replaces.
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns int32: 0 = ok, negative = ADVX_E_*; advx_last_error() gives
 *    the thread-local message.  Nothing throws across the ABI.
 *  - all data pointers are CALLER-OWNED DEVICE pointers to contiguous float32 unless
 *    stated otherwise (tensor.data_ptr()).  The library allocates device memory only
 *    inside opaque plans (tap/index tables of one geometry).
 *  - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); launches are
 *    asynchronous and never synchronise.  They can be captured in a hipGraph once the plan
 *    has been uploaded (advx_plan_upload), but a REPLAY repeats the captured kernel arguments:
 *    entry points that take per-step scalars by value (Philox offset, optimiser scalars, crop
 *    window) replay the step they were captured with.  advx_fused_fwd_sched / advx_fused_bwd_sched
 *    read those scalars from device memory and are the forms meant for replay.
 *  - image tensors are CHW, 3 channels; "stats" is a device float[ADVX_STATS_N].
 */
#ifndef ADVX_H
#define ADVX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADVX_VERSION 100 /* 0.1.0 */

#define ADVX_OK 0
#define ADVX_E_BADARG (-1)
#define ADVX_E_SHAPE (-2)
#define ADVX_E_HIP (-3)
#define ADVX_E_UNSUPPORTED (-4)

/* processor kinds = the reference's four differentiable processors */
#define ADVX_KIND_LLAVA 0   /* llavaprocessor.py:134-149   */
#define ADVX_KIND_MLLAMA 1  /* llama32processor.py:219-405 */
#define ADVX_KIND_PHI3 2    /* phi3processor.py:107-250    */
#define ADVX_KIND_QWEN2VL 3 /* qwen2VLprocessor.py:121-272 */

/* resampling modes (ATen F.interpolate semantics) */
#define ADVX_MODE_AA_BILINEAR 0
#define ADVX_MODE_BILINEAR 1
#define ADVX_MODE_BICUBIC 2

/* slots of the device-side stats vector */
#define ADVX_STAT_SIGMA 0      /* noise sigma used by the NEXT emit (= previous step's QERR_STD) */
#define ADVX_STAT_QERR_STD 1   /* std_unbiased(|q - s|)   attack_model.py:373 */
#define ADVX_STAT_QERR_MEAN 2  /* mean(|q - s|)           attack_model.py:389 */
#define ADVX_STAT_QERR_L1 3    /* sum(|q - s|)            attack_model.py:391 */
#define ADVX_STAT_IMGFIT 4     /* image_fit_loss          attack_model.py:86-106 */
#define ADVX_STAT_X_MEAN 5     /* x.mean()                attack_model.py:386 */
#define ADVX_STAT_X_STD 6      /* x.std()                 attack_model.py:387 */
#define ADVX_STAT_GRAD_NORM 7  /* ||p.grad * mask||_2     attack_model.py:340 */
#define ADVX_STATS_N 16

#define ADVX_OPT_ADAMW 0 /* torch.optim.AdamW, attack_model.py:184 */
#define ADVX_OPT_SIGN 1  /* p -= lr*sign(g): PGD-style variant named by the north star (not in the reference); |g| below FLT_MIN (1.18e-38) counts as 0 */

typedef struct advx_plan advx_plan;

typedef struct advx_plan_desc {
  int32_t kind;     /* ADVX_KIND_* */
  int32_t in_h;     /* native image height H */
  int32_t in_w;     /* native image width  W */
  /* kind-specific integers:
   *  LLAVA  : a0 = crop_size.height, a1 = crop_size.width
   *  MLLAMA : a0 = tile size, a1 = max_image_tiles
   *  PHI3   : a0 = num_crops
   *  QWEN2VL: a0 = patch_size, a1 = merge_size, a2 = temporal_patch_size,
   *           a3 = min_pixels, a4 = max_pixels */
  int64_t a0, a1, a2, a3, a4;
  float mean[3];    /* image_mean */
  float std[3];     /* image_std  */
} advx_plan_desc;

#define ADVX_MAX_STAGES 2

typedef struct advx_stage_info {
  int32_t mode;           /* ADVX_MODE_* */
  int32_t src;            /* 0 = input image, k>0 = canvas of stage k-1 */
  int32_t src_h, src_w;   /* source size */
  int32_t res_h, res_w;   /* resized size */
  int32_t can_h, can_w;   /* canvas (padded) size */
  int32_t off_y, off_x;   /* position of the resized image inside the canvas */
  float pad_value;        /* constant written outside the resized image (before normalise) */
  int32_t normalise;      /* 1: (v-mean)/std applied after padding */
  int32_t inner_axis_h;   /* 1: the inner (first) 1-D pass runs along H (phi3 transposed frame) */
} advx_stage_info;

typedef struct advx_plan_info {
  int32_t kind, in_h, in_w;
  int32_t out_rank;
  int64_t out_shape[6];     /* shape of ONE sample's pixel_values as the reference returns it */
  int64_t out_numel;        /* product of out_shape */
  int32_t n_stage;
  advx_stage_info stage[ADVX_MAX_STAGES];
  int32_t tiles_h, tiles_w; /* MLLAMA / PHI3 local tile arrangement */
  int32_t num_tiles;        /* MLLAMA: real tiles; PHI3: 1 + local tiles; QWEN: grid_h*grid_w */
  int32_t grid_h, grid_w;   /* QWEN patch grid */
  int32_t image_h, image_w; /* PHI3 image_sizes[0] */
  int32_t num_img_tokens;   /* PHI3 (phi3processor.py:244) */
  int32_t aspect_ratio_id;  /* MLLAMA (1-based id of the tile arrangement) */
  int64_t workspace_floats; /* floats of scratch advx_emit / advx_collect need */
} advx_plan_info;

/* ---------------------------------------------------------------- library */
int32_t advx_version(void);
const char* advx_last_error(void);

/* Development switch.  ADVX_TUNE_GENERIC_KERNELS = 1 makes every call take the general kernels (run-time blur
 * radius, one launch per operation) instead of the specialised / merged ones; results are bit-identical, which
 * is what the tests use it for.  Process-wide, not thread-safe. */
#define ADVX_TUNE_RESET_ALL 0         /* every switch back to its default, whatever `value` is */
#define ADVX_TUNE_GENERIC_KERNELS 1
#define ADVX_TUNE_PAIR_LEAN 5       /* experiment (round 4): the float32 Philox pair re-derives s, v in the forward; the backward stores neither
                                     them nor grad_p - measured, not the default */
#define ADVX_TUNE_XCD_MAP 6         /* 1 (default): the grids of the B x P_out writers have gx padded to a multiple of 8, so that all batch
                                     slices of a column block run on one XCD (one L2 fetches the shared canvas / v); 0: rounds 1-3 */
#define ADVX_TUNE_BWD_XCD 7         /* 1 (default): the readers of B x P_out (advx_fused_bwd, the batch reductions) map their workgroups to column
                                     blocks as the writers do under ADVX_TUNE_XCD_MAP (measured: k_fused_bwd 21.0 -> 19.6 us); 0: rounds 1-3 */
#define ADVX_TUNE_IMG_XCD 9         /* rows per group (default 8) of the XCD-aware grids of the image-sized gathers: launched 1-D, groups of that many
                                     rows of workgroups are dealt to the 8 XCDs in turn, so that one XCD's L2 fetches the source rows of its groups only;
                                     0: (column chunk, row, layer) grids dealt round-robin workgroup by workgroup (rounds 1-3).  Same results */
#define ADVX_TUNE_DIRECT_BATCH 15   /* 1 (default): advx_collect_update sums one or two prompts of a plain float32 plan inside its gather; 0: batch reduction first */
#define ADVX_TUNE_COLLECT_UPDATE 14 /* images of >= value * 1000 positions (default 1) are offered advx_collect_update; 0: none (the two calls) */
#define ADVX_TUNE_TAIL3 13          /* 1 (default): the prepared chain's image kernels (prepare, tail, update) handle the three channels of a pixel in
                                     one thread on a (chunk, row) grid for images of 250 k positions and more; 0: one thread per element (rounds
                                     1-3).  Same per-pixel results; the statistics / ||g|| partials are summed over another partition */
#define ADVX_TUNE_BLUR_THREADS 12   /* 512 (default) / 256: threads per 32 x 32 tile of the merged blur backward (radius <= 4).  The per-pixel
                                     results are the same; the ||g|| partial of a tile is summed in another order */
#define ADVX_TUNE_HEAD3 10          /* canvases of >= value * 1000 positions (default 50) are resized by the three-channel windowed forward;
                                     0: one thread per (channel, position) whatever the size (rounds 1-3).  Same results */
#define ADVX_TUNE_ROW_BATCH 8       /* 1 (default): the gathers of the plans' resizes (advx_emit_multi / advx_forward_multi, advx_collect*, the prepared
                                     chain's tail, the crop window's adjoint) run as compiled windows - every load of a thread in flight before the
                                     first use - where their tables' rows have <= 4 taps (<= 6: the composed crop window), a window row at a time
                                     up to 10; 0: one memory round trip per tap (rounds 1-3).  Same results bit for bit */
#define ADVX_TUNE_PAIR_NT_LOADS 2   /* advx_fused_bwd reads grad_out with non-temporal loads (same results) */
#define ADVX_TUNE_SEPARATE_CROP 4   /* 1: never compose a crop window with a plan's stage 0 (advx_forward_multi) - the window is resized
                                     * into `argument` and the plan resamples that, two launches each way, bit-identical to the
                                     * unfused kernels; 2: compose wherever the tables fit, also where it does not pay (tests);
                                     * 0 (default): compose where it was measured to pay - one-stage plans whose antialiased stage 0
                                     * does not up-sample and whose canvas has one gradient image (LLaVA; Llama-3.2-Vision from
                                     * images larger than its canvas) */
#define ADVX_TUNE_FULL_TAP_ROWS 3   /* plans uploaded from now on keep ATen's full tap rows on the device; by default the
                                     * device copies drop the zero-weight taps at the ends of a row (same results) */
int32_t advx_set_tuning(int32_t what, int32_t value);

/* ------------------------------------------------------------------ plans
 * A plan holds the integer geometry and the float32 tap tables of one
 * (H, W) -> processor layout, computed on the HOST exactly as ATen / the reference do
 * (llama32processor.py:255-279, qwen2VLprocessor.py:176-197, phi3processor.py:173-216).
 * Creation needs no GPU; advx_plan_upload copies the tables to the current device. */
int32_t advx_plan_create(const advx_plan_desc* desc, advx_plan** out);
int32_t advx_plan_destroy(advx_plan* plan);
int32_t advx_plan_describe(const advx_plan* plan, advx_plan_info* info);
int32_t advx_plan_upload(advx_plan* plan, void* stream);
/* Host copy of one tap table (tests). transposed=0: per OUTPUT index the taps into the
 * source; transposed=1: per SOURCE index the taps into the output (backward gather).
 * Call with start=NULL to query n / stride only. axis: 0 = H, 1 = W. */
int32_t advx_plan_taps(const advx_plan* plan, int32_t stage, int32_t axis, int32_t transposed,
                       int32_t* n, int32_t* stride, int32_t* start, int32_t* count, float* weight);
/* Host evaluation of the layout map (tests): flat output indices that canvas element
 * (c, y, x) of `stage` is written to; returns how many (0, 1 or 2) in *n_idx. */
int32_t advx_plan_out_index(const advx_plan* plan, int32_t stage, int32_t c, int32_t y, int32_t x,
                            int32_t* n_idx, int64_t idx[2]);
/* [lo, hi): flat indices of one sample that the plan's emits cover.  What lies outside is the
 * constant padding (zero tiles) of llama32processor.py:344-346 / phi3processor.py:232-235. */
int32_t advx_plan_live_range(const advx_plan* plan, int64_t* lo, int64_t* hi);
/* Boundary dtype of the plan's two B x out_numel tensors (`out` of advx_emit*, advx_prepared_fwd and
 * `grad_out` of advx_collect, advx_prepared_bwd*): ADVX_IO_F32 (default, what the reference hands
 * the model), ADVX_IO_F16 or ADVX_IO_BF16 = the model's own dtype - pixel_values are rounded once
 * (the cast the vision tower applies first anyway), the half gradient is widened on load, all
 * arithmetic stays fp32.  With a half dtype the `float*` parameters of those entry points point at
 * half data; `unit_noise` stays float32.  Needs out_numel % 4 == 0. */
int32_t advx_plan_set_io(advx_plan* plan, int32_t io_dtype);
int32_t advx_plan_get_io(const advx_plan* plan);
/* Host tap computation for an arbitrary 1-D resize (tests; also the crop window's tables).  flags: a sum of
 *   ADVX_TAPS_TRANSPOSED   the table of the adjoint (rows = source indices);
 *   ADVX_TAPS_DEVICE_ROWS  the rows as advx_plan_upload stores them: zero-weight taps at either end of a row dropped;
 *   ADVX_TAPS_BUILDER      (with TRANSPOSED) formed the way the device-side builder of the crop window forms them. */
#define ADVX_TAPS_TRANSPOSED 1
#define ADVX_TAPS_DEVICE_ROWS 2
#define ADVX_TAPS_BUILDER 4
int32_t advx_taps_compute(int32_t mode, int32_t in_size, int32_t out_size, int32_t flags,
                          int32_t* n, int32_t* stride, int32_t* start, int32_t* count, float* weight);

/* ------------------------------------------------ processor level (plugin API)
 * advx_emit  = Differentiable*ImageProcessor.process() + `repeat(B,..)` + `randn*sigma`
 *              (attack_model.py:314-321): argument [3,H,W] -> out [B, out_numel].
 *   unit_noise : optional N(0,1) tensor [B, out_numel] (parity mode); if NULL and
 *                use_philox != 0 the noise is generated in-kernel (Philox4x32-10 +
 *                Box-Muller; key = seed, element 4q+k of row b = k-th output of the block
 *                with counter (q, b, offset_lo, offset_hi); oracle/philox.py restates it);
 *                if both are off no noise is added.
 *   sigma_dev  : device pointer to the noise sigma (stats + ADVX_STAT_SIGMA), may be NULL
 *                when no noise is requested.
 * advx_collect = backward of the above: grad_out [B, out_numel] -> grad_argument [3,H,W]
 *              (sum over B, un-tile, /std, drop padding, transposed resize).
 *   accumulate != 0 adds into grad_argument instead of overwriting. */
int32_t advx_emit(advx_plan* plan, const float* argument, int32_t batch, const float* sigma_dev,
                  const float* unit_noise, int32_t use_philox, uint64_t seed, uint64_t offset,
                  float* out, float* workspace, int64_t workspace_floats, void* stream);
/* The constant padding tiles of Mllama / Phi-3.5 (llama32processor.py:344-346,
 * phi3processor.py:232-235) are 3/4 resp. 2/7 of `out`.  The reference adds noise to them too
 * (randn_like of the whole tensor, attack_model.py:320).  Keeping them zero is a DEVIATION, not a free
 * optimisation: Phi-3.5 drops the crops beyond image_sizes, but the Llama-3.2 vision encoder masks only
 * padding-to-padding attention, so the real tile's tokens attend to the padding tiles and the loss depends
 * (weakly) on their pixels - tests/test_mllama_padding_visibility.py shows it on a random model.
 *   ADVX_PAD_NOISE : advx_emit's behaviour - the whole tensor is written, padding = 0 + noise.
 *   ADVX_PAD_KEEP  : only the elements the plan's emits cover are written; the caller keeps
 *                    `out` across steps and has zeroed its padding once (padding stays exactly
 *                    0, what process() itself returns).  Noise of the covered elements is the
 *                    same as with ADVX_PAD_NOISE (same counters). */
#define ADVX_PAD_NOISE 0
#define ADVX_PAD_KEEP 1
int32_t advx_emit_ex(advx_plan* plan, const float* argument, int32_t batch, const float* sigma_dev,
                     const float* unit_noise, int32_t use_philox, uint64_t seed, uint64_t offset,
                     float* out, float* workspace, int64_t workspace_floats, int32_t pad_mode,
                     void* stream);
int32_t advx_collect(advx_plan* plan, const float* grad_out, int32_t batch, float* grad_argument,
                     int32_t accumulate, float* workspace, int64_t workspace_floats, void* stream);

/* Cross-model runs (crossattack_models.py:352-391): n <= 4 processors over ONE image.  The result
 * is, bit for bit, what n calls of advx_emit_ex / advx_collect (the first with `accumulate`, the
 * others adding) give; the resize kernels of all plans that read the image run in one launch each
 * way and the plans' image gradients are summed left to right inside it, without read-modify-write
 * passes; so do the emits, and the batch reductions when the plans read their gradients the same way.  Arrays have n entries, host memory; every plan needs its own workspace; plan i's noise
 * stream is (seed, offsets[i]); unit_noises may be NULL or hold NULL entries; `outs` / `grad_outs`
 * carry each plan's boundary dtype (advx_plan_set_io). */
int32_t advx_emit_multi(int32_t n, advx_plan* const* plans, const float* argument, const int32_t* batches,
                        const float* sigma_dev, const float* const* unit_noises, int32_t use_philox,
                        uint64_t seed, const uint64_t* offsets, void* const* outs,
                        float* const* workspaces, const int64_t* workspace_floats, int32_t pad_mode,
                        void* stream);
int32_t advx_collect_multi(int32_t n, advx_plan* const* plans, const void* const* grad_outs,
                           const int32_t* batches, float* grad_argument, int32_t accumulate,
                           float* const* workspaces, const int64_t* workspace_floats, void* stream);

/* advx_image_fwd followed by advx_emit_multi in one call (the forward half of a step of either trainer
 * whenever the fused / prepared chains do not apply): same tensors; without a crop window the
 * statistics of the image are reduced by one block of the plans' resize launch instead of a launch of
 * their own; with one, the window's transposed tap tables (read only by advx_image_bwd*) are built by the first
 * workgroups of the emit launch instead of the step's first image kernel.  The emits read their sigma from
 * stats[ADVX_STAT_SIGMA].  Arguments as for the two calls.
 *
 * Composed crop.  With ONE plan whose stage 0 reads the image and a window for which advx_crop_composes() is 1, the
 * window's resize back to H x W (attack_model.py:309-310) and the plan's own resize (process(), :314) are applied as
 * ONE table per axis, w_C[y][j] = sum_k w_plan[y][k] * w_window[k][j] (built on the device, per step): one gather
 * image -> canvas instead of two launches around an image-sized intermediate.  `argument` is then NOT written, and
 * the backward of that step is advx_collect_crop (one transposed gather canvas -> image) followed by
 * advx_image_bwd* WITHOUT a crop window.  The composed map drops the float32 rounding of the intermediate image: held
 * to the reference at 1e-4 (tests), not bit-identical to the two-launch form; ADVX_TUNE_SEPARATE_CROP (or ADVX_TUNE_GENERIC_KERNELS) switches it off. */
int32_t advx_crop_composes(const advx_plan* plan, int32_t H, int32_t W, const int32_t* crop_ijhw);
/* row lengths (upper bounds) of the composed tables per axis (0 = H, 1 = W); host only, for the tests */
int32_t advx_crop_compose_strides(const advx_plan* plan, int32_t H, int32_t W, const int32_t* crop_ijhw, int32_t forward[2],
                                  int32_t transposed[2]);
/* Host-side view of the XCD-aware grids of the image-sized launches (tests): the number of workgroups a (gx, gy, gz) grid with
 * `riders` extra blocks is launched with under the current ADVX_TUNE_IMG_XCD and, if `logical` is given (room for that many), the
 * logical block of every physical one: x fastest, then y, then z; riders behind them; -1 = padding. */
int32_t advx_image_grid_map(int32_t gx, int32_t gy, int32_t gz, int32_t riders, int32_t* launch_blocks, int32_t* logical);

/* The composed tables' REAL longest rows for this window (forward, transposed; the larger of the two axes each): the gathers
 * pick their compiled windows by these (rows of up to 6 taps), the strides above only size the tables.  Host arithmetic. */
int32_t advx_crop_compose_rows(const advx_plan* plan, int32_t H, int32_t W, const int32_t* crop_ijhw, int32_t* forward,
                               int32_t* transposed);
int32_t advx_collect_crop(advx_plan* plan, const void* grad_out, int32_t batch, float* grad_s, int32_t accumulate,
                          float* workspace, int64_t workspace_floats, int32_t H, int32_t W, const int32_t* crop_ijhw,
                          float* image_scratch, void* stream);
int32_t advx_forward_multi(const float* p, const float* x0, int32_t H, int32_t W, float epsilon,
                           int32_t blur_kernel, float blur_sigma, const int32_t* crop_ijhw, float* s,
                           float* argument, float* stats, float* image_scratch, int32_t n,
                           advx_plan* const* plans, const int32_t* batches, const float* const* unit_noises,
                           int32_t use_philox, uint64_t seed, const uint64_t* offsets, void* const* outs,
                           float* const* workspaces, const int64_t* workspace_floats, int32_t pad_mode,
                           void* stream);

/* ---------------------------------------------------- image level (trainer)
 * advx_image_fwd : attack_model.py:300-312,329,366-373,386-391
 *     x = eps*tanh(p) ; optional Gaussian blur (kernel k, sigma) ; s = x0 + x ;
 *     optional random-resized-crop window (i,j,h,w) resized back to (H,W) -> argument.
 *     Also reduces image_fit_loss, the quantise-error statistics and x mean/std into
 *     `stats`, and rotates stats[SIGMA] <- old stats[QERR_STD] before overwriting it.
 *     blur_k = 0 disables blur; crop = NULL disables the crop (then argument may alias s).
 *   scratch: float[advx_image_scratch_floats(H,W,blur_k)].
 * advx_image_bwd : the autograd of the above (attack_model.py:332): grad_argument ->
 *     grad wrt p, plus imgfit_scale * d(image_fit_loss)/dp; written (or accumulated)
 *     into grad_p UNMASKED (the mask is applied by advx_update, attack_model.py:336).
 *     Called with the scratch, stream and crop window of the preceding advx_image_fwd it reuses
 *     the crop's tap tables that call left in `scratch` (any other combination rebuilds them):
 *     leave `scratch` alone between the two calls of a step. */
int64_t advx_image_scratch_floats(int32_t H, int32_t W, int32_t blur_k);
int32_t advx_image_fwd(const float* p, const float* x0, int32_t H, int32_t W, float epsilon,
                       int32_t blur_k, float blur_sigma, const int32_t* crop_ijhw,
                       float* s, float* argument, float* stats, float* scratch, void* stream);
int32_t advx_image_bwd(const float* p, const float* s, const float* grad_argument, int32_t H, int32_t W,
                       float epsilon, int32_t blur_k, float blur_sigma, const int32_t* crop_ijhw,
                       float imgfit_scale, float* grad_p, int32_t accumulate,
                       float* scratch, void* stream);

/* advx_update : attack_model.py:335-346 : g = grad_p * mask ; ||g|| -> stats ; optimiser.
 *   ADAMW follows torch.optim.AdamW's single-tensor arithmetic with the scalars the host
 *   derives in double exactly like torch (decay = 1-lr*wd, step_size = lr/(1-b1^t), ...).
 *   apply = 0 only masks + measures the norm (gradient-accumulation iterations). */
typedef struct advx_opt_scalars {
  int32_t kind;         /* ADVX_OPT_* */
  int32_t apply;        /* 1 = take the optimiser step */
  float lr;             /* SIGN: step length */
  float decay;          /* ADAMW: 1 - lr*weight_decay */
  float w1;             /* ADAMW: 1 - beta1 (lerp weight) */
  float beta2;          /* ADAMW */
  float w2;             /* ADAMW: 1 - beta2 */
  float bias2_sqrt;     /* ADAMW: sqrt(1 - beta2^t) */
  float eps;            /* ADAMW */
  float neg_step_size;  /* ADAMW: -(lr / (1 - beta1^t)) */
} advx_opt_scalars;
int32_t advx_update(float* p, float* m, float* v, float* grad_p, const float* mask, int64_t n,
                    const advx_opt_scalars* opt, float* stats, float* scratch, void* stream);

/* advx_image_bwd followed by advx_update in one call, for when nothing sits between them (one rank,
 * or a step inside a gradient-accumulation window): same results, with the tanh backward - and,
 * without blur, the crop window's transposed resize - inside the optimiser's launch.
 * image_scratch / update_scratch as for advx_image_bwd / advx_update. */
int32_t advx_image_bwd_update(float* p, const float* s, const float* grad_argument, int32_t H, int32_t W,
                              float epsilon, int32_t blur_kernel, float blur_sigma, const int32_t* crop_ijhw,
                              float imgfit_scale, float* p_grad, int32_t accumulate, const float* mask,
                              float* m, float* v, const advx_opt_scalars* opt, float* stats,
                              float* image_scratch, float* update_scratch, int32_t finalize_norm, void* stream);

/* advx_collect / advx_collect_crop AND advx_image_bwd_update (no blur) in one call: batch reduction, the upper stages, then ONE
 * launch that gathers the transposed resize of stage 0 (through the composed table when `crop_ijhw` is given, as advx_collect_crop)
 * for the three channels of an image pixel and applies image-fit', tanh', [accumulate], mask, ||g|| partial and the optimiser to
 * them - the two calls' arithmetic, one launch and one round trip of the image gradient less (attack_model.py:332-346).
 * advx_collect_update_supported: whether this step can take it (one plan whose stage 0 reads the image,
 * transposed rows of <= 4 taps (<= 6 for a plan with one canvas copy); crop_ijhw NULL or a composing window); the call
 * returns ADVX_E_UNSUPPORTED otherwise and the caller takes the two calls.  finalize_norm as advx_image_bwd_update's. */
int32_t advx_collect_update_supported(advx_plan* plan, int32_t H, int32_t W, const int32_t* crop_ijhw);
int32_t advx_collect_update(advx_plan* plan, const void* grad_out, int32_t batch, float* ws, int64_t ws_floats, int32_t H, int32_t W,
                            const int32_t* crop_ijhw, float* image_scratch, float* p, const float* s, float eps, float imgfit_scale,
                            float* grad_p, int32_t accumulate, const float* mask, float* m, float* v, const advx_opt_scalars* opt,
                            float* stats, float* update_scratch, int32_t finalize_norm, void* stream);
/* finalize_norm == 0 leaves stats[ADVX_STAT_GRAD_NORM] to a later advx_update_flush (same n, same
 * update_scratch, before the next update overwrites its partial sums): a training loop that logs every
 * k-th step skips the one-block reduction on the others. */
int32_t advx_update_flush(int64_t n, float* stats, float* update_scratch, void* stream);
int64_t advx_update_scratch_floats(int64_t n);

/* ------------------------------------------- step-to-step fusion of the blur chains (round 4)
 * The reference's production presets blur every step (attack_clamp_tanh_llava_gblur.sh:24-60: kernel 9, a crop window
 * per step).  advx_image_step is advx_image_bwd_update of step t (blur, `grad_argument` = gradient w.r.t. the image s, as
 * advx_collect* / advx_collect_crop leave it; no crop window here) AND the first image kernel of step t+1's
 * advx_forward_multi - eps*tanh of the UPDATED p, blur with next_blur_sigma, s_next = x0 + blur, its statistics partials,
 * and with next_crop_ijhw the forward rows of (window o next_plan's stage 0) - in ONE launch: every 32 x 32 tile recomputes
 * the gradient and the optimiser update on its r-halo (same inputs, same order: the owner's bits), only the owner stores.
 * attack_model.py:300-304 of iteration t+1 thereby move behind :335-346 of iteration t; the values do not change.
 *   p, m, v      : state of step t (read only);  p_out, m_out, v_out : state of step t+1 - OTHER buffers (a neighbouring tile
 *                  must read the old p whenever the owner's block runs); the caller ping-pongs.  m / v may be NULL for ADVX_OPT_SIGN.
 *   s            : image of step t (read, with halos);  s_next : image of step t+1, another buffer.
 *   next_crop_ijhw / next_plan : NULL, or a window with advx_crop_composes(next_plan, H, W, window) == 1.
 * Supported (advx_image_step_supported): odd kernel sizes 3..9, min(H, W) >= 32 + 3r + 2, opt->apply == 1.
 * The next forward is advx_forward_multi_ready (same H, W, blur_kernel, window, plans, image_scratch): the plans' resizes -
 * which also reduce the statistics and build the composed tables' transposed rows - and the emits; `s` = s_next above.
 * Bit for bit the results of advx_image_bwd_update + advx_forward_multi. */
int32_t advx_image_step_supported(int32_t H, int32_t W, int32_t blur_kernel);
int32_t advx_image_step(const float* p, const float* m, const float* v, float* p_out, float* m_out, float* v_out,
                        const float* s, const float* grad_argument, int32_t H, int32_t W, float epsilon,
                        int32_t blur_kernel, float blur_sigma, float imgfit_scale, float* p_grad, const float* mask,
                        const advx_opt_scalars* opt, float* image_scratch, float* update_scratch, const float* x0,
                        float next_blur_sigma, const int32_t* next_crop_ijhw, advx_plan* next_plan, float* s_next,
                        void* stream);
int32_t advx_forward_multi_ready(int32_t H, int32_t W, int32_t blur_kernel, const int32_t* crop_ijhw, float* s, float* stats,
                                 float* image_scratch, int32_t n, advx_plan* const* plans, const int32_t* batches,
                                 const float* const* unit_noises, int32_t use_philox, uint64_t seed,
                                 const uint64_t* offsets, void* const* outs, float* const* workspaces,
                                 const int64_t* workspace_floats, int32_t pad_mode, void* stream);

/* ------------------------------------------- fused fast path (headline config)
 * The whole owned step of attack_model.py:300-346,366-373 for a plan whose resize is the
 * identity (LLaVA at native 336x336), no blur, no crop, no gradient accumulation, as a
 * software-pipelined pair of launches:
 *   advx_fused_fwd : out[B, 3*H*W] = v + sigma * N(0,1), v = (x0+eps*tanh(p)-mean)/std and
 *                    sigma = stats[QERR_STD] (quantise error of the PREVIOUS image, as in
 *                    attack_model.py:320,373).  v (v_buf) and s = x0+eps*tanh(p) (s_buf) are
 *                    PREPARED data: written by the previous advx_fused_bwd, or by this call
 *                    itself when prepared == 0 (first step / p changed elsewhere; one extra
 *                    small launch).  Leaves the statistics partials of s.
 *   advx_fused_bwd : reduces those partials (SIGMA <- old QERR_STD, QERR_STD <- new, slots 2..6),
 *                    then grad_out[B,3*H*W] -> sum_b, /std, +imgfit', tanh', mask, ||g||,
 *                    optimiser, and from the UPDATED p the next step's s (s_next) and v (v_buf).
 *                    opt == NULL: gradient only (world_size > 1: all-reduce grad_p, then
 *                    advx_update; the next advx_fused_fwd must then pass prepared = 0).
 * The gradient-norm reduction rides in block (0,0) of the next advx_fused_fwd, so a step is
 * two launches; slot 7 lags until then or until advx_fused_flush.  After advx_fused_bwd of
 * step t every other slot refers to step t.  `scratch` must be ZERO-INITIALISED once and kept
 * for the life of the loop.  advx_fused_flush reduces the pending gradient norm and, with
 * image_too != 0 (only meaningful between a fused_fwd and its fused_bwd), the image statistics. */
int32_t advx_fused_supported(const advx_plan* plan);
/* use_philox: 0 = no in-kernel noise, 1 = Philox addressed (float4 column, batch row, offset) like every other chain,
 * ADVX_PHILOX_STEP_CHAIN = addressed as advx_fused_step addresses it (pixel, group of four batch rows, offset): the
 * forward the one-launch chain runs on its own - first step, first step after a resume - then draws exactly what the
 * chain's own emission would have drawn (float32 boundary only). */
#define ADVX_PHILOX_STEP_CHAIN 2
int32_t advx_fused_fwd(advx_plan* plan, const float* p, const float* x0, float epsilon, int32_t batch,
                       const float* unit_noise, int32_t use_philox, uint64_t seed, uint64_t offset,
                       float* out, float* s_buf, float* v_buf, int32_t prepared,
                       int32_t parity /* row set that receives the statistics rows: 0 for the pair */,
                       float* stats, float* scratch, void* stream);
int32_t advx_fused_bwd(advx_plan* plan, const float* grad_out, int32_t batch, float* p, const float* x0,
                       float epsilon, float imgfit_scale, const float* mask, float* m, float* v,
                       float* grad_p, const advx_opt_scalars* opt, float* s_next, float* v_buf,
                       float* stats, float* scratch, void* stream);
/* Data-parallel tail of the pair: after advx_fused_bwd(opt = NULL) and the all-reduce of
 * grad_p, ONE launch masks the gradient, leaves the ||g|| partials (reduced by the next
 * advx_fused_fwd / advx_fused_flush), takes the optimiser step and prepares s / v of the next
 * forward - which may then be called with prepared = 1. */
int32_t advx_fused_update(advx_plan* plan, float* p, float* m, float* v, float* grad_p, const float* mask,
                          const float* x0, float epsilon, const advx_opt_scalars* opt, float* s_next,
                          float* v_buf, float* scratch, void* stream);
/* The same pair with pixel_values written, and grad_out read, in the VLM's own dtype.  The
 * reference hands fp32 pixel_values to a half-precision model whose first op casts them
 * (attack_model.py:326-333, the vision tower's patch embedding runs in model.dtype), and
 * autograd casts the half gradient back to fp32 on the way out; io_dtype = F16 / BF16 makes
 * both round-to-nearest-even casts part of the pair, so `out` equals the fp32 result cast once
 * and the fp32 copy (2 x 4 x B x P bytes) never touches HBM.  All arithmetic stays fp32. */
#define ADVX_IO_F32 0
#define ADVX_IO_F16 1
#define ADVX_IO_BF16 2
int32_t advx_fused_fwd_io(advx_plan* plan, const float* p, const float* x0, float epsilon, int32_t batch,
                          const float* unit_noise, int32_t use_philox, uint64_t seed, uint64_t offset,
                          void* out, int32_t io_dtype, float* s_buf, float* v_buf, int32_t prepared,
                          int32_t parity, float* stats, float* scratch, void* stream);
int32_t advx_fused_bwd_io(advx_plan* plan, const void* grad_out, int32_t io_dtype, int32_t batch, float* p,
                          const float* x0, float epsilon, float imgfit_scale, const float* mask, float* m,
                          float* v, float* grad_p, const advx_opt_scalars* opt, float* s_next, float* v_buf,
                          float* stats, float* scratch, void* stream);
int64_t advx_fused_scratch_floats(const advx_plan* plan);
int32_t advx_fused_flush(advx_plan* plan, float* stats, float* scratch, int32_t image_too, void* stream);

/* hipGraph replay of the pair.  A captured launch cannot be handed a new Philox offset or new optimiser scalars,
 * so these two forms read them from DEVICE memory: `sched` = advx_sched_bytes(n) bytes, filled on the host by
 * advx_sched_fill (step counters + the optimiser scalars of steps first_step .. first_step + n - 1, derived by
 * the host in double exactly as for advx_fused_bwd) and copied to the device by the caller.  The forward uses
 * offset = offset_base + step, the backward table entry step - first_step, and the pair advances the counter
 * itself (no atomics: each kernel reads one word and writes the other; kernel boundaries order them).  Results
 * are those of advx_fused_fwd_io (prepared = 1, in-kernel noise) / advx_fused_bwd_io with the same scalars, bit
 * for bit.  Capture: advx_fused_bwd_sched(t), advx_fused_fwd_sched(t+1), advx_fused_bwd_sched(t+1),
 * advx_fused_fwd_sched(t+2) with the two s buffers alternating, then replay (tests/test_gpu_graph.py). */
int64_t advx_sched_bytes(int32_t n_opt);
int32_t advx_sched_fill(void* host_buf, int32_t n_opt, const advx_opt_scalars* table, uint64_t first_step);
int32_t advx_fused_fwd_sched(advx_plan* plan, const float* p, const float* x0, float epsilon, int32_t batch,
                             uint64_t seed, uint64_t offset_base, void* out, int32_t io_dtype, float* s_buf,
                             float* v_buf, float* stats, float* scratch, void* sched, void* stream);
int32_t advx_fused_bwd_sched(advx_plan* plan, const void* grad_out, int32_t io_dtype, int32_t batch, float* p,
                             const float* x0, float epsilon, float imgfit_scale, const float* mask, float* m,
                             float* v, float* grad_p, int32_t opt_kind, float* s_next, float* v_buf, float* stats,
                             float* scratch, void* sched, void* stream);

/* One launch per step (single GPU): advx_fused_step = advx_fused_bwd(step t) followed by
 * advx_fused_fwd(step t+1) for the same pixels inside one kernel - grad_out of step t in,
 * pixel_values of step t+1 out - so that the gradient stream of one workgroup overlaps the
 * noise generation and store stream of another.  The loop is
 *     advx_fused_fwd (step 0)  ->  VLM  ->  advx_fused_step  ->  VLM  ->  advx_fused_step ...
 * Partial-reduction rows are double-buffered in `scratch`: `parity` names the set that holds
 * the statistics rows of the CURRENT image (0 after advx_fused_fwd; flips after every step),
 * image_rows_in / norm_rows_in their counts (advx_fused_step_rows; norm_rows_in = 0 on the
 * first step).  The noise of step t+1 uses offset_next.  After the call `stats` describes
 * step t except slot 7, which holds ||g_{t-1}|| until the next step or advx_fused_step_flush
 * (parity / norm_rows = the values to be passed to the NEXT advx_fused_step). */
int32_t advx_fused_step(advx_plan* plan, const float* grad_out, int32_t batch, float* p, const float* x0,
                        float epsilon, float imgfit_scale, const float* mask, float* m, float* v,
                        float* grad_p, const advx_opt_scalars* opt, const float* unit_noise_next,
                        int32_t use_philox, uint64_t seed, uint64_t offset_next, float* out_next,
                        float* s_next, float* v_buf, int32_t parity, int32_t image_rows_in,
                        int32_t norm_rows_in, float* stats, float* scratch, void* stream);
int32_t advx_fused_step_rows(const advx_plan* plan, int32_t* rows_after_fwd, int32_t* rows_after_step);
int32_t advx_fused_step_flush(advx_plan* plan, int32_t parity, int32_t norm_rows, float* stats,
                              float* scratch, void* stream);

/* ------------------------------ prepared chain: the same pipelining for plans that DO resample
 * Any single plan (LLaVA from a non-native image such as the reference's 512x512 gray.png,
 * Mllama, Qwen2-VL; Phi-3.5, whose second stage resamples the first canvas, with one extra
 * launch each way) without blur, crop or gradient accumulation: the backward of step t leaves
 * s_{t+1}, its statistics partials and the processed canvas of step t+1 behind, so a step of
 * attack_model.py:300-346,366-373 is four launches (emit | batch-reduce, tail, head) instead of
 * the seven to nine of advx_image_fwd + advx_emit + advx_collect + advx_image_bwd_update.
 *   advx_prepared_fwd : out[B, out_numel] = canvas (+ sigma*noise), sigma = stats[QERR_STD] of the
 *                       previous image; prepared == 0 (first step / p changed elsewhere) first
 *                       builds s (s_buf), its partials (row set `parity`) and the canvas.
 *   advx_prepared_bwd : grad_out -> sum_b -> resize^T, /std, image-fit', tanh', mask, ||g||,
 *                       optimiser -> p; s_next = x0 + eps*tanh(p_new); reduces the statistics of
 *                       the CURRENT image from row set `parity` (rows_in rows; rotates SIGMA <-
 *                       QERR_STD first), leaves those of s_next in the other set, the next
 *                       canvas in the workspace and the gradient norm in slot 7.  After it every
 *                       slot refers to the step just taken.
 * Row counts: advx_prepared_rows (after a preparing forward / after a backward).  `workspace` is
 * the plan workspace (advx_plan_describe), kept by the caller across steps; `scratch`
 * (advx_prepared_scratch_floats) needs no initialisation. */
int32_t advx_prepared_supported(const advx_plan* plan);
int64_t advx_prepared_scratch_floats(const advx_plan* plan);
int32_t advx_prepared_rows(const advx_plan* plan, int32_t* rows_after_prepare, int32_t* rows_after_bwd);
int32_t advx_prepared_fwd(advx_plan* plan, const float* p, const float* x0, float epsilon, int32_t batch,
                          const float* unit_noise, int32_t use_philox, uint64_t seed, uint64_t offset,
                          float* out, float* s_buf, int32_t prepared, int32_t parity, float* stats,
                          float* scratch, float* workspace, int64_t workspace_floats, int32_t pad_mode,
                          void* stream);
int32_t advx_prepared_bwd(advx_plan* plan, const float* grad_out, int32_t batch, float* p, const float* x0,
                          float epsilon, float imgfit_scale, const float* mask, float* m, float* v,
                          float* grad_p, const advx_opt_scalars* opt, float* s_next, int32_t rows_in,
                          int32_t parity, float* stats, float* scratch, float* workspace,
                          int64_t workspace_floats, void* stream);

/* Data-parallel forms of advx_prepared_bwd: the tail is split around the all-reduce of the image
 * gradient.  advx_prepared_bwd_grad leaves this rank's UNMASKED gradient in grad_p (and reduces
 * the current image's statistics); after the caller's all-reduce (e.g. RCCL) advx_prepared_update
 * masks it, takes the optimiser step and prepares the next step.  advx_prepared_bwd_dp does both
 * around the peer exchange of `comm` (below) in one call. */
int32_t advx_prepared_bwd_grad(advx_plan* plan, const float* grad_out, int32_t batch, const float* p, const float* x0,
                               float epsilon, float imgfit_scale, float* grad_p, int32_t rows_in, int32_t parity,
                               float* stats, float* scratch, float* workspace, int64_t workspace_floats,
                               void* stream);
int32_t advx_prepared_update(advx_plan* plan, float* p, float* m, float* v, float* grad_p, const float* mask,
                             const float* x0, float epsilon, const advx_opt_scalars* opt, float* s_next,
                             int32_t parity, float* stats, float* scratch, float* workspace,
                             int64_t workspace_floats, void* stream);
struct advx_comm;
int32_t advx_prepared_bwd_dp(advx_plan* plan, struct advx_comm* comm, const float* grad_out, int32_t batch, float* p,
                             const float* x0, float epsilon, float imgfit_scale, const float* mask, float* m,
                             float* v, const advx_opt_scalars* opt, float* s_next, int32_t rows_in,
                             int32_t parity, float* stats, float* scratch, float* workspace,
                             int64_t workspace_floats, double timeout_s, void* stream);

/* -------------------------------------------------- data-parallel exchange
 * The reference has no data parallelism (one process, one model per GPU:
 * crossattack_models.py:244-258); the exchange added here is ONE all-reduce(sum) per step of the
 * shared image gradient (P_in floats, 1.35 MB at 336x336), each rank having pre-scaled its
 * share.  Two transports behind the same call sites: RCCL (torch.distributed, host side) and
 * this peer all-reduce, which needs no host library at all:
 *   - every rank creates ONE exchange segment in uncached device memory
 *     [flags | send | recv], exports its HIP IPC handle (64 bytes; the caller carries the
 *     handles between the processes) and maps the segments of all peers over xGMI;
 *   - advx_comm_allreduce = one reduce launch + the consumer's wait (the exchange number
 *     travels as a kernel argument, so these launches are NOT graph-capturable; every rank must
 *     make the same sequence of calls): the reduce kernel meets its peers on entry ("send
 *     complete"), rank r sums the
 *     r-th slice of all send buffers IN RANK ORDER (every replica gets the same bits), posts
 *     the sum into the recv buffer of every peer and signals "slices posted"; whoever reads
 *     recv waits for that signal first (advx_fused_bwd_dp: inside the update kernel; plain
 *     callers: a one-block launch appended by advx_comm_allreduce);
 *   - a wait that does not complete within timeout_s sets a sticky error word and lets its
 *     kernel go on: a lost peer costs a wrong step, reported by advx_comm_status, never a
 *     hung device.
 * send / recv are device pointers owned by the comm (valid until advx_comm_destroy). */
typedef struct advx_comm advx_comm;
#define ADVX_COMM_HANDLE_BYTES 64
#define ADVX_COMM_MEM_AUTO 0        /* uncached, else fine-grained, else ordinary device memory */
#define ADVX_COMM_MEM_UNCACHED 1
#define ADVX_COMM_MEM_FINEGRAINED 2
#define ADVX_COMM_MEM_DEFAULT 3     /* coherent between processes of ONE device only */
int32_t advx_comm_create(int32_t rank, int32_t world, int64_t floats, int32_t mem_kind, advx_comm** out);
int32_t advx_comm_export(advx_comm* comm, void* handle /* ADVX_COMM_HANDLE_BYTES */);
int32_t advx_comm_connect(advx_comm* comm, const void* handles /* world x 64 bytes, rank order */);
float* advx_comm_send_buffer(advx_comm* comm);
float* advx_comm_recv_buffer(advx_comm* comm);
int32_t advx_comm_mem_kind(const advx_comm* comm);
int32_t advx_comm_allreduce(advx_comm* comm, int64_t floats, double timeout_s, void* stream);
int32_t advx_comm_status(advx_comm* comm, int32_t* timed_out, void* stream); /* synchronises */
int32_t advx_comm_destroy(advx_comm* comm);
/* The data-parallel backward of the fused pair in one call: advx_fused_bwd_io(opt = NULL) into
 * the send buffer, advx_comm_allreduce, advx_fused_update from the recv buffer. */
int32_t advx_fused_bwd_dp(advx_plan* plan, advx_comm* comm, const void* grad_out, int32_t io_dtype,
                          int32_t batch, float* p, const float* x0, float epsilon, float imgfit_scale,
                          const float* mask, float* m, float* v, const advx_opt_scalars* opt,
                          float* s_next, float* v_buf, float* stats, float* scratch, double timeout_s,
                          void* stream);

/* ------------------------------------ suffix-only cross entropy
 * The loss of attack_model.py:324-328 / llavaprocessor.py:73-78 (`logits[:, -suffix:-shift]`
 * against the target tokens, mean reduction) on the logits of the TARGET positions only: the
 * host asks the VLM for its last K = suffix_len + 1 positions (`logits_to_keep`), so the
 * [B, S, V] logits tensor of the reference never exists.  logits: [B, K, V] in the model's
 * dtype (element strides batch_stride / row_stride), the first T positions of every row block
 * supervised by targets[B*T] (int64; values outside [0, V) are ignored).
 *   advx_ce_fwd : row_loss[B*T], row_lse[B*T], mean_and_n = {mean loss, number of valid rows}; `scratch`: at least
 *                 advx_ce_scratch_floats(B*T, V, io_dtype) floats, 8-byte aligned (the row chunks' max and sum of exp:
 *                 a row is read once, by one workgroup per 16 384 halfs / 8 192 floats, two launches in all).
 *   advx_ce_bwd : grad[B, K, V] = (softmax - onehot) * upstream[0] / n_valid on the T supervised
 *                 positions, zeros on the others; same layout as logits, may alias it. */
int32_t advx_ce_fwd(const void* logits, int32_t io_dtype, int64_t batch_stride, int64_t row_stride, int32_t T,
                    const int64_t* targets, int64_t rows, int64_t vocab, float* row_loss, float* row_lse,
                    float* mean_and_n, float* scratch, void* stream);
int64_t advx_ce_scratch_floats(int64_t rows, int64_t vocab, int32_t io_dtype);
int32_t advx_ce_bwd(const void* logits, int32_t io_dtype, int64_t batch_stride, int64_t row_stride, int32_t T,
                    int32_t K, const int64_t* targets, int64_t rows, int64_t vocab, const float* row_lse,
                    const float* mean_and_n, const float* upstream, void* grad, void* stream);

/* ---------------------------------------------------------------- profiling
 * Per-kernel device time of the B*P_out movers (k_fused_fwd, k_fused_bwd, k_fused_step):
 * between begin and end each of their launches carries its own start/stop HIP event pair on
 * the launch stream (hipExtLaunchKernelGGL): every stride-th launch, up to max_launches per
 * kernel (a sparse stride keeps the timed loop itself unperturbed).
 * advx_profile_end synchronises those events and returns, for kinds {0 fwd, 1 bwd, 2 step},
 * the summed milliseconds and the number of timed launches. */
int32_t advx_profile_begin(int32_t max_launches, int32_t stride);
int32_t advx_profile_end(double total_ms[3], int64_t launches[3]);

/* The "re-saved" image of attack_model.py:366-371 (tensor2pil -> tmp.png -> pil_to_tensor; PNG is
 * lossless, so the round trip IS the uint8 quantiser): q = float(uint8(clamp(s,0,1)*255))/255.
 * Feeds the optional loss_resaved forward (:375-379) through advx_emit with no noise. */
int32_t advx_quantise(const float* s, float* q, int64_t n, void* stream);

/* ------------------------------------------------- single ops (unit tests)
 * Same kernels the calls above launch, exposed one by one. */
int32_t advx_tanh_fwd(const float* p, float epsilon, float* x, int64_t n, void* stream);
int32_t advx_tanh_bwd(const float* p, const float* grad_x, float epsilon, float* grad_p, int64_t n, void* stream);
int32_t advx_blur_fwd(const float* x, int32_t H, int32_t W, int32_t k, float sigma, float* y, void* stream);
int32_t advx_blur_bwd(const float* grad_y, int32_t H, int32_t W, int32_t k, float sigma, float* grad_x,
                      float* scratch /* float[3*(H+2r)*(W+2r)] */, void* stream);
int32_t advx_crop_resize_fwd(const float* src, int32_t H, int32_t W, const int32_t* crop_ijhw, float* dst,
                             float* scratch, void* stream);
int32_t advx_crop_resize_bwd(const float* grad_dst, int32_t H, int32_t W, const int32_t* crop_ijhw,
                             float* grad_src, float* scratch, void* stream);
int64_t advx_crop_scratch_floats(int32_t H, int32_t W);
int32_t advx_batch_reduce(const float* g, int32_t batch, int64_t n, float* out, void* stream);
int32_t advx_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVX_H */
