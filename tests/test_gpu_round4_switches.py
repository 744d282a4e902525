"""GPU tier: the round-4 forms of the image-sized launches against the forms they replaced, bit for bit.

Each `ADVX_TUNE_*` switch of round 4 (include/advx.h) selects between two ways of running the SAME arithmetic in the same order:
the transposed resizes as compiled windows / a window row at a time or with one memory round trip per tap (ROW_BATCH), the gathers'
grids dealt to the XCDs in row groups or workgroup by workgroup (IMG_XCD), canvases resized three channels per thread or one
(HEAD3), the readers of B x P_out mapped to the XCDs like the writers or not (BWD_XCD), the merged blur backward with eight or four
waves per tile (BLUR_THREADS), the prepared chain's image kernels with three channels or one element per thread (TAIL3), the transposed resize of a single plan inside the optimiser's launch or before it (COLLECT_UPDATE), one or two prompts summed inside that gather or by a batch reduction first (DIRECT_BATCH).  Whole chains are stepped with every switch
off in turn and with all of them off; every tensor they leave must be IDENTICAL to the default build's.  (That the specialised
kernels equal the general ones is tests/test_gpu_fastpaths.py; parity with the oracle is the other GPU tests'.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROW_BATCH, IMG_XCD, HEAD3, BWD_XCD, BLUR_THREADS, TAIL3, COLLECT_UPDATE, DIRECT_BATCH, RESET_ALL = 8, 9, 10, 7, 12, 13, 14, 15, 0


def _chains():
    from adversarialvlm_amd.plan import Plan
    q = dict(min_pixels=28 * 28, max_pixels=28 * 28 * 1280)
    return {
        # (H, W, plans, batches, engine keywords, crop window)
        "llava512_prepared": (512, 512, lambda: [Plan.llava(512, 512)], [3], dict(fused_mode="prepared"), None),
        "llava512_prepared_one_prompt": (512, 512, lambda: [Plan.llava(512, 512)], [1], dict(fused_mode="prepared"), None),
        "llava500x640_prepared_two_prompts": (500, 640, lambda: [Plan.llava(500, 640)], [2], dict(fused_mode="prepared"), None),
        "llava200_prepared_two_prompts": (200, 260, lambda: [Plan.llava(200, 260, 96, 128)], [2], dict(fused_mode="prepared"), None),
        "llava512_generic": (512, 512, lambda: [Plan.llava(512, 512)], [3], dict(allow_fused=False), None),
        "llava512_blur_crop": (512, 512, lambda: [Plan.llava(512, 512)], [2], dict(allow_fused=False, blur_kernel=9, use_crop=True),
                               (40, 30, 400, 420)),
        "llava_odd_crop": (97, 130, lambda: [Plan.llava(97, 130, 56, 72)], [3], dict(allow_fused=False, use_crop=True), (3, 5, 80, 101)),
        "llava512_crop": (512, 512, lambda: [Plan.llava(512, 512)], [2], dict(allow_fused=False, use_crop=True), (40, 30, 400, 420)),
        "llava512_accum": (512, 512, lambda: [Plan.llava(512, 512)], [2], dict(allow_fused=False, grad_accum_steps=2), None),
        "llava512_crop_one_prompt": (512, 512, lambda: [Plan.llava(512, 512)], [1], dict(allow_fused=False, use_crop=True), (60, 20, 380, 440)),
        "qwen512_crop": (512, 512, lambda: [Plan.qwen2vl(512, 512, **q)], [2], dict(allow_fused=False, use_crop=True), (16, 24, 470, 450)),
        "qwen512_prepared": (512, 512, lambda: [Plan.qwen2vl(512, 512, **q)], [2], dict(fused_mode="prepared"), None),
        "qwen512_generic": (512, 512, lambda: [Plan.qwen2vl(512, 512, **q)], [2], dict(allow_fused=False), None),
        "phi3_512_prepared": (512, 512, lambda: [Plan.phi3(512, 512)], [2], dict(fused_mode="prepared"), None),
        "phi3_512_generic": (512, 512, lambda: [Plan.phi3(512, 512)], [2], dict(allow_fused=False), None),
        "mllama336_prepared": (336, 336, lambda: [Plan.mllama(336, 336)], [2], dict(fused_mode="prepared"), None),
        "mllama_wide_generic": (300, 1000, lambda: [Plan.mllama(300, 1000, tile=64)], [2], dict(allow_fused=False), None),
        "cross_blur": (336, 336, lambda: [Plan.phi3(336, 336), Plan.qwen2vl(336, 336, **q), Plan.mllama(336, 336)], [2, 2, 2],
                       dict(allow_fused=False, blur_kernel=5, cross_mode=True), None),
        "llava336_pair": (336, 336, lambda: [Plan.llava(336, 336)], [20], dict(fused_mode="pair"), None),
    }


def _run(name, tuning):
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd.pgd import PixelPGD
    H, W, plans_fn, batches, kw, crop = _chains()[name]
    lib = L.load()
    L.check(lib.advx_set_tuning(RESET_ALL, 0), "advx_set_tuning")
    for what, value in tuning:
        L.check(lib.advx_set_tuning(what, value), "advx_set_tuning")
    try:
        gen = torch.Generator().manual_seed(11)
        plans = plans_fn()
        x0 = torch.rand(3, H, W, generator=gen).to(DEV)
        mask = (torch.rand(3, H, W, generator=gen) > 0.2).float().to(DEV)
        eng = PixelPGD(x0, plans, lr=1e-2, mask=mask, seed=5, **kw)
        gs = [[(torch.randn(b, pl.out_numel, generator=gen) * 0.05).to(DEV) for pl, b in zip(plans, batches)] for _ in range(3)]
        outs = []
        for t in range(3):
            pvs = eng.forward(batches, blur_sigma=1.3 if "blur_kernel" in kw else None, crop=crop)
            outs += [pv.clone() for pv in pvs]
            eng.backward_update(gs[t])
            outs += [eng.p.clone(), eng.m.clone(), eng.v.clone(), eng.grad.clone(), eng.image().clone()]
        stats = eng.stats_dict()
        torch.cuda.synchronize()
        return outs, stats
    finally:
        L.check(lib.advx_set_tuning(RESET_ALL, 0), "advx_set_tuning")


VARIANTS = {
    "row_batch_off": [(ROW_BATCH, 0)],
    "img_xcd_off": [(IMG_XCD, 0)],
    "img_xcd_3_rows": [(IMG_XCD, 3)],
    "img_xcd_whole_bands": [(IMG_XCD, 4096)],
    "head3_off": [(HEAD3, 0)],
    "head3_everywhere": [(HEAD3, 1)],
    "bwd_xcd_off": [(BWD_XCD, 0)],
    "blur_threads_256": [(BLUR_THREADS, 256)],
    "tail3_off": [(TAIL3, 0)],
    "collect_update_off": [(COLLECT_UPDATE, 0)],
    "direct_batch_off": [(DIRECT_BATCH, 0)],
    "tail3_loops": [(ROW_BATCH, 0)],
    "all_off": [(ROW_BATCH, 0), (IMG_XCD, 0), (HEAD3, 0), (BWD_XCD, 0), (BLUR_THREADS, 256), (TAIL3, 0),
                (COLLECT_UPDATE, 0)],
}


@pytest.mark.parametrize("name", sorted(_chains()))
def test_round4_forms_leave_the_bits_of_the_forms_they_replaced(name):
    ref, st_ref = _run(name, [])
    for variant, tuning in VARIANTS.items():
        got, st = _run(name, tuning)
        assert len(got) == len(ref)
        for k, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), (name, variant, k, float((a.float() - b.float()).abs().max()))
        for key in st_ref:
            if any(w == TAIL3 for w, _ in tuning) and isinstance(st_ref[key], float):
                # the prepared chain's statistics / ||g|| partials are summed over another partition of the image (doubles)
                assert st[key] == pytest.approx(st_ref[key], rel=1e-6, abs=1e-12), (name, variant, key)
            elif key == "grad_norm" and any(w in (BLUR_THREADS, COLLECT_UPDATE) for w, _ in tuning):
                # the ||g|| partial of a tile is summed in another order with another number of waves (doubles; the float that
                # comes out of them has been the same in every run so far, but only the per-pixel values are promised)
                assert st[key] == pytest.approx(st_ref[key], rel=1e-6), (name, variant, key)
            else:
                assert st[key] == st_ref[key], (name, variant, key)


@pytest.mark.parametrize("H,W,maker", [(512, 512, "llava"), (512, 640, "llava"), (512, 512, "qwen2vl"), (336, 336, "llava"),
                                       (6000, 48, "llava-narrow")])      # one partial row per image row: MORE rows than one per 256 elements
def test_prepared_split_tail_and_reprepare_share_the_partition(H, W, maker):
    """The prepared chain's image kernels all run on ONE partition of the image (three channels per thread from 250 k positions,
    one element per thread below): advx_prepared_bwd in one call and its data-parallel split (advx_prepared_bwd_grad, [all-reduce],
    advx_prepared_update) must leave IDENTICAL tensors AND identical device statistics - the partial sums are the same doubles
    added in the same order.  (Resume = re-preparing: tests/test_gpu_e2e.py::test_resume_continues_bit_for_bit.)"""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    mk = {"llava": lambda: Plan.llava(H, W), "qwen2vl": lambda: Plan.qwen2vl(H, W), "llava-narrow": lambda: Plan.llava(H, W, 64, 48)}[maker]
    gen = torch.Generator().manual_seed(3)
    x0 = torch.rand(3, H, W, generator=gen).to(DEV)
    mask = (torch.rand(3, H, W, generator=gen) > 0.2).float().to(DEV)
    B = 2

    def run(split):
        plan = mk()
        eng = PixelPGD(x0, [plan], lr=1e-2, mask=mask, seed=9, fused_mode="prepared")
        g = torch.Generator().manual_seed(17)
        outs = []
        for t in range(3):
            eng.forward(B)
            go = (torch.randn(B, plan.out_numel, generator=g) * 0.05).to(DEV)
            if not split:
                eng.backward_update([go])
            else:
                # the two halves the data-parallel engine runs around its all-reduce, on one rank
                opt = eng._opt_scalars(True)
                nxt = 1 - eng.s_cur
                ops.prepared_bwd_grad(plan, go, B, eng.p, eng.x0, eng.eps, eng.imgfit_scale(), eng.grad, eng.rows_in, eng.par,
                                      eng.stats, eng.prep_scratch, eng.workspaces[0])
                ops.prepared_update(plan, eng.p, eng.m, eng.v, eng.grad, eng.mask, eng.x0, eng.eps, opt, eng.s_bufs[nxt], eng.par,
                                    eng.stats, eng.prep_scratch, eng.workspaces[0])
                eng.s_cur = nxt
                eng.par = 1 - eng.par
                eng.rows_in = eng.rows_bwd
                eng._scheduler_step()
                eng.iteration += 1
                eng._last = None
            outs += [eng.p.clone(), eng.m.clone(), eng.v.clone(), eng.grad.clone(), eng.s_bufs[eng.s_cur].clone()]
        eng.forward(B)                               # reduces the statistics partials of the last image
        return outs, eng.stats_dict()

    a, st_a = run(False)
    b, st_b = run(True)
    for k, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), (k, float((u - v).abs().max()))
    for key in st_a:
        assert st_a[key] == st_b[key], key
