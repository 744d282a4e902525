"""GPU tier: size-independent properties at BASELINE.json's full sizes - the checks that do not
need the (slow) CPU oracle.

  * adjointness:  <emit(x), g> == <x, collect(g)>  for every plan (the forward is linear in the
    image up to the constant padding / normalisation offset, so the identity holds for
    differences), and the same for blur and crop-resize;
  * linearity of the batch reduction and of collect in g; zero tiles stay exact zeros;
  * idempotence of the quantiser statistics on lattice images; mask idempotence.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _plans():
    from adversarialvlm_amd.plan import Plan
    return [("llava-336", Plan.llava(336, 336)), ("llava-512", Plan.llava(512, 512)), ("mllama-336", Plan.mllama(336, 336)),
            ("mllama-1352x1988", Plan.mllama(1352, 1988)), ("phi3-512", Plan.phi3(512, 512)),
            ("phi3-tall", Plan.phi3(900, 500)), ("qwen-512", Plan.qwen2vl(512, 512)), ("qwen-336", Plan.qwen2vl(336, 336))]


@pytest.mark.parametrize("name", [n for n, _ in _plans()])
def test_emit_collect_adjoint_full_size(dev, name):
    from adversarialvlm_amd import ops
    plan = dict(_plans())[name]
    gen = torch.Generator(device="cpu").manual_seed(1)
    xa = torch.rand(3, plan.in_h, plan.in_w, generator=gen).to(dev)
    xb = torch.rand(3, plan.in_h, plan.in_w, generator=gen).to(dev)
    B = 3
    g = torch.randn(B, plan.out_numel, generator=gen).to(dev)
    # emit is affine in the image: E(xa) - E(xb) = L(xa - xb), repeated over the batch
    d_out = (ops.emit(plan, xa, B) - ops.emit(plan, xb, B)).double()
    lhs = float((d_out * g.double()).sum())
    gx = ops.collect(plan, g, B).double()
    rhs = float(((xa - xb).double() * gx).sum())
    assert lhs == pytest.approx(rhs, rel=2e-5, abs=1e-3), (lhs, rhs)
    # linearity of collect in g
    g2 = torch.randn(B, plan.out_numel, generator=gen).to(dev)
    lin = ops.collect(plan, g + 2 * g2, B) - (ops.collect(plan, g, B) + 2 * ops.collect(plan, g2, B))
    assert float(lin.abs().max()) <= 2e-5 * float(gx.abs().max() + 1)


def test_blur_and_crop_adjoint_full_size(dev):
    from adversarialvlm_amd import ops
    gen = torch.Generator().manual_seed(2)
    for H, W in [(336, 336), (512, 512), (301, 517)]:
        x = torch.randn(3, H, W, generator=gen).to(dev)
        g = torch.randn(3, H, W, generator=gen).to(dev)
        for k, sig in [(5, 7.0), (9, 10.0), (31, 3.3)]:
            lhs = float((ops.blur_fwd(x, k, sig).double() * g.double()).sum())
            rhs = float((x.double() * ops.blur_bwd(g, k, sig).double()).sum())
            assert lhs == pytest.approx(rhs, rel=1e-5, abs=1e-3)
        for crop in [(10, 20, H - 40, W - 50), (0, 0, H, W), (H // 3, W // 4, H // 2, W // 2)]:
            lhs = float((ops.crop_resize_fwd(x, crop).double() * g.double()).sum())
            rhs = float((x.double() * ops.crop_resize_bwd(g, crop).double()).sum())
            assert lhs == pytest.approx(rhs, rel=1e-5, abs=1e-3)


def test_batch_reduce_linear_and_full_size(dev):
    from adversarialvlm_amd import ops
    n = 3 * 336 * 336
    a = torch.randn(64, n, device=dev)
    b = torch.randn(64, n, device=dev)
    ra, rb, rab = ops.batch_reduce(a), ops.batch_reduce(b), ops.batch_reduce(a + b)
    assert float((rab - (ra + rb)).abs().max()) < 1e-4
    assert torch.allclose(ra, a.double().sum(0).float(), atol=2e-5)


def test_zero_tiles_and_padding_are_constant(dev):
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    plan = Plan.mllama(336, 336)
    a = ops.emit(plan, torch.rand(3, 336, 336, device=dev), 2).view(2, 4, 3, 560, 560)
    assert torch.count_nonzero(a[:, 1:]) == 0
    p3 = Plan.phi3(336, 336)
    b = ops.emit(p3, torch.rand(3, 336, 336, device=dev), 1).view(7, 3, 336, 336)
    assert torch.count_nonzero(b[p3.info.num_tiles:]) == 0 and torch.count_nonzero(b[:p3.info.num_tiles]) > 0


def test_quantiser_statistics_vanish_on_lattice_images(dev):
    """q(s) == s on the 256-level lattice: the error statistics of such an image are exactly 0."""
    from adversarialvlm_amd import _lib as L, ops
    x0 = (torch.randint(0, 256, (3, 336, 336)).float() / 255).to(dev)
    stats = torch.zeros(L.STATS_N, device=dev)
    scratch = ops.image_scratch(336, 336, 0, dev)
    ops.image_fwd(torch.zeros_like(x0), x0, 0.5, stats, scratch)
    st = stats.cpu()
    assert float(st[L.STAT_QERR_STD]) == 0.0 and float(st[L.STAT_QERR_MEAN]) == 0.0 and float(st[L.STAT_QERR_L1]) == 0.0


def test_masked_pixels_never_move(dev):
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    x0 = torch.rand(3, 336, 336, device=dev)
    mask = torch.zeros(3, 336, 336, device=dev)
    mask[:, :100, :100] = 1
    for mode in ("pair", "step"):
        eng = PixelPGD(x0, [Plan.llava(336, 336)], mask=mask, fused_mode=mode)
        g = torch.randn(8, 3, 336, 336, device=dev)
        for _ in range(3):
            eng.forward(8)
            eng.backward_update([g])
        assert torch.count_nonzero(eng.p[:, 100:, :]) == 0 and torch.count_nonzero(eng.p[:, :, 100:]) == 0
        assert torch.count_nonzero(eng.p[:, :100, :100]) > 0


@pytest.mark.slow
@pytest.mark.timeout(900)
def test_tensors_beyond_two_to_the_32_elements(dev):
    """64-bit addressing across the batch: pixel_values and their gradient with more than 2^32 elements (18 GB in float32 - a
    fraction of the 288 GB this part has; BASELINE's 64 prompts are 87 MB).  Llama-3.2-Vision's layout with 1200 prompts through
    emit / collect, and LLaVA's fused pair with 13 000 prompts: every sample is written, the last one like the first; the in-kernel
    noise of the last rows is N(0, sigma^2) and differs from row to row; a gradient that is zero but for the LAST sample gives, bit
    for bit, what that sample alone gives."""
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(7)
    # ---- plan-based emit / collect
    plan = Plan.mllama(336, 336)
    n, B = plan.out_numel, 1200
    assert B * n > 2 ** 32
    img = torch.rand(3, 336, 336, generator=gen).to(dev)
    one = ops.emit(plan, img, 1)
    big = ops.emit(plan, img, B)
    for b in (0, 1, B // 2, B - 2, B - 1):
        assert torch.equal(big[b], one[0]), b
    sigma = torch.tensor([0.25], device=dev)
    noisy = ops.emit(plan, img, B, sigma_dev=sigma, philox=(5, 0), out=big)
    live = slice(0, 3 * 560 * 560)                       # the first tile holds the image
    for b in (B - 1, B - 2):
        z = (noisy[b, live] - one[0, live]) / 0.25
        assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3, b
    assert not torch.equal(noisy[B - 1], noisy[B - 2])
    del noisy
    big.zero_()
    row = (torch.randn(n, generator=gen) * 0.01).to(dev)
    big[B - 1].copy_(row)
    g_big = ops.collect(plan, big, B)
    g_one = ops.collect(plan, row.view(1, n), 1)
    assert torch.equal(g_big, g_one)
    del big
    torch.cuda.empty_cache()
    # ---- the fused pair
    B = 13000
    x0 = torch.rand(3, 336, 336, generator=gen).to(dev)
    engines = [PixelPGD(x0, [Plan.llava(336, 336)], lr=1e-2, fused_mode="pair", seed=3) for _ in range(2)]
    pv = engines[0].forward(B)[0]
    assert pv.numel() > 2 ** 32 and tuple(pv.shape) == (B, 3, 336, 336)
    pv1 = engines[1].forward(1)[0]
    sig = 1e-3                                           # sigma0: the first step's noise level (attack_model.py:261)
    for b in (B - 1, B - 2, B // 2):
        z = (pv[b] - pv[0]) / (sig * 2 ** 0.5)          # difference of two independent draws around the same canvas
        assert abs(float(z.mean())) < 1e-2 and abs(float(z.std()) - 1.0) < 1e-2, b
    assert float((pv[0] - pv1[0]).abs().max()) < 10 * sig
    row = (torch.randn(3, 336, 336, generator=gen) * 0.01).to(dev)
    g = torch.zeros_like(pv)
    del pv
    g[B - 1].copy_(row)
    engines[0].backward_update([g])
    engines[1].backward_update([row[None]])
    assert torch.equal(engines[0].grad, engines[1].grad) and torch.equal(engines[0].p, engines[1].p)


def test_two_engines_on_two_streams_do_not_see_each_other(dev):
    """"No implicit synchronisation, re-entrant across streams, no global mutable state" (SURVEY 8b): two engines - LLaVA's fused
    pair and Llama-3.2-Vision's prepared chain with a crop-less blur-less plan of their own - take their steps interleaved on two
    HIP streams, nothing synchronising between the calls; each ends where it ends when it runs alone."""
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(21)
    xa, xb = torch.rand(3, 336, 336, generator=gen).to(dev), torch.rand(3, 120, 200, generator=gen).to(dev)
    Ba, Bb, steps = 16, 6, 6

    def make():
        return (PixelPGD(xa, [Plan.llava(336, 336)], lr=1e-2, fused_mode="pair", seed=1),
                PixelPGD(xb, [Plan.mllama(120, 200, tile=56)], lr=1e-2, fused_mode="prepared", seed=2))
    na, nb = make()[0].plans[0].out_numel, make()[1].plans[0].out_numel
    ga = [(torch.randn(Ba, na, generator=gen) * 0.01).to(dev) for _ in range(steps)]
    gb = [(torch.randn(Bb, nb, generator=gen) * 0.01).to(dev) for _ in range(steps)]

    def step(eng, B, g):
        pv = eng.forward(B)[0]
        eng.backward_update([g.view_as(pv)])

    alone_a, alone_b = make()
    for t in range(steps):
        step(alone_a, Ba, ga[t])
    for t in range(steps):
        step(alone_b, Bb, gb[t])
    torch.cuda.synchronize()
    ea, eb = make()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for t in range(steps):
        with torch.cuda.stream(sa):
            step(ea, Ba, ga[t])
        with torch.cuda.stream(sb):
            step(eb, Bb, gb[t])
    torch.cuda.synchronize()
    assert torch.equal(ea.p, alone_a.p) and torch.equal(eb.p, alone_b.p)
    assert ea.stats_dict() == alone_a.stats_dict() and eb.stats_dict() == alone_b.stats_dict()


def test_two_host_threads_drive_two_engines(dev):
    """The C ABI is called with the GIL released; its error string is thread-local and plans / scratch belong to their callers.
    Two Python threads, each with an engine and a stream of its own, step at the same time - and a third keeps provoking (and
    reading) argument errors meanwhile; every engine ends bit for bit where it ends alone and no error text crosses threads."""
    import threading

    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    gen = torch.Generator().manual_seed(22)
    xs = [torch.rand(3, 96 + 8 * k, 120, generator=gen).to(dev) for k in range(2)]
    steps, B = 25, 4

    def make(k):
        return PixelPGD(xs[k], [Plan.llava(96 + 8 * k, 120, 56, 56)], lr=1e-2, seed=5 + k, fused_mode="prepared")
    n = make(0).plans[0].out_numel
    gs = [[(torch.randn(B, n, generator=gen) * 0.01).to(dev) for _ in range(steps)] for _ in range(2)]
    alone = [make(k) for k in range(2)]
    for k in range(2):
        for t in range(steps):
            pv = alone[k].forward(B)[0]
            alone[k].backward_update([gs[k][t].view_as(pv)])
    torch.cuda.synchronize()
    engines, errors, stop = [make(k) for k in range(2)], [], threading.Event()

    def work(k):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for t in range(steps):
                    pv = engines[k].forward(B)[0]
                    engines[k].backward_update([gs[k][t].view_as(pv)])
                torch.cuda.current_stream().synchronize()
        except Exception as e:                       # noqa: BLE001 - reported below
            errors.append((k, repr(e)))

    def provoke():
        lib = L.load()
        while not stop.is_set():
            rc = lib.advx_plan_create(None, None)
            msg = lib.advx_last_error()
            if rc == 0 or not msg or b"plan" not in msg:
                errors.append(("provoke", rc, msg))
                return

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)] + [threading.Thread(target=provoke)]
    for th in threads:
        th.start()
    for th in threads[:2]:
        th.join()
    stop.set()
    threads[2].join()
    torch.cuda.synchronize()
    assert not errors, errors
    for k in range(2):
        assert torch.equal(engines[k].p, alone[k].p), k


@pytest.mark.parametrize("composed", [False, True])
def test_table_records_follow_the_scratch_not_the_calling_thread(dev, composed):
    """the record of which crop tables an image scratch holds is keyed by the scratch, not by the host thread.
    ONE scratch: forward with window X on thread A, forward with window Y on thread B, backward of X on thread A - with the
    round-3 thread_local record, thread A trusted its stale "same window" and gathered through Y's tables.  The gradient
    must equal the single-thread run (forward X, backward X) bit for bit; own tables (image_fwd / image_bwd) and composed ones
    (forward_multi / collect_crop)."""
    import threading
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import Plan
    H, W = 96, 80
    gen = torch.Generator().manual_seed(17)
    x0 = torch.rand(3, H, W, generator=gen).to(dev)
    p = (torch.randn(3, H, W, generator=gen) * 0.3).to(dev)
    plan = Plan.llava(H, W, 64, 64)
    X, Y = (6, 4, 70, 60), (20, 10, 60, 64)
    assert ops.crop_composes(plan, H, W, X) and ops.crop_composes(plan, H, W, Y)
    g_img = torch.randn(3, H, W, generator=gen).to(dev)
    g_out = torch.randn(2, plan.out_numel, generator=gen).to(dev)
    ws = torch.empty(plan.workspace_floats, device=dev)

    def run(interleave):
        scratch = ops.image_scratch(H, W, 0, dev)
        stats = torch.zeros(L.STATS_N, device=dev)
        s = torch.empty_like(x0)
        out = {}

        def fwd(win):
            if composed:
                ops.forward_multi(p, x0, 0.5, stats, scratch, [plan], [2], s, crop=win, workspaces=[ws])
            else:
                ops.image_fwd(p, x0, 0.5, stats, scratch, crop=win, s=s)

        def bwd(win):
            if composed:
                out["g"] = ops.collect_crop(plan, g_out, 2, win, scratch, grad_s=torch.empty_like(x0), workspace=ws).clone()
            else:
                out["g"] = ops.image_bwd(p, s, g_img, 0.5, 1.0, torch.empty_like(x0), scratch, crop=win).clone()

        def on_thread(fn, *a):
            err = []

            def body():
                try:
                    fn(*a)
                    torch.cuda.synchronize()
                except Exception as e:      # noqa: BLE001 - re-raised on the main thread
                    err.append(e)
            t = threading.Thread(target=body)
            t.start()
            t.join()
            if err:
                raise err[0]
        fwd(X)                              # thread A = this thread
        if interleave:
            on_thread(fwd, Y)               # thread B overwrites the tables in the same scratch
        bwd(X)                              # thread A again
        torch.cuda.synchronize()
        return out["g"]
    alone, mixed = run(False), run(True)
    assert torch.equal(alone, mixed)
    assert float(alone.abs().max()) > 0


def test_plans_and_engines_give_their_device_memory_back(dev):
    """Plans own device copies of their tap tables, engines own scratch and (data parallel) exchange segments: creating and
    dropping a few hundred of them, of every kind, must not move the device's free memory (hipMemGetInfo) - the library
    allocates nothing that outlives its handles."""
    import gc

    from adversarialvlm_amd import ops
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    img = torch.rand(3, 120, 200, device=dev)

    def churn(rounds):
        for r in range(rounds):
            for mk in (lambda: Plan.llava(120, 200, 56, 56), lambda: Plan.mllama(120, 200, tile=56), lambda: Plan.qwen2vl(120, 200),
                       lambda: Plan.phi3(120, 200, num_crops=4)):
                plan = mk()
                out = ops.emit(plan, img, 2)
                ops.collect(plan, torch.ones_like(out), 2)
                eng = PixelPGD(img, [mk()], blur_kernel=5 if r % 2 else None, use_crop=bool(r % 3 == 0), allow_fused=bool(r % 2))
                pv = eng.forward(2, blur_sigma=1.0 if r % 2 else None)[0]
                eng.backward_update([torch.ones_like(pv) * 0.01])
                del plan, out, eng, pv
            from adversarialvlm_amd import dp
            ex = dp.PeerExchange(3 * 120 * 200, dev)          # one rank: the segment alone (uncached device memory)
            ex.send.fill_(1.0)
            assert float(ex.all_reduce().sum()) == 3 * 120 * 200
            ex.close()
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]
    churn(3)                       # first uses: code objects, allocator pools
    before = churn(5)
    after = churn(60)
    assert before - after < 8 * 1024 * 1024, (before, after)
