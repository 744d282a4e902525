#!/usr/bin/env python3
"""Per-call device timing of the C-ABI entry points (back-to-back launches between two HIP
events, so the figure includes one kernel boundary per launch).  Development tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adversarialvlm_amd import _lib as L, ops  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def timeit(fn, iters=300, warm=30):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    dev = torch.device("cuda:0")
    H = W = 336
    for B in (64, 8, 1):
        plan = Plan.llava(H, W)
        plan.upload()
        p = torch.randn(3, H, W, device=dev) * 0.1
        x0 = torch.rand(3, H, W, device=dev)
        g = torch.randn(B, 3 * H * W, device=dev)
        z = torch.randn(B, 3 * H * W, device=dev)
        out = torch.empty(B, 3 * H * W, device=dev)
        stats = torch.zeros(L.STATS_N, device=dev)
        stats[L.STAT_QERR_STD] = 1e-3
        scr = ops.fused_scratch(plan, dev)
        s = torch.empty_like(p)
        m = torch.zeros_like(p)
        v = torch.zeros_like(p)
        gp = torch.zeros_like(p)
        mask = torch.ones_like(p)
        o = L.OptScalars(kind=0, apply=1, lr=1e-2, decay=1 - 1e-4, w1=0.1, beta2=0.999, w2=0.001, bias2_sqrt=0.0316, eps=1e-8,
                         neg_step_size=-0.1)
        mb = B * 3 * H * W * 4 / 1e6
        vb = torch.empty_like(p)
        ops.fused_fwd(plan, p, x0, 0.5, B, stats, scr, s, vb, False, out=out)
        t0 = timeit(lambda: ops.fused_fwd(plan, p, x0, 0.5, B, stats, scr, s, vb, True, out=out))
        t1 = timeit(lambda: ops.fused_fwd(plan, p, x0, 0.5, B, stats, scr, s, vb, True, unit_noise=z, out=out))
        t2 = timeit(lambda: ops.fused_fwd(plan, p, x0, 0.5, B, stats, scr, s, vb, True, philox=(1, 2), out=out))
        t3 = timeit(lambda: ops.fused_bwd(plan, g, B, p, x0, 0.5, 1.0, gp, stats, scr, mask=mask, m=m, v=v, opt=o, s_next=s, v_buf=vb))
        t4 = timeit(lambda: ops.fused_bwd(plan, g, B, p, x0, 0.5, 1.0, gp, stats, scr))
        rf, rs = ops.fused_step_rows(plan)
        ops.fused_fwd(plan, p, x0, 0.5, B, stats, scr, s, vb, False, out=out, parity=0)
        st = dict(par=0, img=rf, nrm=0)

        def step(noise):
            ops.fused_step(plan, g, B, p, x0, 0.5, 1.0, mask, m, v, gp, o, out, s, vb, st["par"], st["img"], st["nrm"], stats,
                           scr, philox=(1, 2) if noise else None)
            st.update(par=1 - st["par"], img=rs, nrm=rs)
        t8 = timeit(lambda: step(True))
        t9 = timeit(lambda: step(False))
        t5 = timeit(lambda: ops.batch_reduce(g))
        ws = torch.empty(plan.workspace_floats, device=dev)
        t6 = timeit(lambda: ops.emit(plan, x0, B, sigma_dev=stats[:1], philox=(1, 2), workspace=ws, out=out))
        t7 = timeit(lambda: out.copy_(g))
        print(f"B={B:3d} ({mb:6.1f} MB): fused_fwd none {t0:6.1f}us  given-noise {t1:6.1f}us  philox {t2:6.1f}us | "
              f"fused_bwd+adamw {t3:6.1f}us  grad-only {t4:6.1f}us | batch_reduce {t5:6.1f}us | emit(philox) {t6:6.1f}us | "
              f"torch copy {t7:6.1f}us ({2 * mb / t7:.2f} TB/s r+w) | fused_step philox {t8:6.1f}us  no-noise {t9:6.1f}us", flush=True)
        for dt in (torch.float16, torch.bfloat16):
            oh, gh = out.to(dt), g.to(dt)
            h0 = timeit(lambda: ops.fused_fwd(plan, p, x0, 0.5, B, stats, scr, s, vb, True, out=oh))
            h1 = timeit(lambda: ops.fused_fwd(plan, p, x0, 0.5, B, stats, scr, s, vb, True, unit_noise=z, out=oh))
            h2 = timeit(lambda: ops.fused_fwd(plan, p, x0, 0.5, B, stats, scr, s, vb, True, philox=(1, 2), out=oh))
            h3 = timeit(lambda: ops.fused_bwd(plan, gh, B, p, x0, 0.5, 1.0, gp, stats, scr, mask=mask, m=m, v=v, opt=o,
                                              s_next=s, v_buf=vb))
            print(f"      io {str(dt):15s}: fused_fwd none {h0:6.1f}us  given-noise {h1:6.1f}us  philox {h2:6.1f}us | "
                  f"fused_bwd+adamw {h3:6.1f}us", flush=True)


if __name__ == "__main__":
    main()
