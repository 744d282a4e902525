"""GPU tier: the drop-in boundary is a C ABI, so a host that is neither Python nor torch must be able to drive the hot path
with `include/advx.h` alone.  tests/cabi/pair_steps.c (plain C, gcc, hipMalloc'ed buffers, the NULL stream) runs three steps of
the headline pair; the same steps driven from Python through ctypes must leave the same BYTES - optimised tensor, last
pixel_values (in-kernel Philox noise included) and the statistics."""
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import lcg_tensor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_plain_c_host_and_python_host_leave_the_same_bytes(tmp_path):
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd.pgd import PixelPGD
    from adversarialvlm_amd.plan import Plan
    exe, out = str(tmp_path / "pair_steps"), str(tmp_path / "c_host.bin")
    lib_dir = os.path.join(ROOT, "adversarialvlm_amd")
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tests", "cabi", "pair_steps.c"), "-I" + os.path.join(ROOT, "include"),
                    "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + lib_dir, "-ladvx_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, text=True)
    res = subprocess.run([exe, out], capture_output=True, text=True, timeout=200)
    assert res.returncode == 0, res.stderr
    H = W = 64
    B, n = 8, 3 * 64 * 64
    raw = np.fromfile(out, dtype=np.float32)
    assert raw.size == n + n * B + L.STATS_N
    c_p, c_out, c_stats = raw[:n], raw[n:n + n * B], raw[n + n * B:]
    dev = torch.device("cuda:0")
    x0 = (lcg_tensor((n,), 7) + 0.5).reshape(3, H, W)
    eng = PixelPGD(x0.to(dev), [Plan.llava(H, W, H, W)], lr=1e-2, seed=1234, fused_mode="pair")
    for t in range(3):
        pv = eng.forward(B)[0]
        g = (lcg_tensor((n * B,), 100 + t) * 0.02).reshape(B, 3, H, W).to(dev)
        eng.backward_update([g])
    st = eng.stats_dict()
    assert np.array_equal(eng.p.cpu().numpy().ravel(), c_p)
    assert np.array_equal(pv.cpu().numpy().ravel(), c_out)
    assert np.array_equal(eng.stats.cpu().numpy()[:8], c_stats[:8]) and st["grad_norm"] > 0
