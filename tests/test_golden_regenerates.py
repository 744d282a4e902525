"""CPU tier, build container only: the committed fixtures ARE what the reference produces.

`tests/golden/make_golden.py` imports the reference from /root/reference and writes every fixture; with GOLDEN_OUT it writes
them somewhere else.  Where the reference tree exists this test regenerates all of them into a scratch directory and compares
array by array (and the JSON value by value) with the committed files: integers, indices and strings exactly, floats to 1e-6
(they come out bit for bit in this container's default configuration - parallel reductions may group their sums differently under
another thread count); the two files that hold what the reference's
`train()` loops logged run whole models, whose GEMMs may group their sums differently under another thread count, so their float
arrays are held to 2e-5 instead (the final images to 1e-4 in the L2 norm) (in the build container they come out bit for bit as well).  On the GPU box there is no reference
tree: skipped (nothing at test time reads /root/reference there)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree exists in the build container only")
@pytest.mark.timeout(900)
def test_every_fixture_regenerates_from_the_reference(tmp_path):
    out = str(tmp_path)
    res = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py")], env=dict(os.environ, GOLDEN_OUT=out),
                         cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, res.stderr[-3000:]
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert committed and sorted(f for f in os.listdir(out) if f.endswith(".npz")) == committed
    for name in committed:
        a, b = np.load(os.path.join(GOLDEN, name)), np.load(os.path.join(out, name))
        assert sorted(a.files) == sorted(b.files), name
        loops = name in ("trainer_run_reference.npz", "cross_trainer_run_reference.npz")
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            if loops and a[k].dtype.kind == "f" and k.endswith("_final"):
                # the image after a few AdamW steps: sign-like first steps amplify a last-bit difference at pixels whose gradient is ~ 0
                d = a[k].astype(np.float64) - b[k].astype(np.float64)
                assert np.linalg.norm(d) <= 1e-4 * np.linalg.norm(a[k]) and np.abs(d).max() <= 2e-2, (name, k)
            elif loops and a[k].dtype.kind == "f":
                assert np.allclose(a[k], b[k], rtol=2e-5, atol=1e-8), (name, k)
            elif loops and a[k].dtype.kind == "u" and k.endswith("_final_png"):
                assert int(np.abs(a[k].astype(np.int32) - b[k].astype(np.int32)).max()) <= 1, (name, k)
            elif loops and k.endswith("_probe0"):
                assert a[k].shape == b[k].shape and np.array_equal(a[k][:, 0], b[k][:, 0]), (name, k)     # questions; the text is greedy decoding
            elif a[k].dtype.kind == "f":
                # bit for bit in the container's default configuration; a parallel reduction (a checksum, an interpolation's
                # accumulation) may group its sums differently under another thread count
                assert np.allclose(a[k], b[k], rtol=1e-6, atol=1e-6 * max(1.0, float(np.abs(a[k]).max()) if a[k].size else 1.0)), (name, k)
            else:
                assert np.array_equal(a[k], b[k]), (name, k)
    assert json.load(open(os.path.join(GOLDEN, "cli_flags_reference.json"))) == json.load(open(os.path.join(out, "cli_flags_reference.json")))
