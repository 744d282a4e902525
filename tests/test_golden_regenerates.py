"""CPU tier, build container only: the committed fixtures ARE what the reference produces.

`tests/golden/make_golden.py` imports the reference from /root/reference and writes every fixture; with GOLDEN_OUT it writes
them somewhere else.  Where the reference tree exists this test regenerates all of them into a scratch directory and compares
array by array (and the JSON value by value) with the committed files - bit for bit.  On the GPU box there is no reference
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
def test_every_fixture_regenerates_bit_for_bit(tmp_path):
    out = str(tmp_path)
    res = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py")], env=dict(os.environ, GOLDEN_OUT=out),
                         cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, res.stderr[-3000:]
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert committed and sorted(f for f in os.listdir(out) if f.endswith(".npz")) == committed
    for name in committed:
        a, b = np.load(os.path.join(GOLDEN, name)), np.load(os.path.join(out, name))
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (name, k)
    assert json.load(open(os.path.join(GOLDEN, "cli_flags_reference.json"))) == json.load(open(os.path.join(out, "cli_flags_reference.json")))
