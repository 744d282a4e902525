"""GPU tier: the binding INTEGRATION.md shows a maintainer of the reference is code that runs.

Section 1 of INTEGRATION.md holds the ctypes stub (`src/processors/_advx.py` in the reference tree) that replaces the body of
`DifferentiableLlavaImageProcessor.process` (llavaprocessor.py:141-149) by `advx_emit` / `advx_collect`.  This test takes that
code block out of the document verbatim, points its `CDLL` at the in-tree library, and checks that the autograd function it
defines gives what this package's own `ops.ProcessFunction` gives - forward and gradient, bit for bit."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_ctypes_stub_of_the_document_runs():
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd import ops
    from adversarialvlm_amd.plan import CLIP_MEAN, CLIP_STD, Plan
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "class _Process" in b and "make_llava_plan" in b)
    assert 'C.CDLL("libadvx_hip.so")' in stub
    stub = stub.replace('C.CDLL("libadvx_hip.so")', f'C.CDLL({L.LIB_PATH!r})')
    L.load()                                     # torch's HIP runtime first, as the document says
    ns = {}
    exec(compile(stub, "INTEGRATION.md:section-1", "exec"), ns)
    dev = torch.device("cuda:0")
    H, W = 96, 72
    handle = ns["make_llava_plan"](H, W, 56, 56, CLIP_MEAN, CLIP_STD)
    mine = Plan.llava(H, W, 56, 56)
    gen = torch.Generator().manual_seed(3)
    img = torch.rand(3, H, W, generator=gen).to(dev)
    up = torch.randn(1, 3, 56, 56, generator=gen).to(dev)
    a = img.clone().requires_grad_(True)
    out_doc = ns["_Process"].apply(a, handle, mine.out_numel, mine.workspace_floats, (1, 3, 56, 56))
    out_doc.backward(up)
    b = img.clone().requires_grad_(True)
    out_pkg = ops.ProcessFunction.apply(b, mine)
    out_pkg.backward(up.view_as(out_pkg))
    assert torch.equal(out_doc.reshape(-1), out_pkg.reshape(-1))
    assert torch.equal(a.grad, b.grad)
