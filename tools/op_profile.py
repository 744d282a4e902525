#!/usr/bin/env python3
"""Single operations in a loop for a profiler (development tool):
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o p -- python3 tools/op_profile.py [H]
Each op runs 200 times back to back on the same 3 x H x H tensors: kernel time with everything cache-resident and
nothing but the op itself in the dependent chain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from adversarialvlm_amd import ops
H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda:0"
p = torch.randn(3, H, H, device=dev) * 0.3
g = torch.randn(3, H, H, device=dev)
win = (40, 30, H - 112, H - 92)
for _ in range(200):
    ops.tanh_fwd(p, 0.5)
for _ in range(200):
    ops.quantise(p)
for _ in range(200):
    ops.tanh_bwd(p, g, 0.5)
for _ in range(200):
    ops.crop_resize_fwd(p, win)
for _ in range(200):
    ops.crop_resize_bwd(g, win)
for _ in range(200):
    ops.blur_fwd(p, 9, 1.5)
for _ in range(200):
    p.add_(g)            # torch's own pointwise kernel, for scale
for _ in range(200):
    torch.mul(p, 0.5, out=g)
torch.cuda.synchronize()
