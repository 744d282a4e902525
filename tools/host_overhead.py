#!/usr/bin/env python3
"""Host time to ENQUEUE one step of the pair (development tool): the loop is timed without a final synchronise while
the device is kept busy, so what is measured is Python + ctypes + HIP launch cost per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from adversarialvlm_amd.pgd import PixelPGD
from adversarialvlm_amd.plan import Plan
dev = torch.device("cuda:0")
x0 = torch.rand(3, 336, 336, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1        # tiny batch: the device is never the bottleneck
eng = PixelPGD(x0, [Plan.llava(336, 336)], lr=1e-2)
g = torch.randn(B, 3, 336, 336, device=dev)
for _ in range(50):
    eng.forward(B); eng.backward_update([g])
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    eng.forward(B)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
for _ in range(n):
    eng.forward(B); eng.backward_update([g])
t3 = time.perf_counter()
torch.cuda.synchronize()
print(f"B={B}: forward alone {1e6 * (t1 - t0) / n:.1f} us/call enqueue; forward+backward {1e6 * (t3 - t2) / n:.1f} us/step enqueue")
