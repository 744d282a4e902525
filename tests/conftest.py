import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def lcg_tensor(shape, salt):
    """Same generator as tests/golden/make_golden.py (exact integer arithmetic)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    v = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503)) % np.uint64(2 ** 32)
    v = (v * np.uint64(1664525) + np.uint64(1013904223)) % np.uint64(2 ** 32)
    return torch.from_numpy((v.astype(np.float64) / 2 ** 32 - 0.5).astype(np.float32).reshape(shape))


def rel_err(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
