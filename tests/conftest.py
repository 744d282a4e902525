import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the random-init `synthetic/*` model architectures are test support, outside the product's registry: this process registers
# them by importing the package, the trainers the tests run as COMMANDS find them through the environment
os.environ.setdefault("ADVX_PLUGIN_MODULES", "adversarialvlm_amd.testing")
import adversarialvlm_amd.testing  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size GPU cases whose oracle takes tens of seconds of CPU")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """How many exceptions the trajectory comparison made in this run (tests/test_gpu_pgd.py): the deterministic tests
    allow two pixels per trajectory, the random trajectories eight - the count belongs in the log the driver keeps."""
    mod = sys.modules.get("test_gpu_pgd")
    if mod is not None:
        terminalreporter.write_line(f"trajectory comparison: {len(mod.ILL_CONDITIONED)} ill-conditioned pixel-steps of p accepted "
                                    f"(AdamW, every gradient of the pixel <= max(1e3 adam_eps, 1e-3 max|g|); <= 2 per deterministic trajectory - 12 for the 786 432-pixel blur + crop case -, <= 8 per random one), {mod.QUANTISER_FLIPS[0]} quantiser-level flips "
                                    "allowed for, 0 oracle values adopted"
                                    + (f"; apart from these, {len(mod.ILL_LARGE)} on the 8 - 25 M-value images of test_trajectories_on_very_large_images "
                                       "(same rule, budget 1e-4 of the optimised values: at 4K the typical first gradient is 1e-4 and 4.5e-5 of the "
                                       "pixels sit at |g| ~ adam_eps)" if mod.ILL_LARGE else ""))
        # who consumed the budgets: one line per test that vetted anything (test id -> pixel-steps of p accepted, level flips)
        for test_id in sorted(set(mod.ILL_BY_TEST) | set(mod.FLIPS_BY_TEST)):
            terminalreporter.write_line(f"   vetted: {test_id}: {mod.ILL_BY_TEST.get(test_id, 0)} pixel-step(s) of p, "
                                        f"{mod.FLIPS_BY_TEST.get(test_id, 0)} quantiser-level flip(s)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _tuning_switches_back_to_default():
    """The library's development switches (advx_set_tuning) are process-wide: whatever a test did with them - through the
    context managers of ops.py, or directly, or by failing half way - the next test starts from the defaults."""
    yield
    lib_mod = sys.modules.get("adversarialvlm_amd._lib")
    lib = getattr(lib_mod, "_lib", None) if lib_mod is not None else None       # only if a test loaded it
    if lib is not None:
        lib.advx_set_tuning(0, 0)           # ADVX_TUNE_RESET_ALL


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def lcg_tensor(shape, salt):
    """Same generator as tests/golden/make_golden.py (exact integer arithmetic)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    v = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503)) % np.uint64(2 ** 32)
    v = (v * np.uint64(1664525) + np.uint64(1013904223)) % np.uint64(2 ** 32)
    return torch.from_numpy((v.astype(np.float64) / 2 ** 32 - 0.5).astype(np.float32).reshape(shape))


ELEMENTWISE_BAR = 1e-4      # north star: "fp32 loss and pixel grads within 1e-4 relative"


def max_err(a, b):
    """Largest elementwise deviation relative to the largest reference magnitude: max|a - b| / max|b|."""
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_err(a, b, elementwise=ELEMENTWISE_BAR, slack=0.0):
    """L2-norm ratio ||a - b|| / ||b|| (what the callers bound, at 5e-6 ... 1e-4).  A norm ratio can hide a handful
    of entries that are off by far more, so every comparison ALSO has to meet the elementwise bar
    max|a - b| <= elementwise * max|b| (None switches it off where a caller states why).  `slack`: an ABSOLUTE allowance
    added to that bar, for a caller that derives it from counted instances (test_gpu_pgd.py: what the vetted pixels of
    p can move downstream)."""
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    if elementwise is not None and a.numel():
        worst = (a - b).abs()
        k = int(worst.argmax())
        bar = elementwise * float(b.abs().max()) + float(slack)
        assert float(worst[k]) <= bar + 1e-30, (f"elementwise bar: |a-b| = {float(worst[k]):.3e} at flat index {k} "
                                                f"(a = {float(a[k]):.9g}, b = {float(b[k]):.9g}) > {elementwise:g} * max|b| = {bar:.3e}")
    return float((a - b).norm() / (b.norm() + 1e-30))
