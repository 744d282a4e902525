"""CPU tier: the host-side helpers of bench.py (no GPU, nothing launched)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_profile_stride_times_at_least_ten_launches_per_kernel():
    b = _bench()
    for steps in list(range(1, 64)) + [100, 200, 639, 640, 1000, 5000, 100000]:
        stride = b.profile_stride(steps)
        timed = (steps + stride - 1) // stride          # launches 0, stride, 2*stride, ... of `steps`
        assert stride >= 1
        if steps >= 10:
            assert timed >= 10, (steps, stride, timed)
        if steps >= 640:
            assert timed >= 64, (steps, stride, timed)
    assert b.profile_stride(20) == 2 and b.profile_stride(1000) == 15      # the driver's run and the default


def test_host_cores_reports_usable_and_machine_counts():
    b = _bench()
    usable, total = b.host_cores()
    assert 1 <= usable <= total == (os.cpu_count() or 1)


def test_workload_constants_are_baseline_configs_1():
    b = _bench()
    assert (b.H, b.W, b.BATCH) == (336, 336, 64) and b.HBM_PEAK_GBS == 8000.0
