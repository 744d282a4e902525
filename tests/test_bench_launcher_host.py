"""CPU tier: host logic of `python bench.py --gpus N` without a launcher (bench.launch_ranks) - how the ranks are started, with
the child processes replaced by recording fakes: torchrun's environment on 127.0.0.1, rank 0 alone on stdout, gloo when the ranks
have to fold onto fewer devices, exit status 0 only when every rank returned 0, peers of a failed rank stopped, a time limit."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class FakeProc:
    def __init__(self, argv, env, stdout, code, finishes_after):
        self.argv, self.env, self.stdout = argv, env, stdout
        self._code, self._after = code, finishes_after
        self.returncode = None
        self.terminated = self.killed = False

    def poll(self):
        if self.returncode is None and FakeClock.now >= self._after:
            self.returncode = self._code
        return self.returncode

    def terminate(self):
        self.terminated = True
        self.returncode = -15

    def kill(self):
        self.killed = True
        self.returncode = -9

    def wait(self, timeout=None):
        return self.returncode


class FakeClock:
    now = 0.0

    @classmethod
    def monotonic(cls):
        return cls.now

    @classmethod
    def sleep(cls, dt):
        cls.now += max(dt, 0.5)          # the poll loop advances the clock


def _launch(monkeypatch, n, n_dev, codes, after=None, limit=100.0):
    made = []
    after = after or [1.0] * n

    def popen(argv, env=None, stdout=None):
        r = len(made)
        made.append(FakeProc(argv, env, stdout, codes[r], after[r]))
        return made[-1]
    FakeClock.now = 0.0
    monkeypatch.setattr(bench.subprocess, "Popen", popen)
    monkeypatch.setattr(bench, "count_devices", lambda: n_dev)      # the real one asks a child process
    monkeypatch.setattr(bench.time, "monotonic", FakeClock.monotonic)
    monkeypatch.setattr(bench.time, "sleep", FakeClock.sleep)
    rc = bench.launch_ranks(n, ["--gpus", str(n), "--steps", "20"], "nccl", limit)
    return rc, made


def test_ranks_get_the_launcher_environment(monkeypatch):
    rc, procs = _launch(monkeypatch, 8, 8, [0] * 8)
    assert rc == 0 and len(procs) == 8
    ports = {p.env["MASTER_PORT"] for p in procs}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for r, p in enumerate(procs):
        assert p.argv[0] == sys.executable and p.argv[1].endswith("bench.py") and p.argv[2:] == ["--gpus", "8", "--steps", "20"]
        assert (p.env["RANK"], p.env["LOCAL_RANK"], p.env["WORLD_SIZE"], p.env["MASTER_ADDR"]) == (str(r), str(r), "8", "127.0.0.1")
        assert p.env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                    # dmabuf IPC: RCCL and the peer exchange need it
        assert "REHEARSAL" not in p.env["ADVX_BENCH_LAUNCHER"] and "8 ranks on 8 GPU" in p.env["ADVX_BENCH_LAUNCHER"]
        assert (p.stdout is None) == (r == 0)                                # rank 0 inherits stdout, the others go to stderr


def test_fewer_devices_than_ranks_is_a_gloo_rehearsal(monkeypatch):
    rc, procs = _launch(monkeypatch, 4, 1, [0] * 4)
    assert rc == 0
    for p in procs:
        assert p.argv[-2:] == ["--backend", "gloo"] and "REHEARSAL" in p.env["ADVX_BENCH_LAUNCHER"]


def test_no_device_no_ranks(monkeypatch):
    rc, procs = _launch(monkeypatch, 2, 0, [0, 0])
    assert rc == 2 and procs == []


def test_a_failing_rank_fails_the_call_and_its_peers_are_stopped(monkeypatch):
    # rank 1 dies at once, rank 0 would hang in a collective for ever: stopped 15 s later, exit status 1
    rc, procs = _launch(monkeypatch, 2, 2, [0, 3], after=[1e9, 1.0])
    assert rc == 1 and procs[0].terminated and not procs[1].terminated
    assert 15.0 < FakeClock.now < 40.0


def test_nonzero_exit_without_a_hang_and_the_time_limit(monkeypatch):
    rc, _ = _launch(monkeypatch, 2, 2, [0, 1])
    assert rc == 1
    rc, procs = _launch(monkeypatch, 2, 2, [0, 0], after=[1e9, 1e9], limit=30.0)
    assert rc == 1 and all(p.terminated for p in procs) and FakeClock.now < 60.0


def test_devices_are_counted_by_a_child_and_the_launcher_stays_off_the_gpu(monkeypatch):
    """torch.cuda.device_count() may fall back to hipGetDeviceCount - the count is asked of a child instead, and
    the launcher asserts that it holds no HIP context when it starts the ranks."""
    seen = {}

    def run(argv, **kw):
        seen["argv"] = argv

        class R:
            returncode, stdout = 0, "noise\n3\n"
        return R()
    monkeypatch.setattr(bench.subprocess, "run", run)
    assert bench.count_devices() == 3
    assert seen["argv"][0] == sys.executable and "device_count" in seen["argv"][-1]
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(OSError("no python")))
    assert bench.count_devices() == 0
    monkeypatch.setattr(bench.torch.cuda, "is_initialized", lambda: True)
    with pytest.raises(AssertionError):
        _launch(monkeypatch, 2, 2, [0, 0])


def test_e2e_child_outcomes_never_raise(monkeypatch):
    """Timing (B): whatever the child does - result, failure, time-out, garbage - becomes the `e2e` object; (A) is never lost."""
    import argparse
    import json
    import subprocess
    args = argparse.Namespace(e2e_model="synthetic/tiny-llava", e2e_batch=4, e2e_micro=2, e2e_steps=2, e2e_image=56, e2e_budget=5.0)

    class Child:
        def __init__(self, out, code=0, hang=False):
            self.out, self.returncode, self.hang, self.stopped = out, code, hang, False

        def communicate(self, timeout=None):
            if self.hang and not self.stopped:
                raise subprocess.TimeoutExpired("e2e", timeout)
            return self.out, None

        def terminate(self):
            self.stopped = True

        def kill(self):
            self.stopped = True
    rec = {"s_per_step": 1.5, "e2e_prompt_steps_per_s": 42.0, "first_step_s": 6.0, "model": "m", "batch": 4, "micro_batch": 2}
    good = Child("[e2e] chatter\n" + json.dumps(rec) + "\n")
    monkeypatch.setattr(bench.subprocess, "Popen", lambda *a, **k: good)
    out = bench.run_e2e_child(args)
    assert out["s_per_step"] == 1.5 and out["prompt_steps_per_s"] == 42.0 and out["first_step_s"] == 6.0 and "skipped" not in out
    for child, word in ((Child("", code=1), "exit status 1"), (Child("no json here"), "no JSON"), (Child("", hang=True), "budget")):
        monkeypatch.setattr(bench.subprocess, "Popen", lambda *a, _c=child, **k: _c)
        out = bench.run_e2e_child(args)
        assert word in out["skipped"], out
    assert child.stopped                                     # the hung child was stopped (its own PID only)
    monkeypatch.setattr(bench.subprocess, "Popen", lambda *a, **k: (_ for _ in ()).throw(OSError("x")))
    assert "could not start" in bench.run_e2e_child(args)["skipped"]
