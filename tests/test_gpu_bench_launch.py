"""GPU tier: `python3 bench.py --gpus N` exactly as the driver's SCALE stage calls it - no launcher around it.
The parent starts the N ranks itself before it touches the GPU (bench.py: launch_ranks), on a one-GPU box the
ranks fold onto device 0 and the host collectives travel over gloo (said so in `config.launcher`).  One shell-out:
ONE JSON line on stdout, rc 0, the replicas of p bit-identical, weak and strong figures present."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=540)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5
    assert line["scaling"] == "weak" and line["config"]["prompts_per_gpu"] == 64 and line["config"]["global_prompts"] == 128
    assert line["config"]["replicas_identical"] is True
    assert line["config"]["exchange_timed_out"] in (None, False)
    assert "self-launched: 2 ranks" in line["config"]["launcher"]
    assert line["config"]["exchange_report"]["chosen"] in ("peer", "host")
    assert line["strong"]["scaling"] == "strong" and line["strong"]["global_prompts"] == 64 and line["strong"]["value"] > 0
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    print("bench --gpus 2 (self-launched):", line["value"], line["unit"], "| strong", line["strong"]["value"],
          "|", line["config"]["exchange"])


def test_bench_launcher_refuses_without_a_gpu():
    """CPU tier: the parent of a --gpus N run never hangs or falls back - no device, exit status 2 (runs only where
    there is no GPU; on the GPU box the test above covers the entry)."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    assert res.returncode == 2 and "needs a GPU" in res.stderr and res.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_launcher_reports_a_failing_rank():
    """A rank that fails must fail the whole call: no JSON line on stdout, exit status 1, the ranks' exit codes on stderr - never a
    hang and never a line from a partial run.  (The one-launch chain cannot host the gradient exchange: every rank raises.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--chain", "step"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=540)
    assert res.returncode == 1, (res.returncode, res.stderr[-1500:])
    assert res.stdout.strip() == ""
    assert "rank exit codes" in res.stderr or "ranks stopped" in res.stderr
    assert "cannot host the gradient all-reduce" in res.stderr
