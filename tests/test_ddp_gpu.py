"""Two-rank rehearsal of the data-parallel bench flow on ONE GPU (gloo carries the all-reduces
through the host; on a multi-GPU node the same code runs with backend "nccl" = RCCL): phases,
three gradient buckets, per-bucket Adam, replica consistency."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import REPO  # noqa: E402


def test_two_rank_bench_flow_keeps_replicas_identical():
    env = dict(os.environ, RV_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(REPO, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "diverged" not in r.stderr, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8192 and out["scaling"] == "weak"
    assert out["value"] > 0 and 0 < out["final_loss"] < 1
    assert "cpu_baseline" not in out          # reported at N=1 only
