"""GPU tier: the N > 1 path of bench.py executed on whatever the box has — the driver's 8-GPU run must not be
the first time the rank path (sharding, barrier, max over ranks on the reduction device, the training all-reduce) executes.
With >= 2 GPUs visible the ranks sit on their own GPUs over RCCL ("nccl"); on a 1-GPU lease `--share-device` puts both ranks on
cuda:0 over gloo.  Fresh child processes only (bench.py's self_launch): the pytest process never re-execs."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _run(extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--no-cpu-baseline", "--no-b32"] + extra,
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_forward_line():
    d = _run(["--steps", "4", "--warmup", "1"])
    want_backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    assert d["n_gpus"] == 2 and d["collective_backend"] == want_backend
    assert d["rccl_ranks"] == (2 if want_backend == "nccl" else 0)            # no false RCCL claim on a shared device
    assert d["scaling"] == "weak" and d["steps"] == 4 and d["parq_env"] == {} and d["dev_lib"] is False
    # weak-scaling arithmetic: both ranks' scenes x 8 iterations x steps over the slowest rank's time
    iters = 2 * d["config"]["scenes_per_gpu"] * 8 * d["steps"]
    assert abs(d["value"] - iters / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert "dp2" in d["config"]["parallelism"]
    assert d["value"] > 100                                                     # two forwards really ran on the GPU
    assert d["roofline"]["achieved"] and d["roofline"]["frac"] > 0.05


def test_bench_two_ranks_training_line():
    d = _run(["--train", "--steps", "2", "--warmup", "1", "--scenes-per-gpu", "2"])
    want_backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    assert d["n_gpus"] == 2 and d["collective_backend"] == want_backend
    assert d["unit"] == "steps/sec" and d["value"] > 0
    assert abs(d["scenes_per_sec"] - d["value"] * 2 * 2) < 1e-6 * d["scenes_per_sec"]
    assert d["final_loss"] == d["final_loss"]                                   # finite (not NaN) after the all-reduced steps
