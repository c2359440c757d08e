"""GPU tier, N > 1: the HIP backward on two ranks with data-parallel gradient averaging of everything the module trains
(decoder arena + ray-PE encoder), against single-process gradients (VERDICT r01 missing #2/#3; train.py:103-125).
Two GPUs -> RCCL; one GPU -> both ranks share cuda:0 and the collective runs over gloo (RCCL refuses duplicate devices)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_two_rank_hip_backward_gradients_are_the_mean_of_single_process_gradients(tmp_path):
    out = tmp_path / "dp.json"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_gpu_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(out.read_text())
    print("\n2-rank DP step:", res)
    assert res["world"] == 2 and res["same_on_all_ranks"]
    assert res["has_ray_pe"] and res["has_decoder"] and res["n_tensors"] >= 40
    assert res["worst_rel"] < 1e-4, res                       # same kernels, same inputs: only atomics / reduction order differ
    assert abs(res["synced_f1"] - 0.5) < 1e-12
    assert res["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
