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


def _run(extra, gpus=2):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--share-device", "--no-cpu-baseline", "--no-b32"] + extra,
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


def test_bench_eight_ranks_launcher_smoke():
    """the self-launcher, the rendezvous port, the per-rank host-core slices and the shutdown at EIGHT local ranks,
    before the driver's first 8-GPU run (train.py:103-125 is the reference's launcher).  On a 1-GPU lease all eight ranks share cuda:0
    over gloo — the numbers mean nothing, the rank path is the subject: eight per-rank step times, host-core slices that are
    pairwise disjoint (where the topology is readable), one JSON line, exit code 0 (clean destroy_process_group)."""
    d = _run(["--steps", "2", "--warmup", "1"], gpus=8)
    want_backend = "nccl" if torch.cuda.device_count() >= 8 else "gloo"
    assert d["n_gpus"] == 8 and d["collective_backend"] == want_backend
    assert len(d["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in d["per_rank_ms_per_step"])
    assert len(d["per_rank_pinned_cpulist"]) == 8
    assert d["rccl_version"]                                        # the RCCL the build links, whatever backend this box could use
    from parq_amd.parallel import _parse_cpulist
    sets = [set(_parse_cpulist(c)) for c in d["per_rank_pinned_cpulist"] if c]
    assert len(sets) in (0, 8), d["per_rank_pinned_cpulist"]       # all ranks pinned, or none (no topology information)
    for i in range(len(sets)):
        for j in range(i + 1, len(sets)):
            assert not (sets[i] & sets[j]), (i, j, d["per_rank_pinned_cpulist"])
    iters = 8 * d["config"]["scenes_per_gpu"] * 8 * d["steps"]
    assert abs(d["value"] - iters / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert "dp8" in d["config"]["parallelism"]


def test_bench_eight_ranks_four_scenes_per_gpu_variant():
    """SURVEY.md section 8(d) names two lines for the scaling runs: B = #GPUs and a B = 4 per GPU variant (BASELINE cfg 4's shard size).
    The same launcher at eight ranks with --scenes-per-gpu 4, so that the driver's first 8-GPU run can take both lines."""
    d = _run(["--steps", "1", "--warmup", "1", "--scenes-per-gpu", "4"], gpus=8)
    assert d["n_gpus"] == 8 and d["config"]["scenes_per_gpu"] == 4 and "dp8" in d["config"]["parallelism"]
    assert len(d["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in d["per_rank_ms_per_step"])
    iters = 8 * 4 * 8 * d["steps"]
    assert abs(d["value"] - iters / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert d["guard_policy_cost"] is None if "guard_policy_cost" in d else True     # (single-rank extras stay out of multi-rank lines)


def test_bench_single_rank_default_line_contract():
    """The driver's N = 1 command (fewer steps, without the CPU baseline and the optional records): ONE JSON line with the contract's
    fields, the roofline object of the dominant kernel, and the sub-records measured beside `value` — the strict fp16 x 3 run and two
    scenes in flight on two HIP streams, whose outputs must be bit for bit those of one-at-a-time forwards."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-b32",
                        "--no-peaked", "--no-pmc"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["unit"] == "decoder-iterations/sec" and d["higher_is_better"] is True
    assert abs(d["value"] - 8 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and 0.2 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    two = d["two_scenes_in_flight"]
    assert two["streams"] == 2 and two["outputs_bit_identical_to_one_at_a_time"] is True
    assert two["value"] > 0.95 * d["value"]                      # never slower than one at a time (measured + 15-18 %)
    three = d["three_scenes_in_flight"]
    assert three["streams"] == 3 and three["value"] > 0.95 * d["value"]
    assert d["strict_fp16x3"]["attention_mode"] == "split" and d["strict_fp16x3"]["value"] < d["value"]
    # the never-NaN default and what it costs, the captured forward's host time (item 4)
    g = d["guard_policy_cost"]
    assert g["default_policy"] == "sync" and d["attention_guard"]["policy"] == "sync"
    assert g["lazy"]["value"] >= 0.97 * g["sync"]["value"] and 0.0 <= g["cost_of_the_default"] < 0.15
    assert d["host_enqueue_ms"] == g["host_enqueue_ms"] and d["host_enqueue_ms"] < g["host_enqueue_ms_without_captured_forward"]
