"""CPU tier: the N>1 path (scene sharding, barrier, max-over-ranks, ragged all-gather) with two real
processes over gloo on 127.0.0.1.  The oracle stands in for the compute (the HIP path needs a GPU): rank r
runs its shard of scenes, the shards are gathered and must equal the single-process run of the full batch —
i.e. sharding by scene is exact and needs no data-path collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from parq_amd import parallel, synth
from oracle import parq_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case():
    cfg = synth.decoder_cfg(dim=64, queries=16, heads=1, ffn=64, layers=2)
    W = synth.make_decoder_weights(cfg, 3, damped=True)
    sc = synth.make_scene(4, B=3, V=2, h=8, w=10, C=64, smooth=True)      # 3 scenes over 2 ranks: ragged shards
    return cfg, W, sc


def _run(cfg, W, sc, lo, hi):
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES)
    with torch.no_grad():
        outs = od.forward(*(sc[k][lo:hi] for k in ("tokens", "camera", "T_camera_pseudoCam", "T_world_pseudoCam",
                                                    "T_world_local")))
    return outs[-1]["center_unnormalized"], outs[-1]["pred_logits"]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, lr, w = parallel.init(backend="gloo")
    assert (r, w) == (rank, world)
    cfg, W, sc = _case()
    lo, hi = parallel.shard_range(3, r, w)
    ctr, logits = _run(cfg, W, sc, lo, hi)
    parallel.barrier()
    full_ctr = parallel.all_gather_scenes(ctr, 3)
    full_logits = parallel.all_gather_scenes(logits, 3)
    slowest = parallel.max_over_ranks(1.0 + rank)            # rank 1 reports 2.0
    # validation scalars as Lightning's sync_dist=True logs them (model/parq_lightning.py:133-140): mean over ranks,
    # non-scalars untouched
    synced = parallel.all_reduce_mean_scalars({"0.25_f1": 0.2 + 0.4 * rank, "0.5_f1": float(rank), "curve": np.arange(3) + rank})
    assert abs(synced["0.25_f1"] - 0.4) < 1e-12 and abs(synced["0.5_f1"] - 0.5) < 1e-12
    assert np.array_equal(synced["curve"], np.arange(3) + rank)
    # ranks with DIFFERENT key sets (rank 1 saw no valid scene: compute_metrics returned nothing; rank 0 has a key rank 1 lacks):
    # no hang, no mismatched entries — a key is averaged over the ranks that hold it
    ragged = parallel.all_reduce_mean_scalars({"0.25_f1": 0.8, "only_rank0": 3.0, "shared": 1.0} if rank == 0 else {"shared": 2.0})
    assert abs(ragged["0.25_f1"] - 0.8) < 1e-12 and abs(ragged["only_rank0"] - 3.0) < 1e-12 and abs(ragged["shared"] - 1.5) < 1e-12
    assert parallel.all_reduce_mean_scalars({}) == {}                 # every rank empty: nothing to reduce, still no hang
    q.put((rank, (lo, hi), full_ctr.numpy(), full_logits.numpy(), slowest))
    parallel.barrier()
    torch.distributed.destroy_process_group()


def test_shard_range_is_a_balanced_partition():
    for n in (1, 3, 8, 32, 33):
        for world in (1, 2, 4, 8):
            rs = [parallel.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_scene_sharding_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg, W, sc = _case()
    want_ctr, want_logits = _run(cfg, W, sc, 0, 3)
    assert sorted(r[1] for r in res) == [(0, 2), (2, 3)]
    for rank, _, ctr, logits, slowest in res:
        assert ctr.shape == (3, 16, 3)
        assert np.abs(ctr - want_ctr.numpy()).max() < 1e-5, rank         # scenes are independent: exact up to fp32 batching
        assert np.abs(logits - want_logits.numpy()).max() < 1e-5, rank
        assert slowest == 2.0


# ---------------------------------------------------------------- data-parallel gradient step (flat-arena all-reduce)
def _grad_flat(cfg, W, sc, lo, hi):
    """Gradient of the mean-over-scenes surrogate loss on scenes [lo, hi), flattened in sorted-key order (the oracle's
    autograd stands in for parq_backward, which needs a GPU)."""
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    for k in od.W:
        od.W[k].requires_grad_(True)
    od.prepare(*(sc[k][lo:hi] for k in ("tokens", "camera", "T_camera_pseudoCam", "T_world_pseudoCam", "T_world_local")))
    ref = od.initial_ref()
    loss = 0.0
    for k in range(cfg.TRANSFORMER.DEC_LAYERS):
        out, nxt, _ = od.iterate(ref, k)
        loss = loss + (out["center_unnormalized"] ** 2).mean() + (out["pred_logits"] ** 2).mean()
        ref = nxt.detach()
    loss.backward()
    keys = sorted(k for k, v in od.W.items() if v.grad is not None)
    return torch.cat([od.W[k].grad.reshape(-1) for k in keys])


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    parallel.init(backend="gloo")
    cfg, W, sc = _case()
    lo, hi = parallel.shard_range(2, rank, world)                # 2 scenes, one per rank
    flat = _grad_flat(cfg, W, sc, lo, hi)
    # the same gradient averaged in buckets, the way PARQDecoder.backward does it (bucket 0 = the part of the arena that is
    # final after phase 1 of parq_backward, bucket 1 = the rest; include/parq_hip.h parq_grad_bucket): a ragged split with an
    # empty bucket and an untouched tail, the `ready` hook called once per non-empty bucket in order
    bucketed = flat.clone()
    n = bucketed.numel()
    cut, tail = n // 3 + 1, 5
    calls = []
    parallel.all_reduce_mean_buckets_(bucketed, [(cut, n - cut - tail), (0, 0), (0, cut)], ready=lambda i, stream: calls.append((i, stream)))
    assert calls == [(0, None), (1, None)]
    own_tail = flat[n - tail:].clone()
    parallel.all_reduce_mean_(flat)
    assert torch.equal(bucketed[:n - tail], flat[:n - tail])             # bucketed == flat, bit for bit
    assert torch.equal(bucketed[n - tail:], own_tail)                    # outside the buckets: untouched
    q.put((rank, flat.numpy()))
    parallel.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_flat_gradient_all_reduce_equals_single_process_mean():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg, W, sc = _case()
    want = 0.5 * (_grad_flat(cfg, W, sc, 0, 1) + _grad_flat(cfg, W, sc, 1, 2)).numpy()
    assert np.array_equal(res[0], res[1])                          # every rank holds the same averaged gradient
    assert np.abs(res[0] - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


def test_all_reduce_mean_scalars_is_identity_without_a_group():
    m = {"0.25_f1": 0.5, "n": 3, "arr": np.zeros(2)}
    out = parallel.all_reduce_mean_scalars(m)
    assert out["0.25_f1"] == 0.5 and out["n"] == 3 and out["arr"] is m["arr"]


def test_bench_gpus_n_self_launch_refuses_cleanly_without_enough_gpus():
    """`python bench.py --gpus 2` as typed: the parent decides before touching the GPU; here (no GPU) it must exit with a
    clear message instead of asking for torchrun (VERDICT r01 missing #2)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("this host could really launch 2 ranks")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)


# ---------------------------------------------------------------- host placement (NUMA-local cores per rank)
def _fake_sysfs(root, gpus):
    """A KFD topology + PCI tree like a two-socket, four-GPU host: node 0 / 1 are CPUs (simd_count 0), then the GPUs."""
    os.makedirs(os.path.join(root, "class/kfd/kfd/topology/nodes/0"))
    os.makedirs(os.path.join(root, "class/kfd/kfd/topology/nodes/1"))
    for n in (0, 1):
        with open(os.path.join(root, "class/kfd/kfd/topology/nodes/%d/properties" % n), "w") as f:
            f.write("cpu_cores_count 16\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (bus, cpulist) in enumerate(gpus):
        d = os.path.join(root, "class/kfd/kfd/topology/nodes/%d" % (i + 2))
        os.makedirs(d)
        with open(os.path.join(d, "properties"), "w") as f:
            f.write("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        p = os.path.join(root, "bus/pci/devices/0000:%02x:00.0" % bus)
        os.makedirs(p)
        with open(os.path.join(p, "local_cpulist"), "w") as f:
            f.write(cpulist + "\n")


def test_ranks_pin_to_the_cores_next_to_their_gpu(tmp_path):
    root = str(tmp_path)
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip("needs 4 allowed CPUs")
    a, b = allowed[:len(allowed) // 2], allowed[len(allowed) // 2:]
    fmt = lambda cpus: ",".join(str(c) for c in cpus)
    _fake_sysfs(root, [(0x05, fmt(a)), (0x15, fmt(a)), (0x65, fmt(b)), (0x75, fmt(b))])
    assert parallel.gpu_local_cpus(0, root, visible="") == a and parallel.gpu_local_cpus(3, root, visible="") == b
    assert parallel.gpu_local_cpus(0, root, visible="2,3") == b           # HIP_VISIBLE_DEVICES remaps the ordinals
    assert parallel.gpu_local_cpus(7, root, visible="") == []             # no such device: leave the mask alone
    assert parallel._parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    before = os.sched_getaffinity(0)
    try:
        os.environ.pop("HIP_VISIBLE_DEVICES", None); os.environ.pop("ROCR_VISIBLE_DEVICES", None)
        got = [None] * 4
        for r in range(4):
            os.sched_setaffinity(0, before)
            got[r] = parallel.pin_to_local_cores(r, 4, sysfs=root)
            assert set(os.sched_getaffinity(0)) == set(got[r])
        # ranks 0, 1 share socket A and split it; ranks 2, 3 split socket B: disjoint, local, complete
        assert set(got[0]) | set(got[1]) == set(a) and not set(got[0]) & set(got[1])
        assert set(got[2]) | set(got[3]) == set(b) and not set(got[2]) & set(got[3])
        os.sched_setaffinity(0, before)
        assert parallel.pin_to_local_cores(0, 4, sysfs=os.path.join(root, "nothing-here")) == []      # no topology: untouched
        assert os.sched_getaffinity(0) == before
    finally:
        os.sched_setaffinity(0, before)


def test_ranks_that_share_a_device_split_that_gpus_node(tmp_path):
    """bench.py --share-device (the 8-rank launcher smoke on a 1-GPU lease): every rank uses device 0, so the eight of them split
    device 0's NUMA node into pairwise disjoint slices instead of looking up GPUs that do not exist (and leaving their masks alone)."""
    root = str(tmp_path)
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 8:
        pytest.skip("needs 8 allowed CPUs")
    fmt = lambda cpus: ",".join(str(c) for c in cpus)
    _fake_sysfs(root, [(0x05, fmt(allowed))])
    assert parallel.cpulist_string([0, 1, 2, 5, 7, 8]) == "0-2,5,7-8" and parallel.cpulist_string([]) == ""
    assert parallel._parse_cpulist(parallel.cpulist_string(allowed)) == allowed
    before = os.sched_getaffinity(0)
    try:
        os.environ.pop("HIP_VISIBLE_DEVICES", None); os.environ.pop("ROCR_VISIBLE_DEVICES", None)
        got = []
        for r in range(8):
            os.sched_setaffinity(0, before)
            assert parallel.pin_to_local_cores(r, 8, sysfs=root) == ([] if r > 0 else parallel.pin_to_local_cores(0, 8, sysfs=root))
            os.sched_setaffinity(0, before)
            got.append(parallel.pin_to_local_cores(r, 8, sysfs=root, gpu_index_of=lambda _r: 0))
            assert got[r] and set(os.sched_getaffinity(0)) == set(got[r])
        for i in range(8):
            for j in range(i + 1, 8):
                assert not set(got[i]) & set(got[j]), (i, j)
        assert set().union(*got) == set(allowed)
    finally:
        os.sched_setaffinity(0, before)
    assert parallel.gather_objects({"a": 1}) == [{"a": 1}]                 # no process group: identity
