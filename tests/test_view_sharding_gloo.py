"""CPU tier: intra-scene view sharding ("split-N") over two real gloo ranks with the oracle as compute.  Each rank
holds half of every scene's views; per iteration the sampled-feature sums / valid counts and the per-shard cross-attention
(output, log-sum-exp) go through the two collectives of parq_amd.parallel.  The merged result must equal the single-process run
over all views (float64: to rounding), free-running over all iterations — i.e. the protocol the HIP path implements
(include/parq_hip.h parq_iterate_sharded) is exact."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from parq_amd import parallel, synth
from oracle import parq_oracle as O

KEYS = ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d", "sem_cls_prob", "coord_pos")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case():
    cfg = synth.decoder_cfg(dim=64, queries=24, heads=2, ffn=64, layers=3)
    W = synth.make_decoder_weights(cfg, 7, damped=True)
    sc = synth.make_scene(8, B=2, V=5, h=7, w=9, C=64, smooth=True)         # 5 views over 2 ranks: 3 + 2 (ragged)
    return cfg, W, sc


def _views(sc, lo, hi, h, w):
    B, N, C = sc["tokens"].shape
    V = sc["camera"].shape[1]
    tok = sc["tokens"].reshape(B, V, h * w, C)[:, lo:hi].reshape(B, (hi - lo) * h * w, C)
    return tok, sc["camera"][:, lo:hi], sc["T_camera_pseudoCam"][:, lo:hi], sc["T_world_pseudoCam"][:, lo:hi], sc["T_world_local"]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    parallel.init(backend="gloo")
    cfg, W, sc = _case()
    lo, hi = parallel.view_shard(5, rank, world)
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    with torch.no_grad():
        od.prepare(*_views(sc, lo, hi, 7, 9))
        ref = od.initial_ref()
        outs = []
        for k in range(cfg.TRANSFORMER.DEC_LAYERS):
            out, ref = od.iterate_sharded(ref, k, parallel.merge_sample_sums, parallel.merge_attention_shards)
            outs.append({key: out[key].numpy() for key in KEYS})
    q.put((rank, (lo, hi), outs))
    parallel.barrier()
    torch.distributed.destroy_process_group()


def test_view_sharded_iterations_equal_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, shard, outs = q.get(timeout=300)
        res[r] = (shard, outs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == (0, 3) and res[1][0] == (3, 5)
    cfg, W, sc = _case()
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    with torch.no_grad():
        want = od.forward(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
    for k, w in enumerate(want):
        for key in KEYS:
            for r in (0, 1):                                               # every rank holds the full result
                err = np.abs(res[r][1][k][key] - w[key].numpy()).max()
                assert err < 1e-10, (k, key, r, err)


def test_merge_helpers_without_a_process_group_are_the_single_process_formulas():
    sums = torch.arange(24, dtype=torch.float64).reshape(1, 4, 6)
    cnt = torch.tensor([[0, 1, 2, 3]])
    got = parallel.merge_sample_sums(sums, cnt)
    assert torch.equal(got[0, 0], sums[0, 0]) and torch.allclose(got[0, 2], sums[0, 2] / 2)
    o = torch.randn(1, 2, 3, 4, dtype=torch.float64)
    assert parallel.merge_attention_shards(o, torch.zeros(1, 2, 3, dtype=torch.float64)) is o
    assert parallel.view_shard(5, 1, 2) == (3, 5)
