"""GPU tier, N > 1: intra-scene view sharding ("split-N"; include/parq_hip.h parq_iterate_sharded) on the HIP path.
Two ranks each hold half of every scene's views; PARQDecoder.forward_view_sharded must reproduce the single-process forward
over all views: <= 5e-5 teacher-forced and <= 1e-4 free-running on damped weights (only the fp32 summation order over views /
key shards differs; measured 5e-6 .. 2e-5 forced, 2e-5 .. 5e-5 free over 4 iterations), with identical results on every rank.  Two GPUs -> RCCL; one GPU -> both ranks share cuda:0 over gloo."""
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
def test_two_rank_view_sharded_forward_equals_single_process(tmp_path):
    out = tmp_path / "vs.json"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "view_shard_gpu_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(out.read_text())["ranks"]
    print("\n2-rank view-sharded forward:", res)
    assert len(res) == 2
    for rk in res:
        assert rk["world"] == 2 and rk["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
        for tag in ("b2", "b1"):
            assert rk[tag]["ranks_agree"]
            assert rk[tag]["worst_forced"] < 5e-5, rk
            assert rk[tag]["worst_free"] < 1e-4, rk
    assert res[0]["b2"]["views"] == [0, 3] and res[1]["b2"]["views"] == [3, 6]
    # only rank 1's shard left the fp16 range: BOTH ranks flag it and decide alike (the flag is summed in the first exchange)
    for rk in res:
        lazy, sync = rk["range"]["lazy"], rk["range"]["sync"]
        assert lazy["warned"] and lazy["mode_after"] == "fp32" and lazy["all_nan"], rk["range"]
        assert sync["warned"] and sync["mode_after"] == "fp32" and sync["all_finite"] and sync["ranks_agree"], rk["range"]


def test_single_process_view_sharded_call_equals_forward():
    """world = 1 (no process group): the three-phase path with trivial exchanges must equal the ordinary forward."""
    from parq_amd import synth
    from gpu_util import make_decoder, scene_args
    cfg = synth.decoder_cfg(dim=256, queries=64, heads=4, ffn=768, layers=3)
    W = synth.make_decoder_weights(cfg, 95, damped=True)
    sc = synth.make_scene(96, 2, 3, 20, 24, 256, smooth=True)
    dec = make_decoder(cfg, W)
    a = scene_args(sc)
    with torch.no_grad():
        want = dec(*a, feat_hw=(20, 24))
    got = dec.forward_view_sharded(*a, feat_hw=(20, 24))
    for k in range(3):
        for key in want[k]:
            err = float(((want[k][key].double() - got[k][key].double()).abs() / want[k][key].double().abs().clamp(min=1.0)).max())
            # (the sharded call runs the self out-projection and the query projection as two launches, the plain forward as one launch
            # with norm1 pushed through the projection — chain.hip seam_tile: another rounding of the same fp32 arithmetic, 2.4e-6 here;
            # the call also asserts the ADVICE r04 fix: mode "split8" runs as "split" on this path)
            assert err < (5e-6 if k == 0 else 5e-5), (k, key, err)     # free-running (damped weights): 1.0e-5 at iteration 1
    assert dec._mode_set == "split" and dec.attention_mode == "split8"
