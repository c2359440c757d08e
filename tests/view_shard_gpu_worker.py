"""Rank body of tests/test_gpu_view_sharding.py (started by `python -m torch.distributed.run --nproc-per-node 2`).

Every rank holds HALF of the scene's views (tokens, cameras, poses of those views only) and the ranks run
PARQDecoder.forward_view_sharded collectively on the HIP path; rank 0 also runs the ordinary single-process forward over all
views and compares (teacher-forced with the single-process reference points, and free-running on damped weights).  Backend:
RCCL ("nccl") when the box has two GPUs, else gloo with both ranks on cuda:0 (same code path above the collectives)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(out_path):
    from parq_amd import parallel, synth
    from gpu_util import make_decoder
    rank, local_rank, world = parallel.env_world()
    ngpu = torch.cuda.device_count()
    backend = "nccl" if ngpu >= world else "gloo"
    device = torch.device("cuda", local_rank if backend == "nccl" else 0)
    torch.cuda.set_device(device)
    parallel.init(backend=backend, device=device if backend == "nccl" else None)

    B, V, h, w, Cd, Qn, I = 2, 6, 40, 52, 256, 256, 4                 # N = 12 480 keys per scene; the fixed 64-split merge at B = 1 is
    cfg = synth.decoder_cfg(dim=Cd, queries=Qn, heads=4, ffn=768, layers=I)      # covered by the B = 1 pass below
    W = synth.make_decoder_weights(cfg, 91, damped=True)
    res = {"world": world, "backend": backend}
    for tag, Bn in (("b2", B), ("b1", 1)):
        sc = synth.make_scene(92, Bn, V, h, w, Cd, smooth=True)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        lo, hi = parallel.view_shard(V, rank, world)
        tok = sc["tokens"].reshape(Bn, V, h * w, Cd)[:, lo:hi].reshape(Bn, (hi - lo) * h * w, Cd)
        local = (to(tok), to(sc["camera"][:, lo:hi]), to(sc["T_camera_pseudoCam"][:, lo:hi]), to(sc["T_world_pseudoCam"][:, lo:hi]),
                 to(sc["T_world_local"]))
        full = (to(sc["tokens"]), to(sc["camera"]), to(sc["T_camera_pseudoCam"]), to(sc["T_world_pseudoCam"]), to(sc["T_world_local"]))
        dec = make_decoder(cfg, W)
        with torch.no_grad():
            want = dec(*full, feat_hw=(h, w))                                       # single process, all views (every rank computes it)
        got = dec.forward_view_sharded(*local, feat_hw=(h, w))                 # free-running, sharded
        lo_s, hi_s = np.asarray(cfg.TRANSFORMER.SCALE[0::2], np.float32), np.asarray(cfg.TRANSFORMER.SCALE[1::2], np.float32)
        forced = [torch.from_numpy((o["coord_pos"].cpu().numpy() - lo_s) / (hi_s - lo_s)) for o in want]
        got_f = dec.forward_view_sharded(*local, feat_hw=(h, w), forced_refs=forced)
        worst_free = worst_forced = 0.0
        for k in range(I):
            for key in want[k]:
                a, b, c = want[k][key].double(), got[k][key].double(), got_f[k][key].double()
                worst_free = max(worst_free, float(((a - b).abs() / a.abs().clamp(min=1.0)).max()))
                worst_forced = max(worst_forced, float(((a - c).abs() / a.abs().clamp(min=1.0)).max()))
        # all ranks must hold the same result
        t = got[-1]["center_unnormalized"].clone()
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        res[tag] = {"worst_free": worst_free, "worst_forced": worst_forced,
                    "ranks_agree": bool(all(torch.equal(g, gathered[0]) for g in gathered)), "views": [lo, hi]}
    # ---- fp16 operand range: ONE rank's shard is out of range.  The flag travels in the first exchange, so EVERY
    # rank poisons its outputs (NaN, never unflagged numbers merged from the other rank's inf record) and every rank takes the
    # same decision: "lazy" -> all warn and switch to the fp32 kernels; "sync" -> all re-run the forward with them.
    import warnings
    sc = synth.make_scene(93, 1, V, h, w, Cd, smooth=True)
    lo, hi = parallel.view_shard(V, rank, world)
    tok = sc["tokens"].reshape(1, V, h * w, Cd)[:, lo:hi].reshape(1, (hi - lo) * h * w, Cd).copy()
    if rank == world - 1:
        tok[0, 7, 3] = 9.0e4                                        # only the LAST rank holds an out-of-range token
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    local = (to(tok), to(sc["camera"][:, lo:hi]), to(sc["T_camera_pseudoCam"][:, lo:hi]), to(sc["T_world_pseudoCam"][:, lo:hi]),
             to(sc["T_world_local"]))
    rng = {}
    for policy in ("lazy", "sync"):
        dec = make_decoder(cfg, W)
        dec.range_check = policy
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            outs = dec.forward_view_sharded(*local, feat_hw=(h, w))
        torch.cuda.synchronize()
        t = outs[-1]["center_unnormalized"]
        rng[policy] = {"mode_after": dec.attention_mode, "warned": any("fp16 range" in str(r.message) for r in rec),
                       "all_nan": bool(torch.isnan(t).all()), "all_finite": bool(torch.isfinite(t).all())}
        if policy == "sync":                                        # the re-run must agree across the ranks
            gathered = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(gathered, t.contiguous())
            rng[policy]["ranks_agree"] = bool(all(torch.equal(g, gathered[0]) for g in gathered))
    res["range"] = rng
    all_res = [None] * world
    dist.all_gather_object(all_res, res)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"ranks": all_res}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
