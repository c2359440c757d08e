"""Rank body of tests/test_gpu_dp.py (started by `python -m torch.distributed.run --nproc-per-node 2`).

Each rank owns ONE scene, runs PARQ.training_step (ray-PE node -> decoder node -> set loss) on the HIP path with
`set_data_parallel(True)` and keeps the averaged gradients; rank 0 then recomputes both scenes' gradients alone (data
parallelism off) and checks  dp_grad == mean(rank-0 scene grad, rank-1 scene grad)  for EVERY trainable tensor of the
module (train.py:103 DDP semantics), and that both ranks ended with identical gradients.  Backend: RCCL ("nccl") when
the box has two GPUs, else gloo with both ranks on cuda:0 (same code path above the collective)."""
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
    from types import SimpleNamespace as NS
    from parq_amd import PARQ, Camera, Obb3D, Pose, parallel, synth
    rank, local_rank, world = parallel.env_world()
    ngpu = torch.cuda.device_count()
    backend = "nccl" if ngpu >= world else "gloo"
    device = torch.device("cuda", local_rank if backend == "nccl" else 0)
    torch.cuda.set_device(device)
    parallel.init(backend=backend, device=device if backend == "nccl" else None)

    B, V, h, w, Cd, Qn, I = 1, 3, 24, 30, 256, 48, 3              # N = 2160 keys: batched split-precision attention backward
    dcfg = synth.decoder_cfg(dim=Cd, queries=Qn, heads=4, ffn=256, layers=I, dropout=0.0)
    cfg = NS(MODEL=NS(TOKENIZER=NS(OUT_CHANNELS=Cd, RAY_POINTS_SCALE=dcfg.TRANSFORMER.SCALE, NUM_SAMPLES=64, MIN_DEPTH=0.25, MAX_DEPTH=5.25),
                      DECODER=dcfg), OPTIMIZER=NS(LEARNING_RATE=1e-4, AUTOSCALE_LR=False))
    torch.manual_seed(0)                                            # same initial weights on every rank
    model = PARQ(cfg).to(device).train()
    to = lambda a: torch.from_numpy(a).to(device)

    def batch_of(scene):
        cam, T_cp, T_wp, T_wl = synth.make_geometry(500 + scene, B, V, h, w)
        obbs, sym = synth.make_boxes(600 + scene, B, 5, max_box=8)
        return {"all_features": to(synth.normal(700 + scene, "f", (B, V, Cd, h, w), std=0.5)), "camera_feature": Camera(to(cam)),
                "T_camera_pseudoCam": Pose(to(T_cp)), "T_world_pseudoCam": Pose(to(T_wp)), "T_world_local": Pose(to(T_wl)),
                "obbs_padded": Obb3D(to(obbs)), "sym": to(sym)}

    def grads_of(scene):
        model.zero_grad(set_to_none=True)
        np.random.seed(11)                                          # the matcher's proximity cap draws from NumPy's global generator
        loss = model.training_step(batch_of(scene), 0)
        loss.backward()
        return {n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None}, float(loss)

    model.set_data_parallel(True)
    g_dp, loss_dp = grads_of(rank)
    # every rank must hold the same averaged gradients
    flat = torch.cat([g_dp[n].reshape(-1) for n in sorted(g_dp)])
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    same = all(bool(torch.equal(o, flat)) for o in other)
    # validation scalars averaged like sync_dist=True
    synced = parallel.all_reduce_mean_scalars({"0.25_f1": float(rank)})
    res = {"backend": backend, "world": world, "same_on_all_ranks": same, "synced_f1": synced["0.25_f1"], "n_tensors": len(g_dp)}
    if rank == 0:
        model.set_data_parallel(False)
        singles = [grads_of(s)[0] for s in range(world)]
        worst = ("", 0.0)
        for n in g_dp:
            want = sum(s[n] for s in singles) / world
            rel = float((g_dp[n] - want).norm() / want.norm().clamp_min(1e-30))
            worst = max(worst, (n, rel), key=lambda t: t[1])
        res.update(worst_name=worst[0], worst_rel=worst[1],
                   has_ray_pe=any(n.startswith("add_ray_pe.") for n in g_dp), has_decoder=any(n.startswith("box3d_decoder.") for n in g_dp))
        with open(out_path, "w") as f:
            json.dump(res, f)
    parallel.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
