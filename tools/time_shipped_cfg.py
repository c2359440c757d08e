"""Forward time of the reference's SHIPPED decoder size (config/train.yaml: DEC_DIM 1024, 4 heads -> head dim 256, 256 queries,
8 iterations) on the BASELINE cfg-3 geometry (10 views, 120x160 features).  Not the headline metric (BASELINE.json quotes d=256)."""
import os, sys, time
import torch
torch.set_grad_enabled(False)      # inference tool: the reference's drivers run these calls under no_grad (eval.py:46)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.WORKLOAD["dim"] = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
MODE = sys.argv[2] if len(sys.argv) > 2 else None
device = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(device)
inputs = bench.build_inputs(1, device, seed=1000)
if MODE:
    dec.attention_mode = MODE
h, w = bench.WORKLOAD["feat_hw"]
with torch.no_grad():
    for _ in range(2):
        dec(*inputs, feat_hw=(h, w))
    torch.cuda.synchronize()
    dec.profile_enable(True)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        dec(*inputs, feat_hw=(h, w))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    prof = dec.profile_read()
    dec.profile_enable(False)
import json
print(json.dumps({"metric": "decoder-iterations/sec (10 views, 256 queries, d=%d: the reference's shipped decoder size, NOT the BASELINE metric)" % bench.WORKLOAD["dim"],
                  "value": bench.WORKLOAD["iters"] / dt, "unit": "decoder-iterations/sec", "n_gpus": 1, "steps": n, "ms_per_step": dt * 1e3,
                  "attention_mode": dec.attention_mode, "dtype": "f32 (split fp16 hi/lo products in the cross-attention and K/V projection)" if dec.attention_mode == "split" else "f32",
                  "data": "synthetic", "config": {"workload": "cfg-3 geometry (10 views, 120x160 features, N=192000), 256 queries, 8 iterations, d=%d, 4 heads" % bench.WORKLOAD["dim"]},
                  "kernel_groups_ms_per_step": {k: v[0] / n for k, v in prof.items()}}))
