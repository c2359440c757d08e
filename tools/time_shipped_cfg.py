"""Forward time of the reference's SHIPPED decoder size (config/train.yaml: DEC_DIM 1024, 4 heads -> head dim 256, 256 queries,
8 iterations) on the BASELINE cfg-3 geometry (10 views, 120x160 features).  Not the headline metric (BASELINE.json quotes d=256)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.WORKLOAD["dim"] = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
device = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(device)
inputs = bench.build_inputs(1, device, seed=1000)
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
print("d=%d: forward %.2f ms -> %.0f decoder-iterations/s; groups (ms per forward): %s" % (
    bench.WORKLOAD["dim"], dt * 1e3, bench.WORKLOAD["iters"] / dt, {k: round(v[0] / n, 3) for k, v in prof.items()}))
