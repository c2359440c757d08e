"""Where the host time of one inference forward goes (round 6): cProfile over forwards of BASELINE cfg 3 under each policy, and the
wall time of the call's pieces.  Usage: python tools/r06_host_path.py"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(dev)
inputs = bench.build_inputs(1, dev, 1000)
h, w = bench.WORKLOAD["feat_hw"]
step = lambda: dec(*inputs, feat_hw=(h, w))
for _ in range(30):
    step()
torch.cuda.synchronize()


def timed(n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(n):
        t1 = time.perf_counter()
        step()
        host += time.perf_counter() - t1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3


for policy in ("lazy", "sync"):
    for graph in (True, False):
        dec.range_check, dec.use_graph = policy, graph
        for _ in range(5):
            step()
        ms, host = timed()
        print("policy %-5s graph %-5s  %.4f ms per forward, %.4f ms of it inside the call" % (policy, graph, ms, host))
dec.range_check, dec.use_graph = "lazy", True
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(300):
    step()
    if i % 8 == 7:
        torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(35)
print(out.getvalue()[:9000])
