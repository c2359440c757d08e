"""Experiment: BASELINE cfg 3 forwards of independent scenes on ONE stream against TWO (or more) streams with one forward in flight
on each (one module instance per stream: own native handle, own workspace).  The small-op chain of one forward leaves most of the
chip idle; the question is how much of it another forward's K/V projection / cross-attention fills."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
NS = int(os.environ.get("NSTREAMS", "2"))
steps = int(os.environ.get("STEPS", "200"))
h, w = bench.WORKLOAD["feat_hw"]
decs = [bench.build_decoder(dev)[2] for _ in range(NS)]
inputs = [bench.build_inputs(1, dev, seed=1000 + i) for i in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]

def run(nstreams, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            decs[k](*inputs[k], feat_hw=(h, w))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for ns in range(1, NS + 1):
    run(ns, 40)
for rep in range(3):
    for ns in range(1, NS + 1):
        ms = run(ns, steps)
        print("streams %d: %.4f ms per forward = %.0f decoder-iterations/s" % (ns, ms, 8 / ms * 1e3), flush=True)
