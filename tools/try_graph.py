"""Experiment: does replaying the forward as one hipGraph beat stream-ordered launches?  (125 dependent kernels per forward)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
device = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(device)
inputs = bench.build_inputs(1, device, seed=1000)
h, w = bench.WORKLOAD["feat_hw"]
def run():
    return dec(*inputs, feat_hw=(h, w))
with torch.no_grad():
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    print("stream-ordered: %.4f ms per forward" % ((time.perf_counter() - t0) / 20 * 1e3))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            outs = run()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        print("graph replay:   %.4f ms per forward" % ((time.perf_counter() - t0) / 20 * 1e3))
    except Exception as e:                                   # noqa: BLE001
        print("graph capture failed:", repr(e)[:300])
