"""Host time of parq_amd.InFlight at BASELINE cfg 2 (0.38 ms of device time per forward: the host-bound case), cProfile."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from parq_amd import InFlight
torch.set_grad_enabled(False)
conf = bench.CONFIGS["cfg2"]
bench.WORKLOAD.update({k: conf[k] for k in ("views", "image_hw", "feat_hw", "queries", "iters")})
dev = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(dev)
dec.attention_mode = "bf16"
h, w = bench.WORKLOAD["feat_hw"]
pair = [bench.build_inputs(1, dev, 1000), bench.build_inputs(1, dev, 5000)]
runner = InFlight(dec, depth=2)
def go(n):
    tickets = []
    for i in range(n):
        tickets.append(runner.submit(*pair[i & 1], feat_hw=(h, w)))
        if len(tickets) == 2:
            tickets.pop(0).result()
    for t in tickets:
        t.result()
go(20); torch.cuda.synchronize()
t0 = time.perf_counter(); go(400); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("in flight: %.4f ms per forward" % (dt / 400 * 1e3))
t0 = time.perf_counter()
for i in range(400):
    dec(*pair[i & 1], feat_hw=(h, w))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("one at a time: %.4f ms per forward" % (dt / 400 * 1e3))
for graph in (False, True, False, True):
    dec.use_graph = graph
    go(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(400); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("use_graph %s: in flight %.4f ms per forward" % (graph, dt / 400 * 1e3))
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable(); go(400); torch.cuda.synchronize(); pr.disable()
    out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22); print(out.getvalue()[:6000])
