"""parq_amd.InFlight at BASELINE cfg 2 by depth (forwards outstanding) and policy: is the second forward's overlap limited by the pipeline
(depth), by the never-NaN check in Ticket.result() (policy), or by the device running two launch-bound chains at once?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from parq_amd import InFlight
torch.set_grad_enabled(False)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
conf = bench.CONFIGS[name]
if "dim" in conf:
    bench.WORKLOAD["dim"] = conf["dim"]
bench.WORKLOAD.update({k: conf[k] for k in ("views", "image_hw", "feat_hw", "queries", "iters")})
dev = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(dev)
if conf["mode"]:
    dec.attention_mode = conf["mode"]
h, w = bench.WORKLOAD["feat_hw"]
I = bench.WORKLOAD["iters"]
pair = [bench.build_inputs(1, dev, 1000 + 4000 * i) for i in range(4)]
for policy in ("sync", "lazy"):
    dec.range_check = policy
    t0 = time.perf_counter()
    for i in range(50):
        dec(*pair[i & 3], feat_hw=(h, w))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(400):
        dec(*pair[i & 3], feat_hw=(h, w))
    torch.cuda.synchronize(); one = (time.perf_counter() - t0) / 400
    line = "%s policy %-4s one at a time %.4f ms (%.0f it/s)" % (name, policy, one * 1e3, I / one)
    for depth in (2, 3, 4):
        runner = InFlight(dec, depth=depth)
        def go(n):
            tickets = []
            for i in range(n):
                tickets.append(runner.submit(*pair[i & 3], feat_hw=(h, w)))
                if len(tickets) == depth:
                    tickets.pop(0).result()
            for t in tickets:
                t.result()
        go(40); torch.cuda.synchronize()
        t0 = time.perf_counter(); go(400); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
        line += " | depth %d: %.4f ms (%.0f it/s)" % (depth, dt * 1e3, I / dt)
    print(line)
