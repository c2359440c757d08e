"""Soak: parq_amd.InFlight(depth=3) for N submits over 3 input sets at a bench configuration, EVERY ticket's outputs compared bit for bit with
the serial forward of its inputs; mode switches (split8 <-> split) and a weight update every few hundred submits force re-captures."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from parq_amd import InFlight
torch.set_grad_enabled(False)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
conf = bench.CONFIGS[name]
if "dim" in conf:
    bench.WORKLOAD["dim"] = conf["dim"]
bench.WORKLOAD.update({k: conf[k] for k in ("views", "image_hw", "feat_hw", "queries", "iters")})
dev = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(dev)
h, w = bench.WORKLOAD["feat_hw"]
sets = [bench.build_inputs(1, dev, 1000 + 3000 * i) for i in range(3)]
KEYS = ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d", "sem_cls_prob", "coord_pos")


def serial():
    return [[{k: o[k].clone() for k in KEYS} for o in dec(*s, feat_hw=(h, w))] for s in sets]


want = serial()
torch.cuda.synchronize()
runner = InFlight(dec, depth=3)
bad = 0
t0 = time.perf_counter()
tickets = []
modes = [m for m in ("split8", "split") if name != "shipped"] or ["split"]
for i in range(N):
    if i and i % 400 == 0:                       # drain, change something that invalidates the captured graphs, new expectations
        for j, t in tickets:
            out = t.result()
            bad += not all(torch.equal(a[k], b[k]) for a, b in zip(out, want[j]) for k in KEYS)
        tickets = []
        if (i // 400) % 2:
            dec.attention_mode = modes[(i // 800) % len(modes)]
        else:
            with torch.no_grad():
                next(iter(dec.parameters())).mul_(1.0001)
        want = serial()
        torch.cuda.synchronize()
    tickets.append((i % 3, runner.submit(*sets[i % 3], feat_hw=(h, w))))
    if len(tickets) == 3:
        j, t = tickets.pop(0)
        out = t.result()
        bad += not all(torch.equal(a[k], b[k]) for a, b in zip(out, want[j]) for k in KEYS)
for j, t in tickets:
    out = t.result()
    bad += not all(torch.equal(a[k], b[k]) for a, b in zip(out, want[j]) for k in KEYS)
torch.cuda.synchronize()
print("%s: %d submits with three outstanding in %.1f s, %d tickets differ from the serial forward, attention mode at the end %s, finite %s"
      % (name, N, time.perf_counter() - t0, bad, dec.attention_mode, all(torch.isfinite(o[k]).all().item() for o in out for k in KEYS)))
