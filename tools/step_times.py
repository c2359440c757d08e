"""GPU box: per-forward times (hipEvents) of 16 consecutive forwards after a synchronisation, three repetitions — the first forward of a
process loads the code objects, the next ~12 run 2-12 % slow while the clock / power controller settles (why bench.py spins the
device up before its warm-up steps)."""
import os, sys, torch
torch.set_grad_enabled(False)      # inference tool: the reference's drivers run these calls under no_grad (eval.py:46)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
device = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(device)
inputs = bench.build_inputs(1, device, seed=1000)
h, w = bench.WORKLOAD["feat_hw"]
with torch.no_grad():
    for rep in range(3):
        n = 16
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        torch.cuda.synchronize()
        ev[0].record()
        for i in range(n):
            dec(*inputs, feat_hw=(h, w))
            ev[i + 1].record()
        torch.cuda.synchronize()
        print("rep", rep, " ".join("%.3f" % ev[i].elapsed_time(ev[i + 1]) for i in range(n)))
