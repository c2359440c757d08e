"""How long the HOST needs to enqueue one inference forward (one parq_forward call = ~90 launches + the Python around it): a tiny scene
(the device finishes a forward faster than the host enqueues it, so wall time per forward = host time), and the benchmark scene with the
queue never drained inside the loop."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from parq_amd import synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg, W, dec = bench.build_decoder(dev)
h, w = bench.WORKLOAD["feat_hw"]
big = bench.build_inputs(1, dev, seed=1000)
sc = synth.make_scene(5, 1, 2, 16, 24, 256, smooth=True)
small = tuple(torch.from_numpy(sc[k]).to(dev) for k in ("tokens", "camera", "T_camera_pseudoCam", "T_world_pseudoCam", "T_world_local"))
dec.range_check = "off"
for name, inp, hw in (("tiny scene (2 views 16x24)", small, (16, 24)), ("cfg 3 scene", big, (h, w))):
    for _ in range(30):
        dec(*inp, feat_hw=hw)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            dec(*inp, feat_hw=hw)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s: host loop %.3f ms per forward, wall %.3f ms per forward" % (name, (t1 - t0) * 5, (t2 - t0) * 5), flush=True)
