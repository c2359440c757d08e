"""Host-side cost of the set loss at the cfg-4 per-GPU shard (4 scenes, 8 iterations, 256 queries, 12 boxes)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parq_amd import Obb3D, Pose, synth
from parq_amd.loss import HungarianMatcherModified, decoder_loss
B, Q, I = 4, 256, 8
dev = "cuda"
outs = [{"pred_logits": torch.randn(B, Q, 10, device=dev, requires_grad=True), "center_unnormalized": torch.randn(B, Q, 3, device=dev, requires_grad=True),
         "size_unnormalized": torch.rand(B, Q, 3, device=dev, requires_grad=True), "ortho6d": torch.randn(B, Q, 6, device=dev, requires_grad=True),
         "coord_pos": torch.randn(B, Q, 3, device=dev)} for _ in range(I)]
obbs, sym = synth.make_boxes(1, B, 12)
obbs, sym = Obb3D(torch.from_numpy(obbs).to(dev)), torch.from_numpy(sym).to(dev)
T_wl = Pose(torch.from_numpy(synth.make_geometry(2, B, 2, 8, 8)[3]).to(dev))
cw = torch.ones(10); cw[9] = 0.1
m = HungarianMatcherModified(2, 0.25)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l = decoder_loss(outs, obbs, T_wl, sym, matcher=m, loss_weight=[5., 5., 5., 1.], num_semcls=9, class_weight=cw)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    l["total_loss"].backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("loss forward %.1f ms, backward %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
