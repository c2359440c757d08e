"""cProfile of the batched set loss at the cfg-4 per-GPU shard (host-side cost; the device work is a few dozen tiny launches)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parq_amd import Obb3D, PARQDecoder, Pose, synth
B, Q, I = 4, 256, 8
dev = "cuda"
cfg = synth.decoder_cfg(dim=256, queries=Q, heads=4, ffn=768, layers=I, dropout=0.1)
dec = PARQDecoder(cfg)
outs = [{"pred_logits": torch.randn(B, Q, 10, device=dev, requires_grad=True), "center_unnormalized": torch.randn(B, Q, 3, device=dev, requires_grad=True),
         "size_unnormalized": torch.rand(B, Q, 3, device=dev, requires_grad=True), "ortho6d": torch.randn(B, Q, 6, device=dev, requires_grad=True),
         "coord_pos": torch.randn(B, Q, 3, device=dev)} for _ in range(I)]
obbs, sym = synth.make_boxes(1, B, 12)
obbs, sym = Obb3D(torch.from_numpy(obbs).to(dev)), torch.from_numpy(sym).to(dev)
T_wl = Pose(torch.from_numpy(synth.make_geometry(2, B, 2, 8, 8)[3]).to(dev))
np.random.seed(0)
for _ in range(3):
    dec.loss(outs, obbs, T_wl, sym)["total_loss"].backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    l = dec.loss(outs, obbs, T_wl, sym)["total_loss"]
torch.cuda.synchronize()
print("loss forward: %.2f ms" % ((time.perf_counter() - t0) * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    l = dec.loss(outs, obbs, T_wl, sym)["total_loss"]
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
