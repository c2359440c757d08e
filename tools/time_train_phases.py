"""Wall-clock phases of the bench.py --train step (each phase closed by a device synchronisation)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from parq_amd import Obb3D, PARQDecoder, Pose, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Wk = bench.WORKLOAD
V, (h, w), Q, C, I = Wk["views"], Wk["feat_hw"], Wk["queries"], Wk["dim"], Wk["iters"]
cfg = synth.decoder_cfg(dim=C, queries=Q, heads=Wk["heads"], ffn=Wk["ffn"], layers=I, dropout=0.1)
Wt = synth.make_decoder_weights(cfg, 41, damped=True)
dec = PARQDecoder(cfg)
dec.load_state_dict({k: torch.from_numpy(v) for k, v in Wt.items()}, strict=False)
device = torch.device("cuda", 0)
dec = dec.to(device).train()
inputs = bench.build_inputs(B, device, seed=2000)
obbs, sym = synth.make_boxes(3000, B, 12)
obbs, sym = Obb3D(torch.from_numpy(obbs).to(device)), torch.from_numpy(sym).to(device)
T_wl = Pose(inputs[4])
opt = torch.optim.AdamW([p for p in dec.parameters() if p.requires_grad], lr=1e-5, foreach=True)
np.random.seed(1)
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(5):
    t0 = sync()
    opt.zero_grad(set_to_none=True)
    outs = dec(*inputs, feat_hw=(h, w)); t1 = sync()
    loss = dec.loss(outs, obbs, T_wl, sym)["total_loss"]; t2 = sync()
    loss.backward(); t3 = sync()
    torch.nn.utils.clip_grad_norm_(dec.parameters(), 1.0); t4 = sync()
    opt.step(); t5 = sync()
    print("forward %.2f  loss %.2f  backward %.2f  clip %.2f  adamw %.2f  total %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)))
