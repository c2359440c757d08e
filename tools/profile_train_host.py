"""GPU-box helper: host-side profile (cProfile) of the LOSS phase of the training step of `bench.py --train`, device idle at its
start (synchronised), i.e. what the device waits for between the forward and the backward.  usage: python tools/profile_train_host.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from parq_amd import Obb3D, PARQDecoder, Pose, synth  # noqa: E402

W_ = bench.WORKLOAD
dev = torch.device("cuda:0")
B, V, (h, w), Q, C, I = 4, W_["views"], W_["feat_hw"], W_["queries"], W_["dim"], W_["iters"]
cfg = synth.decoder_cfg(dim=C, queries=Q, heads=W_["heads"], ffn=W_["ffn"], layers=I, dropout=0.1)
Wt = synth.make_decoder_weights(cfg, 41, damped=True)
dec = PARQDecoder(cfg)
dec.load_state_dict({k: torch.from_numpy(v) for k, v in Wt.items()}, strict=False)
dec = dec.to(dev).train()
inputs = bench.build_inputs(B, dev, seed=2000)
obbs, sym = synth.make_boxes(3000, B, 12)
obbs, sym = Obb3D(torch.from_numpy(obbs).to(dev)), torch.from_numpy(sym).to(dev)
T_wl = Pose(inputs[4])
np.random.seed(1)
pr = cProfile.Profile()
tl, tb = [], []
for it in range(8):
    outs = dec(*inputs, feat_hw=(h, w))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if it >= 3:
        pr.enable()
    loss = dec.loss(outs, obbs, T_wl, sym)["total_loss"]
    torch.cuda.synchronize()
    pr.disable()
    t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tl.append((t1 - t0) * 1e3); tb.append((t2 - t1) * 1e3)
    dec.zero_grad(set_to_none=True)
print("loss ms", [round(x, 2) for x in tl], "backward ms", [round(x, 2) for x in tb])
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
