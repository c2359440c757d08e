"""GPU box: socket power and shader clock (sysfs, in-process) while ONE phase of the training step loops back to back at the
BASELINE cfg-4 per-GPU shard (4 scenes, 10 views, 256 queries, 8 iterations, dropout 0.1): the training forward alone, the HIP
backward alone (repeated on one stash through the direct API).  Tells which phases sit at the 1400 W cap."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from parq_amd import PARQDecoder, synth  # noqa: E402

dev = torch.device("cuda", 0)
B = 4
V, (h, w), Q, C, I = bench.WORKLOAD["views"], bench.WORKLOAD["feat_hw"], bench.WORKLOAD["queries"], bench.WORKLOAD["dim"], bench.WORKLOAD["iters"]
cfg = synth.decoder_cfg(dim=C, queries=Q, heads=4, ffn=768, layers=I, dropout=0.1)
W = synth.make_decoder_weights(cfg, 41, damped=True)
dec = PARQDecoder(cfg)
dec.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
dec = dec.to(dev).train()
inputs = bench.build_inputs(B, dev, seed=2000)
cots = {k: torch.randn(I, B, Q, wd, device=dev) * 1e-2 for k, wd in (("pred_logits", 10), ("center_unnormalized", 3), ("size_unnormalized", 3), ("ortho6d", 6))}


def fwd():
    dec.forward_train(*inputs, feat_hw=(h, w))


def bwd():
    dec.backward(cots, want_token_grad=False)


for _ in range(3):
    fwd(); bwd()
torch.cuda.synchronize()
for name, fn in (("training forward", fwd), ("HIP backward", bwd)):
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    st = bench.device_state_under_load(fn, seconds=2.0)
    print("%-18s %.2f ms per call; looping: %s" % (name, ms, {k: round(v, 1) for k, v in (st or {}).items() if k in ("sclk_mhz", "socket_power_w")}), flush=True)
