"""Forward time at BASELINE cfg 5 (20 views 960x1280 -> 240x320 features, N = 1 536 000 tokens, 512 queries, 12 iterations, d = 256)
in the attention mode given on the command line (default fp16, the arithmetic BASELINE.json names for that config).  Not the headline
metric (that is cfg 3); the numbers go into DESIGN.md."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[2] == "dev":          # development library (PARQ_* switches)
    from parq_amd import _lib  # noqa: E402
    _lib.use_dev_library()
from parq_amd import synth  # noqa: E402
from parq_amd.decoder import PARQDecoder  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "fp16"
I, Q, V, h, w = 12, 512, 20, 240, 320
dev = torch.device("cuda", 0)
cfg = synth.decoder_cfg(dim=256, queries=Q, heads=4, ffn=768, layers=I)
W = synth.make_decoder_weights(cfg, 551, damped=True)
dec = PARQDecoder(cfg).eval()
dec.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
dec = dec.to(dev)
dec.attention_mode = MODE
cam, T_cp, T_wp, T_wl = synth.make_geometry(552, 1, V, h, w)
g = torch.Generator(device=dev).manual_seed(553)
tokens = torch.randn(1, V * h * w, 256, device=dev, generator=g)
args = (tokens,) + tuple(torch.from_numpy(a).to(dev) for a in (cam, T_cp, T_wp, T_wl))
with torch.no_grad():
    for _ in range(2):
        dec(*args, feat_hw=(h, w))
    torch.cuda.synchronize()
    dec.profile_enable(True)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        dec(*args, feat_hw=(h, w))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    prof = dec.profile_read()
print(json.dumps({"metric": "decoder-iterations/sec at BASELINE cfg 5 (20 views, 240x320 features, 512 queries, 12 iterations, d=256)",
                  "value": I / dt, "unit": "decoder-iterations/sec", "ms_per_forward": dt * 1e3, "attention_mode": MODE, "n_gpus": 1, "steps": n,
                  "data": "synthetic", "kernel_groups_ms_per_forward": {k: v[0] / n for k, v in prof.items()}}))
