"""GPU box: attention mode "split8" against mode "split" on BASELINE cfg 3's synthetic workload while the cross-attention is sharpened
(the query projection scaled), with the guard switched off so that mode 4 runs whatever the rows look like: the largest difference of
the two modes' first-iteration outputs (mode "split" is 2e-6 from float64, so this IS mode 4's error to that accuracy), the state of
the guard's flag, and the smallest row sum the guard saw.  Where the flag is down, the difference must be small — that is the
guard's job.  SMOOTH=1: FPN-like smooth tokens (synth.make_scene(smooth=True)) instead of white noise — smooth features keep a larger error at
the same row sum (fewer distinct values behind a row).  usage: python tools/split8_guard_sweep.py [scale ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from parq_amd.decoder import PARQDecoder, OUTPUT_KEYS  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
scales = [float(x) for x in sys.argv[1:]] or [1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 6.0]
h, w = bench.WORKLOAD["feat_hw"]
for seed in (2024, 7):
    for sc in scales:
        from parq_amd import synth
        cfg = synth.decoder_cfg(dim=256, queries=256, heads=4, ffn=768, layers=1)
        W = synth.make_decoder_weights(cfg, seed=seed)
        key = "parq_module.decoder.layers.0.multihead_attn.in_proj_weight"
        wq = W[key].copy(); wq[:256] *= sc; W[key] = wq
        outs = {}
        flag = None
        for mode in ("split", "split8"):
            dec = PARQDecoder(cfg).eval()
            sd = dec.state_dict()
            for k in sd:
                sd[k] = torch.from_numpy(W[k.replace("parq_module.decoder.mlp_heads.", "mlp_heads.")]).reshape(sd[k].shape)
            dec.load_state_dict(sd, strict=True)
            dec = dec.to(dev)
            dec.attention_mode = mode
            dec.range_check = "off"
            inputs = bench.build_inputs(1, dev, seed=1000 + seed)
            if os.environ.get("SMOOTH"):
                scn = synth.make_scene(1000 + seed, 1, 10, h, w, 256, smooth=True)
                inputs = tuple(torch.from_numpy(scn[k]).to(dev) for k in ("tokens", "camera", "T_camera_pseudoCam", "T_world_pseudoCam", "T_world_local"))
            o = dec(*inputs, feat_hw=(h, w))[0]
            torch.cuda.synchronize()
            outs[mode] = {k: v.double().cpu() for k, v in o.items()}
            if mode == "split8":
                flag = dec.attention_too_peaked()
                lmin = dec.attention_min_row_sum()
        diff = max(float(((outs["split8"][k] - outs["split"][k]).abs() / outs["split"][k].abs().clamp(min=1)).max()) for k in outs["split"])
        print("weights seed %4d, W_q x %.1f: split8 vs split %.2e   smallest row sum %8.1f   guard flag %s"
              % (seed, sc, diff, lmin if lmin is not None else float("nan"), "UP (heads move to fp16 x 3)" if flag else "down"), flush=True)
