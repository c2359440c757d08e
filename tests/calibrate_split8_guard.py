"""Test infrastructure (run by hand: python tests/calibrate_split8_guard.py g15_cfg5_shape): the CPU model of attention mode 4 on a
reference fixture whose cross-attention is sharpened step by step (query projection x 1 .. x 4), beside the statistics the guard
of that mode can see — how concentrated a row is against what the mode's arithmetic does to the decoder outputs.  The table it
prints is profiles/r04_split8_peakedness_calibration.txt; the guard threshold (row probability sum < 64 -> fall back to "split")
comes from it."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, math
import torch.nn.functional as F
from parq_amd import synth
from oracle import parq_oracle as O
import golden_util as G
import emulate_attention_arithmetic as E
torch.set_grad_enabled(False)
STAT = {}
def mha_variant(pmode):
    def mha(query, key, value, in_w, in_b, out_w, out_b, H, reference_ops=False):
        B, L, C = query.shape
        S = key.shape[1]
        if S <= 1024:
            return E._exact(query, key, value, in_w, in_b, out_w, out_b, H, reference_ops)
        dh = C // H
        q = F.linear(query, in_w[:C], in_b[:C]).view(B, L, H, dh).transpose(1, 2)
        k = F.linear(key, in_w[C:2 * C], in_b[C:2 * C]).view(B, S, H, dh).transpose(1, 2)
        v = F.linear(value, in_w[2 * C:], in_b[2 * C:]).view(B, S, H, dh).transpose(1, 2)
        if pmode == "exact":
            s = (q * (math.log2(math.e) / math.sqrt(dh))) @ k.transpose(-1, -2)
            p = torch.exp2(s - s.max(-1, keepdim=True).values)
            l = p.sum(-1)
            w = p / p.sum(-1, keepdim=True)
            STAT.setdefault("l", []).append(l.flatten())
            STAT.setdefault("neff", []).append((1.0 / (w * w).sum(-1)).flatten())
            STAT.setdefault("smax", []).append((s.max(-1).values - s.mean(-1)).flatten())
            o = w @ v
        else:
            s = E.product(q * (math.log2(math.e) / math.sqrt(dh)), k.transpose(-1, -2), "split8")
            p = torch.exp2(s - s.max(-1, keepdim=True).values).float().double()
            if pmode == "full":
                o = E.product(p, v, "split8", a_scale8=64.0) / p.sum(-1, keepdim=True)
            else:
                # the decoder's kernel: probabilities and values as ONE fp16 value each (round to nearest), normaliser over the same values
                ph = p.float().half().double()
                o = (ph @ v.float().half().double()) / ph.sum(-1, keepdim=True)
        return F.linear(o.transpose(1, 2).reshape(B, L, C), out_w, out_b)
    return mha
name = sys.argv[1]
case, z = G.load(name)
cfg, W0, sc = G.inputs(case)
refs = G.forced_refs(z, cfg.TRANSFORMER.SCALE)
print("keys:", sc["tokens"].shape)
for scale in (1.0, 1.5, 2.0, 3.0, 4.0):
    W = dict(W0)
    key = "parq_module.decoder.layers.0.multihead_attn.in_proj_weight"
    C = W[key].shape[1]
    w = W[key].copy(); w[:C] *= scale; W[key] = w
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
    od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
    n = 1
    STAT.clear()
    O.mha = mha_variant("exact")
    exact = [od.iterate(torch.from_numpy(refs[k]).double(), k)[0] for k in range(n)]
    l = torch.cat(STAT["l"]); ne = torch.cat(STAT["neff"]); sm = torch.cat(STAT["smax"])
    res = {}
    for pmode in ("full", "p16"):
        O.mha = mha_variant(pmode)
        worst = 0
        for k in range(n):
            out = od.iterate(torch.from_numpy(refs[k]).double(), k)[0]
            for kk in G.KEYS:
                worst = max(worst, float(((out[kk] - exact[k][kk]).abs() / exact[k][kk].abs().clamp(min=1)).max()))
        res[pmode] = worst
    O.mha = E._exact
    print("W_q x %.1f: row sum l (relative to the row max): min %.1f median %.0f | effective keys min %.0f median %.0f | max - mean score (log2) median %.1f max %.1f | split8 %.2e  p16 %.2e"
          % (scale, float(l.min()), float(l.median()), float(ne.min()), float(ne.median()), float(sm.median()), float(sm.max()), res["full"], res["p16"]), flush=True)
