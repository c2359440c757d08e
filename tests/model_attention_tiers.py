"""Test infrastructure (run by hand: python tests/model_attention_tiers.py g15_cfg5_shape [scales...]): CPU model of candidate
cross-attention arithmetics ("tiers") between mode "split8" (fp16 hi.hi + MX-fp8 cross terms, one fp16 P V product) and mode "split"
(three fp16 products everywhere) on a reference fixture whose cross-attention is sharpened step by step (query projection x 1 .. x 6).
The table it prints is profiles/r05_tier_model_*.txt; the tier the guard falls back to was chosen from it BEFORE its kernel existed.

Every tier is described by (score arithmetic, probability form, value form):
  scores  "s8"   hi16.hi16 + e4m3(q) e4m3(k_lo) + e4m3(q_lo) e4m3(k)                      (mode 4)
          "s16"  hi16.hi16 + hi16.lo16 + lo16.hi16 (three fp16 products, fp32 accumulate)  (mode 1)
          "s16a" hi16.hi16 + q_lo16.k_hi16 (fp16) + e4m3(q) e4m3(k_lo)                    (one fp16 cross term, one MX)
  P       "p16"  fp16 round-to-nearest, the row sum over the same rounded values
          "p16+8" p16 + e4m3 of the residual (MX term against the values' hi form)
          "p2"   hi16 + lo16
  V       "v16"  fp16 round-to-nearest
          "v16+8" v16 + e4m3 of the residual (MX term)
          "v2"   hi16 + lo16
"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F

from parq_amd import synth
from oracle import parq_oracle as O
import golden_util as G
import emulate_attention_arithmetic as E

torch.set_grad_enabled(False)

TIERS = {
    "split8 (s8,p16,v16)": ("s8", "p16", "v16"),
    "A (s16,p16,v16)": ("s16", "p16", "v16"),
    "B (s16,p16,v16+8)": ("s16", "p16", "v16+8"),
    "C (s16,p16+8,v16+8)": ("s16", "p16+8", "v16+8"),
    "D (s16,p16,v2)": ("s16", "p16", "v2"),
    "F (s16a,p16,v16)": ("s16a", "p16", "v16"),
    "G (s8,p16+8,v16+8)": ("s8", "p16+8", "v16+8"),
    "split (s16,p2,v2)": ("s16", "p2", "v2"),
}


def _rn16(x):
    return x.float().half().double()


def scores(q, kt, form):
    if form == "s8":
        return E.product(q, kt, "split8")
    if form == "s16":
        return E.product(q, kt, "split")
    if form == "s16a":
        a, b = q.float().double(), kt.float().double()
        ah, bh = E._rtz16(a), E._rtz16(b)
        al, bl = a - ah, b - bh
        return ah @ bh + _rn16(al) @ bh + E._e4(a) @ E._e4(bl, 1024.0)
    raise ValueError(form)


def pv(p, v, pform, vform):
    """un-normalised output and the matching row sum"""
    p, v = p.float().double(), v.float().double()
    if pform == "p2":
        ph = E._rtz16(p)
        pl = _rn16(p - ph)
        psum = p.sum(-1, keepdim=True)
    else:
        ph = _rn16(p)
        pl = E._e4(p - ph, 2.0 ** 16) if pform == "p16+8" else None
        psum = ph.sum(-1, keepdim=True) + (pl.sum(-1, keepdim=True) if pl is not None else 0.0)
    if vform == "v2":
        vh = E._rtz16(v)
        vl = _rn16(v - vh)
    else:
        vh = _rn16(v)
        vl = E._e4(v - vh, 2.0 ** 12) if vform == "v16+8" else None
    o = ph @ vh
    if vl is not None:
        # an MX term multiplies the e4m3 form of the other operand; an fp16 lo plane multiplies the fp16 hi form
        o = o + (ph @ vl if vform == "v2" else E._e4(p, 64.0) @ vl)
    if pl is not None:
        o = o + (pl @ vh if pform == "p2" else pl @ E._e4(v))
    return o, psum


STAT = {}


def mha_variant(tier):
    def mha(query, key, value, in_w, in_b, out_w, out_b, H, reference_ops=False):
        B, L, C = query.shape
        S = key.shape[1]
        if S <= 1024:
            return E._exact(query, key, value, in_w, in_b, out_w, out_b, H, reference_ops)
        dh = C // H
        q = F.linear(query, in_w[:C], in_b[:C]).view(B, L, H, dh).transpose(1, 2)
        k = F.linear(key, in_w[C:2 * C], in_b[C:2 * C]).view(B, S, H, dh).transpose(1, 2)
        v = F.linear(value, in_w[2 * C:], in_b[2 * C:]).view(B, S, H, dh).transpose(1, 2)
        qs = q * (math.log2(math.e) / math.sqrt(dh))
        if tier is None:
            s = qs @ k.transpose(-1, -2)
            p = torch.exp2(s - s.max(-1, keepdim=True).values)
            STAT.setdefault("l", []).append(p.sum(-1).flatten())
            o = (p / p.sum(-1, keepdim=True)) @ v
        else:
            sform, pform, vform = tier
            s = scores(qs, k.transpose(-1, -2), sform)
            p = torch.exp2(s - s.max(-1, keepdim=True).values).float().double()
            o, psum = pv(p, v, pform, vform)
            o = o / psum
        return F.linear(o.transpose(1, 2).reshape(B, L, C), out_w, out_b)
    return mha


def main():
    name = sys.argv[1]
    scales = [float(x) for x in sys.argv[2:]] or [1.0, 2.0, 3.0, 4.0, 6.0]
    n_iter = int(os.environ.get("TIER_ITERS", "1"))
    only = os.environ.get("TIERS")                      # e.g. TIERS="split8,C,split": leading words of the tier names
    tiers = {k: v for k, v in TIERS.items() if only is None or k.split()[0] in only.split(",")}
    case, z = G.load(name)
    cfg, W0, sc = G.inputs(case)
    refs = G.forced_refs(z, cfg.TRANSFORMER.SCALE)
    print("# fixture %s, keys %s, %d teacher-forced iteration(s); errors = largest |a-b|/max(1,|b|) over the decoder outputs vs float64"
          % (name, tuple(sc["tokens"].shape), n_iter), flush=True)
    for scale in scales:
        W = dict(W0)
        key = "parq_module.decoder.layers.0.multihead_attn.in_proj_weight"
        C = W[key].shape[1]
        w = W[key].copy()
        w[:C] *= scale
        W[key] = w
        od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
        od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
        STAT.clear()
        O.mha = mha_variant(None)
        exact = [od.iterate(torch.from_numpy(refs[k]).double(), k)[0] for k in range(n_iter)]
        l = torch.cat(STAT["l"])
        line = "W_q x %.1f: row sum min %.1f median %.0f |" % (scale, float(l.min()), float(l.median()))
        for tname, tier in tiers.items():
            O.mha = mha_variant(tier)
            worst = 0.0
            for k in range(n_iter):
                out = od.iterate(torch.from_numpy(refs[k]).double(), k)[0]
                for kk in G.KEYS:
                    worst = max(worst, float(((out[kk] - exact[k][kk]).abs() / exact[k][kk].abs().clamp(min=1)).max()))
            line += " %s %.2e |" % (tname, worst)
        O.mha = E._exact
        print(line, flush=True)


if __name__ == "__main__":
    main()
