"""Dev tool (GPU box): cfg-3-size backward vs float64 oracle autograd, error broken down by tensor and, for the cross-attention
in-projection, by q / k / v block; variants: default, PARQ_KVPROJ_BWD=fp32, PARQ_BWD_BATCHED=0, attention_mode=fp32."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from parq_amd import synth
from oracle import parq_oracle as O
from gpu_util import make_decoder, scene_args

V, FH, FW, Q, DIM, I = 10, int(os.environ.get("FH", 120)), int(os.environ.get("FW", 160)), 256, 256, 2
GKEYS = ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d")
cfg = synth.decoder_cfg(dim=DIM, queries=Q, heads=4, ffn=768, layers=I, dropout=0.0)
W = synth.make_decoder_weights(cfg, 441, damped=True)
sc = synth.make_scene(442, 1, V, FH, FW, DIM, smooth=True)
cots = {"pred_logits": synth.normal(443, "cl", (I, 1, Q, 10)), "center_unnormalized": synth.normal(444, "cc", (I, 1, Q, 3)),
        "size_unnormalized": synth.normal(445, "cs", (I, 1, Q, 3)), "ortho6d": synth.normal(446, "cr", (I, 1, Q, 6))}
t0 = time.time()
od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
for k in od.W:
    od.W[k].requires_grad_(True)
od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
od.tokens.requires_grad_(True)
ref = od.initial_ref(); loss = 0.0
for k in range(I):
    out, nxt, _ = od.iterate(ref, k)
    for key in GKEYS:
        loss = loss + (out[key] * torch.from_numpy(cots[key][k]).double()).sum()
    ref = nxt.detach()
loss.backward()
want = {k: v.grad.numpy() for k, v in od.W.items() if v.grad is not None}
print("oracle %.1fs" % (time.time() - t0))
name = "parq_module.decoder.layers.0.multihead_attn.in_proj_"
for tag, env, mode in (("default", {}, None), ("kvproj_bwd=fp32", {"PARQ_KVPROJ_BWD": "fp32"}, None), ("unbatched", {"PARQ_BWD_BATCHED": "0"}, None),
                       ("fp32 mode", {}, "fp32")):
    if tag == "kvproj_bwd=fp32":
        continue          # static env read: needs its own process (run with PARQ_KVPROJ_BWD=fp32 set outside)
    for k2, v in env.items():
        os.environ[k2] = v
    dec = make_decoder(cfg, W).train()
    if mode:
        dec.attention_mode = mode
    dec.forward_train(*scene_args(sc), feat_hw=(FH, FW))
    grads, d_tok = dec.backward({k: torch.from_numpy(v) for k, v in cots.items()})
    for k2 in env:
        os.environ.pop(k2)
    rows = []
    for n, g in grads.items():
        if n in want:
            d = g.cpu().numpy().astype(np.float64) - want[n]
            rows.append((np.linalg.norm(d) / max(np.linalg.norm(want[n]), 1e-30), n))
    rows.sort(reverse=True)
    print("==", tag, "(PARQ_KVPROJ_BWD=%s)" % os.environ.get("PARQ_KVPROJ_BWD"), " worst:", ["%s %.2e" % (n.split("decoder.")[-1], e) for e, n in rows[:4]])
    for suffix in ("weight", "bias"):
        g = grads[name + suffix].cpu().numpy().astype(np.float64); w = want[name + suffix]
        for bi, bn in enumerate("qkv"):
            a, b = g[bi * DIM:(bi + 1) * DIM], w[bi * DIM:(bi + 1) * DIM]
            print("   in_proj_%s[%s]: |want| %.3e  |err| %.3e  rel %.2e" % (suffix, bn, np.linalg.norm(b), np.linalg.norm(a - b), np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)))
    wt = od.tokens.grad.numpy(); dt = d_tok.cpu().numpy().astype(np.float64) - wt
    print("   tokens rel %.2e" % (np.linalg.norm(dt) / np.linalg.norm(wt)))
    del dec; torch.cuda.empty_cache()
