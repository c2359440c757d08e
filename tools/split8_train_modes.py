"""Development: one training step (forward_train + backward, same dropout seed) in attention modes split8 / split / fp32; pairwise
Frobenius-relative gradient differences.  Usage: python tools/split8_train_modes.py [pdrop] [iterations] [w] [B V h Q smooth]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parq_amd import synth
from gpu_util import make_decoder, scene_args

pdrop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
I = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = int(sys.argv[3]) if len(sys.argv) > 3 else 36
B, V, h, Q, heads, dim, ffn = 2, 2, 32, 24, 4, 256, 128
smooth = True
if len(sys.argv) > 8:
    B, V, h, Q, smooth = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8] == "1"
cfg = synth.decoder_cfg(dim=dim, queries=Q, heads=heads, ffn=ffn, layers=I, dropout=pdrop)
W = synth.make_decoder_weights(cfg, 71, damped=True)
sc = synth.make_scene(72, B, V, h, w, dim, smooth=smooth)
ncls = cfg.NUM_SEMCLS + 1
cots = {"pred_logits": synth.normal(73, "cl", (I, B, Q, ncls)), "center_unnormalized": synth.normal(74, "cc", (I, B, Q, 3)),
        "size_unnormalized": synth.normal(75, "cs", (I, B, Q, 3)), "ortho6d": synth.normal(76, "cr", (I, B, Q, 6))}
res = {}
for mode in ("split8", "split", "fp32"):
    dec = make_decoder(cfg, W)
    dec = dec.train() if pdrop > 0 else dec
    dec.attention_mode = mode
    dec.train_split8 = True
    torch.manual_seed(11)
    outs = dec.forward_train(*scene_args(sc), feat_hw=(h, w))
    grads, d_tok = dec.backward({k: torch.from_numpy(v) for k, v in cots.items()})
    torch.cuda.synchronize()
    res[mode] = ({k: v.cpu().numpy().astype(np.float64) for k, v in grads.items()}, d_tok.cpu().numpy().astype(np.float64),
                 np.concatenate([o[key].cpu().numpy().ravel() for o in outs for key in ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d")]))
for a, b in (("split8", "split"), ("split8", "fp32"), ("split", "fp32")):
    worst = ("", 0.0)
    for name, ga in res[a][0].items():
        gb = res[b][0][name]
        if np.abs(gb).max() == 0:
            continue
        worst = max(worst, (name, np.linalg.norm(ga - gb) / np.linalg.norm(gb)), key=lambda t: t[1])
    tok = np.linalg.norm(res[a][1] - res[b][1]) / np.linalg.norm(res[b][1])
    fwd = float((np.abs(res[a][2] - res[b][2]) / np.maximum(1, np.abs(res[b][2]))).max())
    rl = sorted(np.linalg.norm(ga - res[b][0][n]) / np.linalg.norm(res[b][0][n]) for n, ga in res[a][0].items() if np.abs(res[b][0][n]).max() > 0)
    print("median %.2e " % rl[len(rl) // 2], end="")
    print("p=%.2f I=%d w=%d  %-6s vs %-6s: outputs %.2e  gradients %.2e (%s)  d tokens %.2e" % (pdrop, I, w, a, b, fwd, worst[1], worst[0], tok))
