"""Dev tool: error of both attention modes vs the fp64 oracle on a synthetic scene (iteration 0)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from parq_amd import synth
from oracle import parq_oracle as O
from gpu_util import make_decoder, scene_args, dev

V, h, w, Q = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (10, 120, 160, 256))]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
cfg = synth.decoder_cfg(dim=256, queries=Q, heads=4, ffn=768, layers=2)
W = synth.make_decoder_weights(cfg, 41)
sc = synth.make_scene(42, B, V, h, w, 256)
od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=torch.float64)
t0 = time.time()
od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
ref0 = od.initial_ref().float().double()      # the float32-rounded points both sides consume
with torch.no_grad():
    o64, _, i64 = od.iterate(ref0, 0)
print("oracle fp64 %.1fs" % (time.time() - t0))
def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
for mode in ("fp32", "split"):
    dec = make_decoder(cfg, W); dec.attention_mode = mode
    dec.prepare(*scene_args(sc))
    out, _ = dec.iterate(0, dev(ref0.float().numpy()))
    torch.cuda.synchronize()
    Cn = 256
    g = lambda n: dec.intermediate(n).view(B, Q, Cn).cpu().numpy()
    print(mode, "tgt %.2e pos %.2e | logits %.2e center %.2e rot %.2e size(rel) %.2e  range_flag=%s"
          % (err(g("tgt"), i64["tgt"]), err(g("pos_feat"), i64["pos"]), err(out["pred_logits"].cpu(), o64["pred_logits"]),
             err(out["center_unnormalized"].cpu(), o64["center_unnormalized"]), err(out["ortho6d"].cpu(), o64["ortho6d"]),
             float((np.abs(out["size_unnormalized"].cpu().numpy() - o64["size_unnormalized"].numpy()) / np.maximum(1, o64["size_unnormalized"].abs().numpy())).max()),
             dec.fp16_range_exceeded()))
