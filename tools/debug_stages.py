"""Dev tool: per-stage error of one teacher-forced iteration vs the fp64 and fp32 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from parq_amd import synth
from oracle import parq_oracle as O
import golden_util as G
from gpu_util import make_decoder, scene_args, dev

name = sys.argv[1] if len(sys.argv) > 1 else "g2_forced"
it = int(sys.argv[2]) if len(sys.argv) > 2 else 0
case, z = G.load(name)
cfg, W, sc = G.inputs(case)
dec = make_decoder(cfg, W)
dec.prepare(*scene_args(sc))
refs = G.forced_refs(z, cfg.TRANSFORMER.SCALE)
out, _ = dec.iterate(it, dev(refs[it]))
torch.cuda.synchronize()
B, Q, Cn = case["B"], cfg.NUM_QUERIES, cfg.DIM_IN
res = {}
for dt in (torch.float64, torch.float32):
    od = O.OracleDecoder(cfg, W, synth.SCANNET_MEAN_SIZES, dtype=dt)
    od.prepare(sc["tokens"], sc["camera"], sc["T_camera_pseudoCam"], sc["T_world_pseudoCam"], sc["T_world_local"])
    with torch.no_grad():
        o, _, inter = od.iterate(torch.from_numpy(refs[it]).to(dt), it)
    res[dt] = (od, o, inter)
od64, o64, i64 = res[torch.float64]
od32, o32, i32 = res[torch.float32]
def err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max())
print("stage            mine-vs-fp64   oracle32-vs-fp64   mine-vs-oracle32")
T = dec.intermediate("T_camera_local_f64").view(torch.float64).view(B, -1, 12).cpu().numpy()
print("T_cl        %12.3e %12.3e %12.3e" % (err(T, od64.T_cl), err(od32.T_cl, od64.T_cl), err(T, od32.T_cl)))
for nm, key in (("tgt", "tgt"), ("pos_feat", "pos")):
    m = dec.intermediate(nm).view(B, Q, Cn).cpu().numpy()
    print("%-10s  %12.3e %12.3e %12.3e" % (nm, err(m, i64[key]), err(i32[key], i64[key]), err(m, i32[key])))
    if nm == "tgt":
        e = np.abs(m - i64[key].numpy()).max(-1)
        bad = np.argwhere(e > 1e-3)
        print("   tgt rows with err>1e-3:", bad[:10].tolist(), "valid counts", i64["valid"].sum(1)[tuple(bad[:10].T)].tolist() if len(bad) else "")
for k in o64:
    m = out[k].cpu().numpy()
    print("%-22s %10.3e %10.3e %10.3e   golden: %10.3e" % (k, err(m, o64[k]), err(o32[k], o64[k]), err(m, o32[k]), err(m, z["it%d_%s" % (it, k)])))
