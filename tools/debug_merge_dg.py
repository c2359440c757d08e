"""Dev tool: training forward + backward with PARQ_MERGE_DG = 8 / 16 in separate processes; per-tensor gradient differences."""
import os, subprocess, sys, pickle
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
    import numpy as np, torch
    from parq_amd import synth
    from gpu_util import make_decoder, scene_args
    B, V, h, w, Q, heads, dim, ffn, layers = 2, 2, 32, 41, 24, 4, 256, 128, 3
    cfg = synth.decoder_cfg(dim=dim, queries=Q, heads=heads, ffn=ffn, layers=layers, dropout=0.0)
    W = synth.make_decoder_weights(cfg, 71)
    sc = synth.make_scene(72, B, V, h, w, dim, smooth=True)
    cots = {"pred_logits": synth.normal(73, "cl", (layers, B, Q, 10)), "center_unnormalized": synth.normal(74, "cc", (layers, B, Q, 3)),
            "size_unnormalized": synth.normal(75, "cs", (layers, B, Q, 3)), "ortho6d": synth.normal(76, "cr", (layers, B, Q, 6))}
    dec = make_decoder(cfg, W); dec.attention_mode = "fp32"
    outs = dec.forward_train(*scene_args(sc))
    grads, d_tok = dec.backward({k: torch.from_numpy(v) for k, v in cots.items()})
    res = {"outs": [{k: v.cpu().numpy() for k, v in o.items()} for o in outs], "grads": {k: v.cpu().numpy() for k, v in grads.items()}}
    pickle.dump(res, open(sys.argv[1], "wb"))
else:
    import numpy as np
    for dg in ("8", "16"):
        subprocess.check_call([sys.executable, __file__, "/tmp/dg%s.pkl" % dg], env=dict(os.environ, PARQ_MERGE_DG=dg))
    a, b = pickle.load(open("/tmp/dg8.pkl", "rb")), pickle.load(open("/tmp/dg16.pkl", "rb"))
    for k in range(3):
        print("iter", k, {key: float(np.abs(a["outs"][k][key] - b["outs"][k][key]).max()) for key in a["outs"][k]})
    rows = sorted(((np.linalg.norm(a["grads"][n] - b["grads"][n]) / max(np.linalg.norm(b["grads"][n]), 1e-30), n) for n in a["grads"]), reverse=True)
    for e, n in rows[:12]:
        print("%.3e %s" % (e, n))
