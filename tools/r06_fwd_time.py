"""Forward time and an output digest at a bench configuration (default cfg3; argv[1] = cfg3 | shipped | cfg2 | cfg5, argv[2] = scenes), development
library — or one of two development builds (PARQ_AB_SO=libparq_hip_dev_a.so | _b.so under parq_amd/_C/ab/) for a same-box A/B in which the
digest says whether the two builds compute the same bits."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from parq_amd import _lib  # noqa: E402
if os.environ.get("PARQ_AB_SO"):
    _lib.DEV_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parq_amd", "_C", "ab", os.environ["PARQ_AB_SO"])
_lib.use_dev_library()
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
conf = bench.CONFIGS[name]
if "dim" in conf:
    bench.WORKLOAD["dim"] = conf["dim"]
bench.WORKLOAD.update({k: conf[k] for k in ("views", "image_hw", "feat_hw", "queries", "iters")})
cfg, W, dec = bench.build_decoder(dev)
if conf["mode"]:
    dec.attention_mode = conf["mode"]
inputs = bench.build_inputs(B, dev, 1000)
h, w = bench.WORKLOAD["feat_hw"]
KEYS = ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d", "sem_cls_prob", "coord_pos")
out = dec(*inputs, feat_hw=(h, w))
torch.cuda.synchronize()
dg = hashlib.sha256(b"".join(o[k].float().cpu().numpy().tobytes() for o in out for k in KEYS)).hexdigest()[:16]
for _ in range(20):
    dec(*inputs, feat_hw=(h, w))
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 100
    for _ in range(n):
        dec(*inputs, feat_hw=(h, w))
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / n)
print("%s %s B=%d  forward %.4f ms  outputs sha256 %s" % (os.environ.get("PARQ_AB_SO", "dev"), name, B, best, dg))
