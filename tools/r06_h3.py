"""Shipped decoder width (d = 1024): fp16 x 3 chain tile (default) against the fp32 MFMA tiles (PARQ_CHAIN_H3=0), development library.
Prints the first iteration's largest output difference to the fp32-tile run saved by the PARQ_CHAIN_H3=0 call and the forward time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from parq_amd import _lib  # noqa: E402
if os.environ.get("PARQ_AB_SO"):          # A/B of two development builds on one box: parq_amd/_C/ab/libparq_hip_dev_{a,b}.so
    _lib.DEV_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parq_amd", "_C", "ab", os.environ["PARQ_AB_SO"])
_lib.use_dev_library()
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
conf = bench.CONFIGS["shipped"]
bench.WORKLOAD["dim"] = conf["dim"]
bench.WORKLOAD.update({k: conf[k] for k in ("views", "image_hw", "feat_hw", "queries", "iters")})
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg, W, dec = bench.build_decoder(dev)
dec.range_check = "off"
inputs = bench.build_inputs(B, dev, 1000)
h, w = bench.WORKLOAD["feat_hw"]
KEYS = ("pred_logits", "center_unnormalized", "size_unnormalized", "ortho6d", "sem_cls_prob", "coord_pos")
out = dec(*inputs, feat_hw=(h, w))
torch.cuda.synchronize()
flat = {"%d.%s" % (i, k): o[k].float().cpu() for i, o in enumerate(out) for k in KEYS}
ref_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "h3_ref_B%d.pt" % B)
tag = os.environ.get("PARQ_CHAIN_H3", "1")
if tag == "0":
    torch.save(flat, ref_path)
    diff = "reference saved"
else:
    ref = torch.load(ref_path)
    it0 = max((flat[k] - ref[k]).abs().max().item() / (ref[k].abs().max().item() + 1e-30) for k in flat if k.startswith("0."))
    fin = all(torch.isfinite(v).all().item() for v in flat.values())
    diff = "iteration 0: largest |difference| / max|value| to the fp32-tile run %.2e, all iterations finite %s" % (it0, fin)
for _ in range(10):
    dec(*inputs, feat_hw=(h, w))
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 60
    for _ in range(n):
        dec(*inputs, feat_hw=(h, w))
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / n)
print("%sH3=%s B=%d  forward %.3f ms   %s" % ((os.environ.get("PARQ_AB_SO", "") + " ") if os.environ.get("PARQ_AB_SO") else "", tag, B, best, diff))
