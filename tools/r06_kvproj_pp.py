"""K/V projection, two-phase form against the one-phase form (round 6): hash of the K/V cache image and the time of parq_prepare (prologue
+ K/V projection) at BASELINE cfg 3, in attention modes split8 / split / split8 with two safe heads.  Development library: the form is
chosen by PARQ_KVPROJ_PP (read once per process) — run once with PARQ_KVPROJ_PP=0 and once with =1 and compare the lines."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from parq_amd import _lib  # noqa: E402
_lib.use_dev_library()
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(dev)
dec.range_check = "off"
inputs = bench.build_inputs(1, dev, 1000)
h, w = bench.WORKLOAD["feat_hw"]
for mode, safe in (("split8", 0), ("split", 0), ("split8", 0b0110)):
    dec.attention_mode, dec.safe_heads = mode, safe
    dec.prepare(*inputs, feat_hw=(h, w))
    torch.cuda.synchronize()
    ws = next(reversed(dec._ws.values())).ws
    digest = hashlib.sha256(ws.view(torch.int32).cpu().numpy().tobytes()).hexdigest()[:16]
    for _ in range(5):
        dec.prepare(*inputs, feat_hw=(h, w))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 40
    for _ in range(n):
        dec.prepare(*inputs, feat_hw=(h, w))
    e1.record()
    torch.cuda.synchronize()
    print("PARQ_KVPROJ_PP=%s mode %-6s safe %s: workspace sha256 %s   prepare %.1f us" % (os.environ.get("PARQ_KVPROJ_PP", "-"), mode, bin(safe), digest, e0.elapsed_time(e1) / n * 1e3))
