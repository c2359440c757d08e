"""Phase marks inside chain_linear_h3_kernel (development library built with the marks:
PARQ_DEV_EXTRA_FLAGS=-DPARQ_TL_MARKS python -c 'import __graft_entry__ as g; g.build_dev()' — they cost ~1.5 % of a forward even with the
timeline off, so ordinary development builds leave them out): one stamped forward at d = 1024, per launch of the `linear` family the
median over workgroups of: start -> operands A arrived + prologue + row maxima (1) -> row-maximum barrier + split (2) -> W arrived +
products issued (3) -> folded (4) -> end (stores acknowledged).  Times in us (10 ns ticks of s_memrealtime)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np   # noqa: E402
import torch
torch.set_grad_enabled(False)
from parq_amd import _lib   # noqa: E402
_lib.use_dev_library()
import bench         # noqa: E402
import iter_timeline_stamps as T   # noqa: E402

bench.WORKLOAD["dim"] = 1024
conf = bench.CONFIGS["shipped"]
bench.WORKLOAD.update({k: conf[k] for k in ("views", "image_hw", "feat_hw", "queries", "iters")})
dev = torch.device("cuda", 0)
cfg, W, dec = bench.build_decoder(dev)
dec.use_graph = False
inputs = bench.build_inputs(int(sys.argv[1]) if len(sys.argv) > 1 else 1, dev, seed=1000)
rec = T.collect(dec, inputs, bench.WORKLOAD["feat_hw"])
kid = (rec[:, 0] & np.uint64(0xff)).astype(np.int64)
nblk = ((rec[:, 0] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
ph = (rec[:, 1] >> np.uint64(20)).astype(np.uint64)
t0 = rec[:, 2].astype(np.int64); t1 = rec[:, 3].astype(np.int64)
order = np.lexsort((t1, t0))
i = 0
n = 0
print("launch  wgs   A+prologue  barrier+split  W+products  fold   epilogue+stores   workgroup (median us)   body")
while i < len(order):
    g = int(nblk[order[i]])
    idx = order[i:i + g]
    i += g
    if not ((kid[idx] == 1).all() and (ph[idx] != 0).all()):
        continue
    m = [((ph[idx] >> np.uint64(11 * k)) & np.uint64(2047)).astype(np.float64) * 0.01 for k in range(4)]
    tot = (t1[idx] - t0[idx]) * 0.01
    seg = [m[0], m[1] - m[0], m[2] - m[1], m[3] - m[2], tot - m[3]]
    body = (t1[idx].max() - t0[idx].min()) * 0.01
    print("%4d %6d   %8.2f  %12.2f  %10.2f  %5.2f  %12.2f   %14.2f   %10.2f" % ((n, g) + tuple(float(np.median(x)) for x in seg) + (float(np.median(tot)), body)))
    n += 1
    if n >= 27:
        break
