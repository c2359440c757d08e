"""GPU box (development library): the feature / weight scale sweep of tests/test_gpu_range.py with the norm1 seam on and off
(PARQ_FUSE_SEAMS is read once per process: run once per setting).  Prints the worst teacher-forced error against float64 per mode."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from parq_amd import _lib
_lib.use_dev_library()
import torch
torch.set_grad_enabled(False)
import test_gpu_range as T
for fs in (1e-3, 1.0, 1e2):
    for wsc in (0.1, 1.0, 10.0):
        e = {m: T._decoder_errors(fs, wsc, m)[0] for m in ("split", "fp32")}
        print("PARQ_FUSE_SEAMS=%s features x%g in-proj x%g: split %.2e fp32 %.2e" % (os.environ.get("PARQ_FUSE_SEAMS", "default"), fs, wsc, e["split"], e["fp32"]), flush=True)
