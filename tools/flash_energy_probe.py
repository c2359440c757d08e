"""GPU box: is the cross-attention kernel bound by the power budget?  The SAME instruction stream (flash_split_pipe_kernel through
parq_k_attention_split, BASELINE cfg-3 shape: 4 heads, 256 queries, 192 000 keys) on operands of different bit activity:
  random   K, V ~ N(0,1): hi and lo parts carry random mantissas (the benchmark's case)
  exact16  K, V rounded to fp16-representable values: K_lo = V_lo = 0, so one of the three MFMAs of every product multiplies zeros
  zeros    K = V = 0
Run under rocprofv3 --kernel-trace --stats and read the kernel's average duration per mode (tools/r04_energy.sh); the clock and
socket power are sampled from sysfs while the kernel loops.  usage: python tools/flash_energy_probe.py <mode>"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parq_amd import _lib  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "random"
torch.set_grad_enabled(False)
lib = _lib.load()
B, H, Lq, Lk, C = 1, 4, 256, 192000, 256
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(B, Lq, C, device="cuda", generator=g)
k = torch.randn(B, Lk, C, device="cuda", generator=g)
v = torch.randn(B, Lk, C, device="cuda", generator=g)
if mode == "exact16":
    k, v = k.half().float(), v.half().float()
elif mode == "zeros":
    k, v = torch.zeros_like(k), torch.zeros_like(v)
nbytes = lib.parq_k_attention_split_scratch_bytes(B, H, Lq, Lk)
scratch = torch.empty(nbytes // 4 + 1, device="cuda")
out = torch.empty(B, Lq, C, device="cuda")


def step():
    _lib.check(lib.parq_k_attention_split(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(out), B, H, Lq, Lk, _lib.ptr(scratch), nbytes,
                                          _lib.stream_ptr()), "attention_split")


for _ in range(20):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    step()
e1.record()
torch.cuda.synchronize()
state = bench.device_state_under_load(step, seconds=1.5)
print("mode %-8s  convert+flash+merge %.1f us per call; under load: %s" % (mode, e0.elapsed_time(e1) / 50 * 1e3,
      {k_: (round(v_, 1) if isinstance(v_, float) else v_) for k_, v_ in (state or {}).items() if k_ in ("sclk_mhz", "socket_power_w")}), flush=True)
