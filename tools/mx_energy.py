"""GPU box: cost of a split-precision product block with the cross terms on the MX-scaled fp8 matrix instruction, against the shipped
three fp16 products (tools/bench_src/mx_energy.hip).  Prints time per block, clock and socket power under load, for random and
zero operands.  Build first:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bench_src/libmx_energy.so tools/bench_src/mx_energy.hip"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_src", "libmx_energy.so"))
lib.mx_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
n = 256 * 512
g = torch.Generator(device="cuda").manual_seed(3)
# fp16 operands: normal-distributed values; fp8 operands: random bytes that are valid e4m3 (0x7F / 0xFF are its NaNs)
rand16 = torch.randn(n * 32, device="cuda", generator=g).half()
rand8 = torch.randint(0, 256, (n * 64,), device="cuda", generator=g, dtype=torch.uint8)
rand8 = torch.where((rand8 & 0x7F) == 0x7F, rand8 & 0xF7, rand8)
zeros = torch.zeros(n * 64, device="cuda", dtype=torch.uint8)
out = torch.empty(n, device="cuda")
names = {0: "12 f16 (split, shipped)", 1: "4 f16 + 2 MX fp8 K=64", 2: "4 f16 (single product)", 3: "6 MX fp8 only", 4: "8 f16 (two terms)"}
iters = 4000
for data in ("random", "zeros"):
    for mode in (0, 1, 2, 3, 4):
        a = b = zeros if data == "zeros" else rand16.view(torch.uint8)
        a8 = b8 = zeros if data == "zeros" else rand8
        s = torch.cuda.current_stream().cuda_stream

        def step():
            lib.mx_launch(mode, iters, a.data_ptr(), b.data_ptr(), a8.data_ptr(), b8.data_ptr(), out.data_ptr(), s)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = bench.device_state_under_load(step, seconds=1.5)
        for _ in range(50):
            step()
        e0.record()
        for _ in range(50):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        blocks = iters * 2                               # per wave
        ns_per_block = ms * 1e6 / blocks
        print("%-7s mode %d %-26s %.3f ms  = %.1f ns per block and wave (2 waves per SIMD)  clock %s MHz  power %s W  nan=%s" % (
            data, mode, names[mode], ms, ns_per_block, st and round(st.get("sclk_mhz") or 0), st and round(st.get("socket_power_w") or 0),
            bool(torch.isnan(out).any())), flush=True)
