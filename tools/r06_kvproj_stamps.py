"""Phase stamps of kvproj_dma_kernel at BASELINE cfg 3 (development library; round 6): per k-step of waves 0 and 4 of four workgroups,
cycles spent (a) waiting for the DMA / LDS counters, (b) at the barrier, (c) in the k-step's body, and the epilogue per tile."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
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
lib = _lib.load()
lib.parq_dev_timeline.restype = C.c_int
lib.parq_dev_timeline.argtypes = [C.c_void_p, C.c_uint]
for _ in range(20):
    dec.prepare(*inputs, feat_hw=(h, w))
torch.cuda.synchronize()
cap = 1 << 16
buf = torch.zeros(4 + 4 * cap, dtype=torch.int64, device="cuda")
_lib.check(lib.parq_dev_timeline(C.c_void_p(buf.data_ptr()), cap), "tl")
dec.prepare(*inputs, feat_hw=(h, w))
torch.cuda.synchronize()
buf[:4].zero_()
dec.prepare(*inputs, feat_hw=(h, w))
torch.cuda.synchronize()
_lib.check(lib.parq_dev_timeline(None, 0), "tl off")
hbuf = buf.cpu().numpy().astype(np.uint64)
n = int(hbuf[0])
rec = hbuf[4:4 + 4 * min(n, cap)].reshape(-1, 4)
rec = rec[(rec[:, 0] & np.uint64(0xff)) == 100]
print("# %d phase records" % len(rec))
for blk in range(4):
    for wave in (0, 4):
        sel = rec[(((rec[:, 0] >> np.uint64(8)) & np.uint64(0xff)) == wave) & ((rec[:, 0] >> np.uint64(16)) == blk)]
        if not len(sel):
            continue
        key = sel[:, 1].astype(np.int64)
        cyc = sel[:, 2].astype(np.int64)
        wall = sel[:, 3].astype(np.int64)
        order = np.argsort(key, kind="stable")
        key, cyc, wall = key[order], cyc[order], wall[order]
        t = {int(k): (int(c), int(wl)) for k, c, wl in zip(key, cyc, wall)}
        f2 = lambda x: "%7.0f (p10 %6.0f p90 %6.0f)" % (np.median(x), np.percentile(x, 10), np.percentile(x, 90)) if len(x) else "-"
        waits, bars, bodies, epis, ksteps = [], [], [], [], []
        e_drain, e_pieces, e_blk0, e_blk1, body3 = [], [], [], [], []
        for step in range(4, 92):
            p0, p1, p2 = t.get(step * 8), t.get(step * 8 + 1), t.get(step * 8 + 2)
            nxt = t.get((step + 1) * 8)
            if not (p0 and p1 and p2 and nxt):
                continue
            waits.append(p1[0] - p0[0]); bars.append(p2[0] - p1[0]); ksteps.append(nxt[0] - p0[0])
            if step % 4 != 3:
                bodies.append(nxt[0] - p2[0])
            else:
                # the epilogue's stamps carry the step counter AFTER the increment: step + 1
                e = [t.get((step + 1) * 8 + k) for k in (3, 5, 6, 7, 4)]
                if all(e):
                    body3.append(e[0][0] - p2[0]); e_drain.append(e[1][0] - e[0][0]); e_pieces.append(e[2][0] - e[1][0])
                    e_blk0.append(e[3][0] - e[2][0]); e_blk1.append(e[4][0] - e[3][0]); epis.append(e[4][0] - e[0][0])
        if os.environ.get("PARQ_KVPROJ_PP") == "1":
            issue, tail, epi2 = [], [], []
            for step in range(4, 92):
                p2, p5, nxt = t.get(step * 8 + 2), t.get(step * 8 + 5), t.get((step + 1) * 8)
                e3, e4 = t.get(step * 8 + 3), t.get(step * 8 + 4)
                if p2 and p5 and nxt:
                    ep = (e4[0] - e3[0]) if (e3 and e4) else 0
                    if ep:
                        epi2.append(ep)
                    issue.append(p5[0] - p2[0] - ep); tail.append(nxt[0] - p5[0])
            print("   pipelined form: reads + MFMAs issued in %s | conversion tail %s | epilogue (inside a tile's first k-step) %s" % (f2(issue), f2(tail), f2(epi2)))
        wall_per_step = (t[88 * 8][0] - t[8 * 8][0]) / 80.0 if (88 * 8 in t and 8 * 8 in t) else float("nan")
        f = lambda x: "%7.0f (p10 %6.0f p90 %6.0f)" % (np.median(x), np.percentile(x, 10), np.percentile(x, 90)) if len(x) else "-"
        print("workgroup %d wave %d: cycle-counter ticks per k-step %s | counter wait %s | barrier %s | body %s | last k-step's body %s | epilogue %s = first accumulator read %s + e4m3 pieces of block 0 %s + rest of block 0 %s + block 1 %s | %.0f ticks per k-step over 80 steps"
              % (blk, wave, f(ksteps), f(waits), f(bars), f(bodies), f(body3), f(epis), f(e_drain), f(e_pieces), f(e_blk0), f(e_blk1), wall_per_step))
