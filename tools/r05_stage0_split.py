import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import iter_timeline_stamps as T
import bench
dev = torch.device("cuda:0")
cfg, W, dec = bench.build_decoder(dev)
inputs = bench.build_inputs(1, dev, seed=1000)
h, w = bench.WORKLOAD["feat_hw"]
rec = T.collect(dec, inputs, (h, w))
kid = (rec[:, 0] & np.uint64(0xff)).astype(np.int64)
nblk = ((rec[:, 0] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
t0 = rec[:, 2].astype(np.int64); t1 = rec[:, 3].astype(np.int64)
order = np.lexsort((t1, t0)); i = 0; n = 0
while i < len(order):
    g = int(nblk[order[i]]); idx = order[i:i + g]; i += g
    ids = set(kid[idx].tolist())
    if ids == {1, 2}:
        n += 1
        first = t0[idx].min()
        for k in (1, 2):
            m = kid[idx] == k
            d = (t1[idx][m] - t0[idx][m]) * 0.01
            st = (t0[idx][m] - first) * 0.01; en = (t1[idx][m] - first) * 0.01
            print("launch %d id %d (%s): n %d  run min/med/p90/max %.2f %.2f %.2f %.2f  start med/max %.2f %.2f  end med/max %.2f %.2f" % (
                n, k, T.NAMES[k], m.sum(), d.min(), np.median(d), np.percentile(d, 90), d.max(), np.median(st), st.max(), np.median(en), en.max()))
