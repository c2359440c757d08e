"""In-kernel time stamps of one decoder forward at the benchmark workload (development library, -DPARQ_DEV_PROBES).

Every workgroup of the per-iteration kernels records s_memrealtime (100 MHz, chip-wide) at its first instruction and after its
last store was acknowledged.  Per launch this gives: when the first / last workgroup started, when the last one ended, and the
gap to the previous launch's last end = the dependent-dispatch boundary; i.e. it separates the kernel boundary from in-kernel
time, which rocprofv3's kernel trace cannot (its durations abut with 0.00 us gaps).

    python tools/iter_timeline_stamps.py [--iteration K] [--dim 256] > profiles/r03_iter_timeline.txt
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np   # noqa: E402
import torch
torch.set_grad_enabled(False)      # inference tools: the reference's drivers run these calls under no_grad (eval.py:46)         # noqa: E402

from parq_amd import _lib   # noqa: E402

_lib.use_dev_library()
import bench         # noqa: E402

NAMES = {1: "linear", 2: "project_sample", 3: "self_attn", 4: "flash_split", 5: "flash_merge", 6: "box_decode", 7: "posemb", 8: "kv_proj",
         9: "chain"}
LABELS = ["pe1", "pe2", "sample", "in-proj", "self", "out-proj", "q-proj(LN1)", "flash", "merge", "cross-out(rLN1)", "ffn1(LN2)",
          "ffn2(rLN2)", "heads1(LN3,gn)", "heads2(gn,gn)", "decode"]


def collect(dec, inputs, hw, cap=1 << 18):
    lib = _lib.load()
    lib.parq_dev_timeline.restype = C.c_int
    lib.parq_dev_timeline.argtypes = [C.c_void_p, C.c_uint]
    buf = torch.zeros(4 + 4 * cap, dtype=torch.int64, device="cuda")
    for _ in range(30):
        dec(*inputs, feat_hw=hw)
    torch.cuda.synchronize()
    _lib.check(lib.parq_dev_timeline(C.c_void_p(buf.data_ptr()), cap), "parq_dev_timeline")
    dec(*inputs, feat_hw=hw)                       # a cold-ish first stamped forward (stamp code paths), then the one we read
    torch.cuda.synchronize()
    buf[:4].zero_()
    dec(*inputs, feat_hw=hw)
    torch.cuda.synchronize()
    _lib.check(lib.parq_dev_timeline(None, 0), "parq_dev_timeline(off)")
    h = buf.cpu().numpy().astype(np.uint64)
    n = int(h[0])
    assert n <= cap, "timeline buffer overflow: %d records" % n
    rec = h[4:4 + 4 * n].reshape(n, 4)
    return rec


def launches(rec):
    """Split the records into launches: launches of one stream are serialised, so in start order the records of a launch are
    contiguous; a launch of G workgroups contributes exactly G records."""
    kid = (rec[:, 0] & np.uint64(0xff)).astype(np.int64)
    nblk = ((rec[:, 0] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
    xcc = ((rec[:, 0] >> np.uint64(40)) & np.uint64(0xf)).astype(np.int64)
    t0 = rec[:, 2].astype(np.int64)
    t1 = rec[:, 3].astype(np.int64)
    order = np.lexsort((t1, t0))
    out, i = [], 0
    while i < len(order):
        j = order[i]
        g = int(nblk[j])
        idx = order[i:i + g]
        # (a fused launch stamps two kernel ids — pe1_sample_kernel: linear tiles + sampling workgroups — so a launch is the next g
        # records of one grid size, whatever their ids)
        assert (nblk[idx] == g).all(), "records of two launches interleave (launch %d)" % len(out)
        ids = sorted(set(kid[idx].tolist()))
        name = "+".join(NAMES.get(int(k), str(k)) for k in ids)
        blk = rec[idx, 1].astype(np.int64)
        dur = (t1[idx] - t0[idx]) * 0.01
        sl = (blk % 16) >> 3                            # kvproj_dma_kernel at C = 256: slice 0 = K heads, 1 = V heads
        extra = {}
        if name == "flash_split":
            # workgroup run time by XCD and by head (grid = (key splits, 1, heads): linear index = split + nsplit * head)
            ns = g // 4 if g % 4 == 0 else g
            extra["by_xcd"] = [(int(x), float(np.median(dur[xcc[idx] == x])), float(dur[xcc[idx] == x].max())) for x in sorted(set(xcc[idx].tolist()))]
            extra["by_head"] = [(int(hd), float(np.median(dur[blk // ns == hd])), float(dur[blk // ns == hd].max())) for hd in sorted(set((blk // ns).tolist()))]
            sp = blk % ns
            extra["by_split_quarter"] = [(q, float(np.median(dur[(sp * 4) // ns == q])), float(dur[(sp * 4) // ns == q].max())) for q in range(4)]
            extra["start_by_xcd"] = [(int(x), float(np.median((t0[idx] - t0[idx].min())[xcc[idx] == x])) * 0.01) for x in sorted(set(xcc[idx].tolist()))]
        if name == "kv_proj" and (sl == 0).any() and (sl == 1).any():
            extra = {"k_wg": (float(np.median(dur[sl == 0])), float(dur[sl == 0].max())), "v_wg": (float(np.median(dur[sl == 1])), float(dur[sl == 1].max()))}
        out.append({**extra, "kernel": name, "wgs": g, "first_start": int(t0[idx].min()), "last_start": int(t0[idx].max()),
                    "first_end": int(t1[idx].min()), "last_end": int(t1[idx].max()),
                    "median_wg_us": float(np.median(t1[idx] - t0[idx])) * 0.01, "max_wg_us": float((t1[idx] - t0[idx]).max()) * 0.01,
                    "xcds": len(set(xcc[idx].tolist())), "min_wg_us": float((t1[idx] - t0[idx]).min()) * 0.01,
                    "p10_wg_us": float(np.percentile(t1[idx] - t0[idx], 10)) * 0.01, "p90_wg_us": float(np.percentile(t1[idx] - t0[idx], 90)) * 0.01,
                    "start_p50": float(np.median(t0[idx] - t0[idx].min())) * 0.01, "end_p10": float(np.percentile(t1[idx] - t0[idx].min(), 10)) * 0.01,
                    "end_p50": float(np.median(t1[idx] - t0[idx].min())) * 0.01})
        i += g
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iteration", type=int, default=4)
    ap.add_argument("--dim", type=int, default=256)
    args = ap.parse_args()
    bench.WORKLOAD["dim"] = args.dim
    dev = torch.device("cuda", 0)
    cfg, W, dec = bench.build_decoder(dev)
    inputs = bench.build_inputs(1, dev, seed=1000)
    hw = bench.WORKLOAD["feat_hw"]
    rec = collect(dec, inputs, hw)
    ls = launches(rec)
    print("# %s" % _lib.load().parq_version().decode())
    print("# one stamped forward: %d workgroup records, %d launches; stamps are 10 ns ticks of s_memrealtime" % (len(rec), len(ls)))
    for l in ls:
        if l["kernel"] == "kv_proj":
            print("# K/V projection: %d workgroups, body %.2f us; workgroup run time min %.2f / p10 %.2f / median %.2f / p90 %.2f / max %.2f us"
                  % (l["wgs"], (l["last_end"] - l["first_start"]) * 0.01, l["min_wg_us"], l["p10_wg_us"], l["median_wg_us"], l["p90_wg_us"], l["max_wg_us"]))
            if "k_wg" in l:
                print("#   K-slice workgroups median %.2f / max %.2f us, V-slice workgroups median %.2f / max %.2f us" % (l["k_wg"] + l["v_wg"]))
    # per-iteration table: the chain of 15 launches starting at the first 'linear' after the (k-1)-th box_decode
    dec_idx = [i for i, l in enumerate(ls) if l["kernel"] == "box_decode"]
    k = args.iteration
    lo = dec_idx[k - 1] + 1 if k > 0 else next(i for i, l in enumerate(ls) if l["kernel"].startswith("linear"))
    hi = dec_idx[k] + 1
    it = ls[lo:hi]
    prev_end = ls[lo - 1]["last_end"]
    print("# iteration %d: %d launches.  boundary = first workgroup start - previous launch's last end; ramp = last start - first start;"
          % (k, len(it)))
    print("# body = last end - first start; wg = median / max time of one workgroup; all in us")
    print("%-18s %-15s %5s %9s %6s %7s %13s %7s" % ("stage", "kernel", "wgs", "boundary", "ramp", "body", "wg med/max", "total"))
    tot_b = tot_body = 0.0
    small_b = small_body = 0.0
    for lab, l in zip(LABELS if len(it) == 15 else [str(i) for i in range(len(it))], it):
        b = (l["first_start"] - prev_end) * 0.01
        ramp = (l["last_start"] - l["first_start"]) * 0.01
        body = (l["last_end"] - l["first_start"]) * 0.01
        print("%-18s %-15s %5d %9.2f %6.2f %7.2f %6.2f/%-6.2f %7.2f" % (lab, l["kernel"], l["wgs"], b, ramp, body, l["median_wg_us"], l["max_wg_us"], b + body))
        tot_b += b
        tot_body += body
        if l["kernel"] == "flash_split":
            print("#   cross-attention workgroups: run time min %.2f / p10 %.2f / median %.2f / p90 %.2f / max %.2f us; starts: median %.2f, last %.2f us after the "
                  "first; ends: first %.2f, p10 %.2f, median %.2f, last %.2f us after the first start"
                  % (l["min_wg_us"], l["p10_wg_us"], l["median_wg_us"], l["p90_wg_us"], l["max_wg_us"], l["start_p50"], ramp,
                     (l["first_end"] - l["first_start"]) * 0.01, l["end_p10"], l["end_p50"], body))
        if l["kernel"] == "flash_split" and "by_xcd" in l:
            print("#   run time median/max by XCD: " + ", ".join("%d: %.1f/%.1f" % t for t in l["by_xcd"]))
            print("#   by head: " + ", ".join("%d: %.1f/%.1f" % t for t in l["by_head"]) + "; by quarter of the key axis: " + ", ".join("%d: %.1f/%.1f" % t for t in l["by_split_quarter"]))
        if l["kernel"] != "flash_split":
            small_b += b
            small_body += body
        prev_end = l["last_end"]
    print("# iteration span %.2f us: boundaries %.2f + bodies %.2f; the %d launches other than the cross-attention: boundaries %.2f + bodies %.2f = %.2f us"
          % (tot_b + tot_body, tot_b, tot_body, len(it) - 1, small_b, small_body, small_b + small_body))
    fwd = (ls[-1]["last_end"] - ls[0]["first_start"]) * 0.01
    print("# stamped forward, first start to last end: %.1f us (%d launches recorded; un-instrumented kernels: camera_local, initial_ref)" % (fwd, len(ls)))


if __name__ == "__main__":
    main()
