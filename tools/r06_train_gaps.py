"""Training step: how much of a step is the device idle?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --train`, takes the LAST whole
step (delimited by attn_bwd_split2_kernel launches), and prints busy time (union of kernel intervals), idle gaps by size, and the busy time
by phase around the batched attention backward."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
big = [i for i, e in enumerate(ev) if "attn_bwd_split2_kernel" in e[2]]
assert len(big) >= 3, "need at least three steps in the trace"
a, b = big[-3], big[-2]          # one step = from the end of one batched backward to the end of the next
t0, t1 = ev[a][1], ev[b][1]
step = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print("step window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
# union of intervals
busy = 0
cur_s, cur_e = None, None
gaps = []
for s, e, n in step:
    if cur_e is None:
        cur_s, cur_e = s, e
        prev_name = n
        gaps.append((s - t0, "start"))
    elif s <= cur_e:
        if e > cur_e: prev_name = n
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, "%-60s at +%.2f ms, after %s" % (n[:60], (s - t0) / 1e6, prev_name[:50])))
        cur_s, cur_e = s, e
        prev_name = n
busy += cur_e - cur_s
print("device busy %.3f ms, idle %.3f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
gs = sorted((g for g, _ in gaps), reverse=True)
print("idle gaps: %d; > 20 us: %d (%.3f ms); 5-20 us: %d (%.3f ms); < 5 us: %d (%.3f ms)" % (
    len(gs), sum(g > 20000 for g in gs), sum(g for g in gs if g > 20000) / 1e6, sum(5000 < g <= 20000 for g in gs),
    sum(g for g in gs if 5000 < g <= 20000) / 1e6, sum(g <= 5000 for g in gs), sum(g for g in gs if g <= 5000) / 1e6))
for g, n in sorted(gaps, reverse=True)[:24]:
    print("   gap %8.1f us before %s" % (g / 1e3, n))
# phases: kernels before the step's batched backward start, the backward itself, after
bs, be = ev[b][0], ev[b][1]
pre = [e for e in step if e[1] <= bs]
names = {}
for s, e, n in step:
    k = n.split("(")[0][-50:]
    names[k] = names.get(k, 0) + (e - s)
first_fwd = min((e[0] for e in step if "flash_split" in e[2]), default=t0)
print("from the previous backward's end to this step's first cross-attention forward kernel: %.3f ms" % ((first_fwd - t0) / 1e6))
print("from there to the batched backward's start: %.3f ms; batched backward %.3f ms" % ((bs - first_fwd) / 1e6, (be - bs) / 1e6))
# the window behind the batched backward (phase 2 of the iterations, K/V-projection backward, optimizer): start offset, duration, name
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        for s, e, n in step:
            if s - t0 > 3.5e6:
                break
            i = n.find("parq::(anonymous namespace)::")
            f.write("%9.1f %8.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (n[i + 29:] if i >= 0 else n)[:90]))
