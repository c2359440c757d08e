"""Per-kernel durations of one decoder iteration from a rocprofv3 kernel trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'parq' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'camera_local' in r['Kernel_Name']]
seg = rows[idx[3]:idx[4]]
names = []
for r in seg:
    n = r['Kernel_Name']
    for k in ('linear', 'flash_split', 'flash_f32', 'flash_merge', 'self_attn', 'project_sample', 'box_decode', 'posemb', 'kvproj', 'camera', 'initial', 'fill', 'copy'):
        if k in n:
            names.append((k, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp']), int(r['End_Timestamp']))); break
labels = ['pe1', 'pe2', 'sample', 'in-proj', 'self', 'out-proj', 'q-proj(LN1)', 'flash', 'merge', 'cross-out(rLN1)', 'ffn1(LN2)', 'ffn2(rLN2)', 'heads1(LN3,gn)', 'heads2(gn,gn)', 'decode']
start = [i for i, n in enumerate(names) if n[0] == 'posemb'][0] + 1
it = names[start + 15:start + 30]
prev_end = names[start + 14][3]
for l, (k, d, t0, t1) in zip(labels, it):
    print("%-18s %-14s %7.2f us  (gap before %5.2f us)" % (l, k, d, (t0 - prev_end) / 1e3))
    prev_end = t1
print("iteration span %.1f us, kernel time %.1f us; forward span %.1f us, %d kernels" % ((it[-1][3] - names[start + 14][3]) / 1e3, sum(x[1] for x in it), (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3, len(seg)))
