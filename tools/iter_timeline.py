"""Per-kernel durations and gaps of one steady-state decoder iteration from a rocprofv3 kernel trace CSV (timed forwards of
bench.py: no profiling events in between)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'parq' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'camera_local' in r['Kernel_Name']]
seg = rows[idx[4]:idx[5]]                              # one whole forward in the timed region
KEYS = ('linear', 'flash_split', 'flash_f32', 'flash_merge', 'self_attn', 'project_sample', 'box_decode', 'posemb', 'kvproj', 'camera', 'initial')
names = []
for r in seg:
    for k in KEYS:
        if k in r['Kernel_Name']:
            names.append((k, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
            break
labels = ['pe1', 'pe2', 'sample', 'in-proj', 'self', 'out-proj', 'q-proj(LN1)', 'flash', 'merge', 'cross-out(rLN1)', 'ffn1(LN2)', 'ffn2(rLN2)',
          'heads1(LN3,gn)', 'heads2(gn,gn)', 'decode']
start = [i for i, n in enumerate(names) if n[0] == 'posemb'][0] + 1
it = names[start + 45:start + 60]                      # the fourth iteration
prev_end = names[start + 44][2]
tot_k = tot_g = 0.0
for l, (k, t0, t1) in zip(labels, it):
    d, g = (t1 - t0) / 1e3, (t0 - prev_end) / 1e3
    print("%-18s %-14s %7.2f us  (gap before %5.2f us)" % (l, k, d, g))
    if k != 'flash_split':
        tot_k += d
    tot_g += g
    prev_end = t1
print("iteration span %.1f us; small kernels %.1f us, gaps %.1f us; forward span %.1f us, %d kernels"
      % ((it[-1][2] - names[start + 44][2]) / 1e3, tot_k, tot_g, (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3, len(seg)))
