"""Per-kernel durations and gaps of one steady-state decoder iteration from a rocprofv3 kernel trace CSV (timed forwards of
bench.py: no profiling events in between).  An iteration = the launches after one box-decode kernel up to and including the next."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'parq' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fwd = [i for i, r in enumerate(rows) if 'forward_prologue' in r['Kernel_Name'] or 'camera_local' in r['Kernel_Name']]
seg = rows[fwd[4]:fwd[5]]                              # one whole forward in the steady state
dec = [i for i, r in enumerate(seg) if 'box_decode' in r['Kernel_Name']]
it = seg[dec[3] + 1:dec[4] + 1]                        # the fifth iteration
prev_end = int(seg[dec[3]]['End_Timestamp'])


def short(name):
    m = re.search(r'(chain_linear_stream_kernel|chain_linear_kernel|pe1_sample_kernel|linear_f32_kernel|flash_split\w*|flash_merge\w*|'
                  r'self_attn_kernel|project_sample_kernel|box_decode\w*|posemb_kernel|kvproj\w*|flash_f32_kernel)(<[^>]*>)?', name)
    if not m:
        m2 = re.search(r'_GLOBAL__N_1\d+(\w+?)I', name)
        return m2.group(1) if m2 else name[:50]
    return m.group(1) + (m.group(2) or '')


tot_k = tot_g = 0.0
for r in it:
    t0, t1 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d, g = (t1 - t0) / 1e3, (t0 - prev_end) / 1e3
    nm = short(r['Kernel_Name'])
    print("%-70s %7.2f us  (gap before %5.2f us)" % (nm, d, g))
    if 'flash_split' not in nm:
        tot_k += d
    tot_g += g
    prev_end = t1
print("iteration: %d launches, span %.1f us; kernels other than the cross-attention %.1f us, gaps %.1f us; forward span %.1f us, %d kernels"
      % (len(it), (int(it[-1]['End_Timestamp']) - int(seg[dec[3]]['End_Timestamp'])) / 1e3, tot_k, tot_g,
         (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3, len(seg)))
