#!/bin/bash
# GPU-box helper: per-kernel durations of one forward at the shipped size (d=1024)
cd /root/repo
rm -rf gpurun_out/prof_sh
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_sh -o sh -- python3 /root/repo/tools/time_shipped_cfg.py 1024 > /root/repo/gpurun_out/prof_sh.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_sh/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last forward: find the last kvproj_big launch and print the first iteration after it
idx = max(i for i, r in enumerate(rows) if 'kvproj_big_kernel' in r['Kernel_Name'])
for r in rows[idx - 1: idx + 18]:
    print("%-70s grid %-8s %8.1f us" % (r['Kernel_Name'][:70], r['Grid_Size_X'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
