#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats of forward_train + backward at the shipped size (d=1024, one scene)
cd /root/repo
rm -rf gpurun_out/prof_ts
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_ts -o ts -- python3 /root/repo/tools/time_train_step.py 1 1024 > /root/repo/gpurun_out/prof_ts.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_ts/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-80s n=%5s avg=%9.1f us  %5.1f%%" % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
tail -1 gpurun_out/prof_ts.log
