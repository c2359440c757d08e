#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats of the training-step benchmark (top kernels by total time).
cd /root/repo
rm -rf gpurun_out/prof_train
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_train -o tr -- python3 /root/repo/bench.py --train --steps 3 --warmup 1 > /root/repo/gpurun_out/prof_train.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_train/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step (4 steps incl. warm-up):", tot / 4e6)
for r in rows[:28]:
    print("%-90s n=%5s avg=%9.1f us  %5.1f%%" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
tail -2 gpurun_out/prof_train.log | cut -c1-300
