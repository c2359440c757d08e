#!/bin/bash
# GPU-box helper: rocprofv3 average duration of kernels matching $1 in the training-step bench, for each value of env var $2.
cd /root/repo
pat=$1; var=$2; shift 2
for v in "$@"; do
  export $var=$v
  rm -rf gpurun_out/prof_ab
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_ab -o tr -- python3 /root/repo/bench.py --train --steps 2 --warmup 1 > /root/repo/gpurun_out/prof_ab.log 2>&1)
  echo -n "$var=$v: "
  python - "$pat" <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/prof_ab/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[1] in r['Name']:
        print("%s n=%s avg=%.1f us;" % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3), end=" ")
print()
PY
done
