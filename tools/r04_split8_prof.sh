#!/bin/bash
# GPU box: per-kernel times of the default forward (attention mode split8) from rocprofv3 --kernel-trace --stats
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8 -o s8 -- python3 /root/repo/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-b32 > /tmp/prof8.log 2>&1
f=$(find /tmp/prof8 -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-90s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cp "$f" $R/gpurun_out/r04/split8_kernel_stats.csv
