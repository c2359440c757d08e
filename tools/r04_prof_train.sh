cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/proft; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft -o t -- python3 /root/repo/bench.py --train --steps 6 --warmup 2 $PARQ_BENCH_ARGS > /tmp/proft.log 2>&1
f=$(find /tmp/proft -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-100s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
