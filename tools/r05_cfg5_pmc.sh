#!/bin/bash
# round 5: HBM-side traffic of the cross-attention at BASELINE cfg 5 (two query tiles per head stream the same K/V): FETCH_SIZE per launch against the cache size
cd /root/repo
out=/root/repo/gpurun_out/${1:-r05_cfg5_pmc}
mkdir -p $out
export TMPDIR=/tmp
for ctr in FETCH_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o pmc -- python3 /root/repo/bench.py --kernels-only --config cfg5 --steps 2 --warmup 1 > $out/$ctr.log 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob("$out/FETCH_SIZE/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"][:110]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print("%-112s launches %4d  FETCH_SIZE mean %9.1f MB (x2 = %9.1f MB)" % (k, len(v), sum(v) / len(v) / 1024, 2 * sum(v) / len(v) / 1024))
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
