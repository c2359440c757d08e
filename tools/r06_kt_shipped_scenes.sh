#!/bin/bash
# round 6: per-launch averages of the chain GEMMs at the shipped width by scenes per call (development library)
cd /root/repo; out=/root/repo/gpurun_out/r06r; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
for b in "$@"; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$b -o kt -- python3 /root/repo/bench.py --dev-lib --kernels-only --config shipped --steps 20 --warmup 3 --scenes-per-gpu $b > $out/kt_$b.log 2>&1)
  cp $(find $out/kt_$b -name "*kernel_stats.csv" | head -1) $out/b${b}_kernel_stats.csv; rm -rf $out/kt_$b
  echo "== $b scene(s)"; python - <<PY
import csv
tot=0
for r in csv.DictReader(open('$out/b${b}_kernel_stats.csv')):
    if 'chain_linear' in r['Name']:
        nm=r['Name']; i=nm.find('chain_linear'); print("%-70s %s %.1f"%(nm[i:i+68], r['Calls'], float(r['AverageNs'])/1e3)); tot+=float(r['AverageNs'])
print("sum %.1f us"%(tot/1e3))
PY
done
