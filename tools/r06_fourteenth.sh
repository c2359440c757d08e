#!/bin/bash
# round 6: fp16 x 3 chain tile, column sub-tiles per workgroup (PARQ_CHAIN_NT_H3) — per-launch times at the shipped width, 1 and 4 scenes
cd /root/repo
out=/root/repo/gpurun_out/r06n
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$name -o kt -- python3 /root/repo/bench.py --dev-lib --kernels-only --config shipped --steps 20 --warmup 3 "$@" > $out/kt_$name.log 2>&1); cp $(find $out/kt_$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv; rm -rf $out/kt_$name; }
kt nt4
export PARQ_CHAIN_NT_H3=2; kt nt2; unset PARQ_CHAIN_NT_H3
kt nt4_b4 --scenes-per-gpu 4
export PARQ_CHAIN_NT_H3=2; kt nt2_b4 --scenes-per-gpu 4; unset PARQ_CHAIN_NT_H3
for n in nt4 nt2 nt4_b4 nt2_b4; do echo "== $n"; python - <<PY
import csv
tot=0
for r in csv.DictReader(open('$out/${n}_kernel_stats.csv')):
    if 'chain_linear' in r['Name']:
        nm=r['Name']; i=nm.find('chain_linear'); print("%-80s %s %.1f"%(nm[i:i+78], r['Calls'], float(r['AverageNs']))); tot+=float(r['AverageNs'])
print("sum of the averages %.1f us"%(tot/1e3))
PY
done | tee $out/forms.txt
for rep in 1 2; do for v in 4 2; do echo -n "NT=$v "; PARQ_CHAIN_NT_H3=$v python tools/r06_h3.py 1 2>&1 | grep "H3="; done; done | tee $out/ab.txt
