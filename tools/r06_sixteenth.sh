#!/bin/bash
# round 6: fp16 x 3 chain tile, threshold of the 32-row form (PARQ_CHAIN_H3_ROWS32 = minimum workgroups of the 32-row grid) — per-launch
# times at the shipped width, 1 / 2 / 4 scenes (development library)
cd /root/repo
out=/root/repo/gpurun_out/r06q
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$name -o kt -- python3 /root/repo/bench.py --dev-lib --kernels-only --config shipped --steps 20 --warmup 3 "$@" > $out/kt_$name.log 2>&1); cp $(find $out/kt_$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv; rm -rf $out/kt_$name; }
for b in 1 2 4; do
  for th in 256 512 1000000; do
    export PARQ_CHAIN_H3_ROWS32=$th
    kt b${b}_th${th} --scenes-per-gpu $b
  done
done
python - <<'PY' | tee $out/table.txt
import csv, re
out='/root/repo/gpurun_out/r06q'
for b in (1,2,4):
    cols={}
    for th in (256,512,1000000):
        d={}
        for r in csv.DictReader(open('%s/b%d_th%d_kernel_stats.csv'%(out,b,th))):
            m=re.search(r'chain_linear_h3_kernel<(\d+), (\d), (\d), (\d), (\d), (\w+), (\w+), (\d), (\w+), (\d)>', r['Name'])
            if m:
                key=(m.group(1),m.group(4),m.group(5),m.group(7),m.group(8),m.group(9))   # K, prologue, addend, relu, residual, moments
                d[key]=(int(m.group(2)),int(m.group(3)),float(r['AverageNs'])/1e3)
        cols[th]=d
    print("== %d scene(s): launch (K, prologue, addend, relu, residual, moments) -> us [sub-tiles x row halves] at thresholds 256 | 512 | never"%b)
    for key in sorted(cols[256]):
        print("  %-46s"%str(key), ' | '.join("%6.1f [%dx%d]"%(cols[th][key][2],cols[th][key][0],cols[th][key][1]) if key in cols[th] else '   -   ' for th in (256,512,1000000)))
    print("  sum", ' | '.join("%6.1f"%sum(v[2] for v in cols[th].values()) for th in (256,512,1000000)))
PY
