#!/bin/bash
# round 6: fp16 x 3 chain tile, threshold of the 32-row form (PARQ_CHAIN_H3_ROWS32 = minimum workgroups of the 32-row grid) — per-launch
# times at the shipped width, 1 / 2 / 4 scenes (development library)
cd /root/repo
out=/root/repo/gpurun_out/r06q
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$name -o kt -- python3 /root/repo/bench.py --dev-lib --kernels-only --config shipped --steps 20 --warmup 3 "$@" > $out/kt_$name.log 2>&1); cp $(find $out/kt_$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv; rm -rf $out/kt_$name; }
for b in 1 2 4; do
  for th in 256 1000000; do for af in 1 2; do export PARQ_CHAIN_H3_ROWS32_ADD=$af;
    export PARQ_CHAIN_H3_ROWS32=$th
    kt b${b}_th${th}_af${af} --scenes-per-gpu $b; done
  done
done
python - <<'PY' | tee $out/table.txt
import csv, re
out='/root/repo/gpurun_out/r06q'
forms=[(256,1),(256,2),(1000000,1)]
for b in (1,2,4):
    cols={}
    for f in forms:
        d={}
        for r in csv.DictReader(open('%s/b%d_th%d_af%d_kernel_stats.csv'%((out,b)+f))):
            m=re.search(r'chain_linear_h3_kernel<(\d+), (\d), (\d), (\d), (\d), (\w+), (\w+), (\d), (\w+), (\d)>', r['Name'])
            if m:
                key=(m.group(1),m.group(4),m.group(5),m.group(7),m.group(8),m.group(9))   # K, prologue, addend, relu, residual, moments
                d[key]=(int(m.group(2)),int(m.group(3)),float(r['AverageNs'])/1e3)
        cols[f]=d
    print("== %d scene(s): launch (K, prologue, addend, relu, residual, moments) -> us [sub-tiles x row halves]: 32-row grid >= 256 (addend launch too) | >= 256 (addend launch >= 512) | 16-row tiles only"%b)
    for key in sorted(cols[forms[0]]):
        print("  %-46s"%str(key), ' | '.join("%6.1f [%dx%d]"%(cols[f][key][2],cols[f][key][0],cols[f][key][1]) if key in cols[f] else '   -   ' for f in forms))
    print("  sum", ' | '.join("%6.1f"%sum(v[2] for v in cols[f].values()) for f in forms))
PY
