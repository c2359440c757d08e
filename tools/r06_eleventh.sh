#!/bin/bash
# round 6: fp16 x 3 chain tile at the shipped width — kernel parity, decoder parity, A/B against the fp32 tiles, per-launch times
cd /root/repo
out=/root/repo/gpurun_out/r06k
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "linear" 2>&1 | tail -5 | tee $out/tests_kernels.txt
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_graph.py tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
for rep in 1 2; do
  for v in 0 1; do PARQ_CHAIN_H3=$v python tools/r06_h3.py 1 2>&1 | grep "H3="; done
done | tee $out/ab.txt
for v in 0 1; do PARQ_CHAIN_H3=$v python tools/r06_h3.py 4 2>&1 | grep "H3="; done | tee -a $out/ab.txt
kt() { name=$1; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$name -o kt -- python3 /root/repo/bench.py --dev-lib --kernels-only --config shipped --steps 20 --warmup 3 > $out/kt_$name.log 2>&1); cp $(find $out/kt_$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv; rm -rf $out/kt_$name; }
kt h3
export PARQ_CHAIN_H3_ROWS32=0; kt h3_rows16; unset PARQ_CHAIN_H3_ROWS32
for n in h3 h3_rows16; do echo "== $n"; python - <<PY
import csv
tot=0
for r in csv.DictReader(open('$out/${n}_kernel_stats.csv')):
    if 'chain_linear' in r['Name']:
        nm=r['Name']; i=nm.find('chain_linear'); print("%-80s %s %.1f"%(nm[i:i+78], r['Calls'], float(r['AverageNs']))); tot+=float(r['AverageNs'])
print("sum of the averages %.1f us"%(tot/1e3))
PY
done | tee $out/forms.txt
