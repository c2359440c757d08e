#!/bin/bash
# round 6: parity of the 32-row streamed tile as the default (decoder / backward tests at the shipped width), then per-launch times of the
# other K = 1024 forms (8 waves splitting K; narrower column tiles) for the launches the 32-row form does not take
cd /root/repo
out=/root/repo/gpurun_out/r06j
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_backward.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
kt() { name=$1; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$name -o kt -- python3 /root/repo/bench.py --dev-lib --kernels-only --config shipped --steps 20 --warmup 3 > $out/kt_$name.log 2>&1); cp $(find $out/kt_$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv; rm -rf $out/kt_$name; }
kt default
export PARQ_CHAIN_K1024_ROWS32=0
export PARQ_CHAIN_K1024=8; kt form8; unset PARQ_CHAIN_K1024
export PARQ_CHAIN_NT_K1024=2; kt nt2
export PARQ_CHAIN_NT_K1024=3; kt nt3
export PARQ_CHAIN_NT_K1024=1; kt nt1
for n in default form8 nt2 nt3 nt1; do echo "== $n"; python - <<PY
import csv
tot=0
for r in csv.DictReader(open('$out/${n}_kernel_stats.csv')):
    if 'chain_linear' in r['Name']:
        nm=r['Name']; i=nm.find('chain_linear'); print("%-80s %s %.1f"%(nm[i:i+78], r['Calls'], float(r['AverageNs']))); tot+=float(r['AverageNs'])
print("sum of the averages %.1f us"%(tot/1e3))
PY
done | tee $out/forms.txt
