#!/bin/bash
# round 6: 32-row streamed chain tile at the shipped width — A/B on one box (alternating), then kernel statistics of both forms
cd /root/repo
out=/root/repo/gpurun_out/r06i
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do
  for v in 0 1 256 384 700; do
    PARQ_CHAIN_K1024_ROWS32=$v python tools/r06_rows32.py 1 2>&1 | grep ROWS32
  done
done | tee $out/ab.txt
for v in 0 1; do PARQ_CHAIN_K1024_ROWS32=$v python tools/r06_rows32.py 4 2>&1 | grep ROWS32; done | tee -a $out/ab.txt
for v in 0 1; do
  export PARQ_CHAIN_K1024_ROWS32=$v
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$v -o kt -- python3 /root/repo/bench.py --dev-lib --kernels-only --config shipped --steps 20 --warmup 3 > $out/kt_$v.log 2>&1)
  cp $(find $out/kt_$v -name "*kernel_stats.csv" | head -1) $out/rows32_${v}_kernel_stats.csv
  find $out/kt_$v -name "*kernel_trace.csv" -delete; find $out/kt_$v -name "*.db" -delete
done
for v in 0 1; do echo "== ROWS32=$v"; grep "chain_linear\|self_attn" $out/rows32_${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200; done
