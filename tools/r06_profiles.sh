#!/bin/bash
# round 6: rocprofv3 kernel statistics of the bench commands (the program itself behind `--`, --kernels-only: nothing but the timed workload's
# forwards) + PMC passes of the headline configuration; summaries land under gpurun_out/r06p/ and are copied into profiles/ by hand
cd /root/repo
out=/root/repo/gpurun_out/r06p
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
kt() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$name -o kt -- python3 /root/repo/bench.py "$@" > $out/kt_$name.log 2>&1); cp $(find $out/kt_$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv 2>/dev/null; find $out/kt_$name -name "*kernel_trace.csv" -size +200k -delete; find $out/kt_$name -name "*.db" -delete; }
kt cfg3 --kernels-only --steps 20 --warmup 3
kt shipped --kernels-only --config shipped --steps 20 --warmup 3
kt cfg5 --kernels-only --config cfg5 --steps 6 --warmup 2
kt cfg2 --kernels-only --config cfg2 --steps 20 --warmup 3
kt train --train --steps 4 --warmup 2
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  name=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/pmc_$name -o pmc -- python3 /root/repo/bench.py --kernels-only --steps 2 --warmup 1 > $out/pmc_$name.log 2>&1)
done
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/pmc32_$pass -o pmc -- python3 /root/repo/bench.py --kernels-only --scenes-per-gpu 32 --steps 1 --warmup 1 > $out/pmc32_$pass.log 2>&1)
done
python tools/pmc_summary.py $(find $out -name "*counter_collection.csv" | sort) > $out/pmc_summary.txt 2>&1
python tools/make_pmc_json.py $out > $out/pmc.json 2> $out/pmc_json.err
find $out -name "*counter_collection.csv" -size +1M -delete; find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +200k -delete
ls -la $out | head -40
head -12 $out/cfg3_kernel_stats.csv
