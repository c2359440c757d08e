#!/bin/bash
# Run on the GPU box (through gpurun): writes the round's measurement artefacts under gpurun_out/<tag>/.
#   tools/collect_profiles.sh r02
# bench lines (B=1 with the CPU baseline and the 32-scene project+sample figure, B=8, fp16 mode, training step), rocprofv3 kernel
# stats of the default bench command, PMC passes (HBM bytes at 1 and 32 scenes; MFMA / LDS counters), then tools/make_pmc_json.py.
tag=${1:-r05}
cd /root/repo
out=/root/repo/gpurun_out/$tag
rm -rf $out; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --scenes-per-gpu 8 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_8scenes.json 2>> $out/bench.err
python bench.py --attention-mode fp16 --steps 10 --warmup 2 --no-cpu-baseline --no-b32 > $out/bench_fp16.json 2>> $out/bench.err
python bench.py --train --steps 8 --warmup 2 > $out/bench_train.json 2>> $out/bench.err
python bench.py --dim 1024 --steps 10 --warmup 2 --no-b32 > $out/bench_d1024.json 2>> $out/bench.err
# the other BASELINE configurations as driver-reproducible lines (same JSON: workload string, roofline, bounded cpu_baseline)
python bench.py --config cfg2 --steps 20 --warmup 3 > $out/bench_cfg2.json 2>> $out/bench.err
python bench.py --config cfg5 --steps 10 --warmup 2 > $out/bench_cfg5.json 2>> $out/bench.err
python bench.py --config shipped --steps 20 --warmup 3 > $out/bench_shipped.json 2>> $out/bench.err
python bench.py --train --steps 8 --warmup 2 --phase-times > $out/bench_train_phases.json 2>> $out/bench.err
python bench.py --gpus 2 --share-device --steps 5 --warmup 1 --no-cpu-baseline --no-b32 2>> $out/bench.err | grep "^{" > $out/bench_2ranks_shared.json
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 /root/repo/bench.py --kernels-only --steps 20 --warmup 3 > $out/kt.log 2>&1)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  name=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/pmc_$name -o pmc -- python3 /root/repo/bench.py --kernels-only --steps 2 --warmup 1 > $out/pmc_$name.log 2>&1)
done
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/pmc32_$pass -o pmc -- python3 /root/repo/bench.py --kernels-only --scenes-per-gpu 32 --steps 1 --warmup 1 > $out/pmc32_$pass.log 2>&1)
done
python tools/pmc_summary.py $(find $out -name "*counter_collection.csv" | sort) > $out/pmc_summary.txt 2>&1
python tools/make_pmc_json.py $out > $out/pmc.json 2> $out/pmc_json.err
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
find $out -name "*kernel_trace.csv" -size +200k -delete
# gpurun merges at most 64 MiB back: the per-dispatch counter tables have been summarised above
find $out -name "*counter_collection.csv" -size +1M -delete; find $out -name "*.db" -delete
ls -la $out | head -40
