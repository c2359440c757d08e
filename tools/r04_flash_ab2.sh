#!/bin/bash
# round 4: second A/B of the cross-attention step variants (fine-grained softmax halves), cfg 5 in fp16 mode with the tail variants,
# then the stall attribution (SQ counters) of variant 0 and of the best variant
cd /root/repo; mkdir -p gpurun_out/r04
python tools/flash_variants.py 0,27,32,59,25 3 > gpurun_out/r04/flash_variants2.txt 2>&1
tail -8 gpurun_out/r04/flash_variants2.txt
for v in 0 25 0 25; do PARQ_FLASH_VAR=$v python tools/time_cfg5.py fp16 dev 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 fp16 var $v', round(d['ms_per_forward'],3), d['kernel_groups_ms_per_forward'])"; done | tee gpurun_out/r04/cfg5_fp16_variants.txt
PARQ_FLASH_VAR=0 bash tools/pmc_stall.sh r04_stall_var0 --dev-lib --no-b32 > /dev/null 2>&1; grep -A1 "flash_split" gpurun_out/r04_stall_var0/summary.txt | head -20
PARQ_FLASH_VAR=27 bash tools/pmc_stall.sh r04_stall_var27 --dev-lib --no-b32 > /dev/null 2>&1; grep -A1 "flash_split" gpurun_out/r04_stall_var27/summary.txt | head -20
find gpurun_out/r04_stall_var0 gpurun_out/r04_stall_var27 -name "*.csv" -size +100k -delete
