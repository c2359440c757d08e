#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04
python tools/flash_variants.py 27,155,27,155 2 2>&1 | tail -6 | tee gpurun_out/r04/flash_variants4.txt
for v in 25 153 25 153; do PARQ_FLASH_VAR=$v python tools/time_cfg5.py fp16 dev 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 fp16 var $v', round(d['ms_per_forward'],3), round(d['kernel_groups_ms_per_forward']['cross_attn'],3))"; done | tee -a gpurun_out/r04/flash_variants4.txt
