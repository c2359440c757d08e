#!/bin/bash
# round 4: A/B of the cross-attention step variants and of the sweep direction on one box (development library)
cd /root/repo; mkdir -p gpurun_out/r04
python tools/flash_variants.py 0,1,4,8,9,16,25,27,29,31 3 > gpurun_out/r04/flash_variants.txt 2>&1
tail -14 gpurun_out/r04/flash_variants.txt
python tools/flash_variants.py 0,25 2 PARQ_FLASH_ALTERNATE=0 > gpurun_out/r04/flash_alt0.txt 2>&1; tail -3 gpurun_out/r04/flash_alt0.txt
python tools/flash_variants.py 0,25 2 PARQ_FLASH_ALT_PHASE=1 > gpurun_out/r04/flash_altphase1.txt 2>&1; tail -3 gpurun_out/r04/flash_altphase1.txt
