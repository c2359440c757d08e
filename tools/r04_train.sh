#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_backward.py tests/test_gpu_headline.py tests/test_gpu_loss.py tests/test_gpu_range.py tests/test_gpu_reference_pins.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do python bench.py --train --steps 10 --warmup 3 --phase-times 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train', round(d['ms_per_step'],3), d['phase_ms'])"; done | tee gpurun_out/r04/train_cache_bwd.txt
