#!/bin/bash
# GPU box: cost of each ingredient of the mode-4 cross-attention step (PARQ_FLASH_PROBE, flash_split8.hip) at BASELINE cfg 3
for p in ${PROBES:-0 2 4 6 8 16 22 32}; do
  PARQ_FLASH_PROBE=$p python bench.py --dev-lib --steps 10 --warmup 3 --no-cpu-baseline --no-b32 --attention-mode split8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('probe=$p flash %.1f us  forward %.4f ms' % (d['roofline']['avg_launch_ms']*1000, d['ms_per_step']))"
done
