#!/bin/bash
# GPU box: cost of each ingredient of the pipelined cross-attention step (PARQ_FLASH_PROBE, flash_split.hip) at BASELINE cfg 3.
cd /root/repo
for p in 0 1 2 4 6 7 8 9 16 17 25 31; do
  PARQ_FLASH_PROBE=$p python bench.py --dev-lib --steps 10 --warmup 3 --no-cpu-baseline --no-b32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('probe=$p flash %.1f us' % (d['roofline']['avg_launch_ms']*1000))"
done
