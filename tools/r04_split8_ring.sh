#!/bin/bash
for r in 4 5 4 5; do
  PARQ_FLASH_RING=$r python bench.py --dev-lib --steps 20 --warmup 3 --no-cpu-baseline --no-b32 --attention-mode split8 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ring=$r flash %.1f us  forward %.4f ms' % (d['roofline']['avg_launch_ms']*1000, d['ms_per_step']))"
done
