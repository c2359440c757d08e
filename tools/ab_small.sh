#!/bin/bash
# GPU box: A/B of the small knobs (merge d-group, decode rows per workgroup, flash younger-half priority) on the default bench.
cd /root/repo
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-b32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['kernel_groups_ms_per_step']; print('$*', 'it/s %.1f  ms %.4f  flash %.1f us  merge %.1f us  other %.1f us  linear %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3, g['cross_attn_merge']/8*1e3, g['other']/8*1e3, g['linear']/8*1e3))"; }
for rep in 1 2; do
run PARQ_MERGE_DG=16 PARQ_DECODE_ROWS=4 PARQ_FLASH_PRIO=0
run PARQ_MERGE_DG=8 PARQ_DECODE_ROWS=4 PARQ_FLASH_PRIO=0
run PARQ_MERGE_DG=16 PARQ_DECODE_ROWS=1 PARQ_FLASH_PRIO=0
run PARQ_MERGE_DG=16 PARQ_DECODE_ROWS=4 PARQ_FLASH_PRIO=1
run PARQ_MERGE_DG=8 PARQ_DECODE_ROWS=1 PARQ_FLASH_PRIO=1
done
