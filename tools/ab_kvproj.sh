#!/bin/bash
# GPU box: A/B of the K/V projection implementations (PARQ_KVPROJ_RING: LDS-DMA ring depth 4 or 5) on the default bench.
cd /root/repo
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-b32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['kernel_groups_ms_per_step']; print('$*', 'it/s %.1f  ms %.4f  flash %.1f us  kvproj %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3, g['kv_proj']*1e3))"; }
for rep in 1 2; do
run PARQ_KVPROJ_RING=4
run PARQ_KVPROJ_RING=5
done
