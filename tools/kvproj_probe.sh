#!/bin/bash
# GPU box: the K/V projection kernel with ingredients removed (PARQ_KVPROJ_PROBE bit mask: 1 no MFMAs, 2 no global stores,
# 4 no conversion, 8 no token DMA, 16 no epilogue) — where its time goes.  Results are wrong by construction.
cd /root/repo
run() { v=$1; shift; env $v python bench.py "$@" --dev-lib --steps 20 --warmup 5 --no-cpu-baseline --no-b32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['kernel_groups_ms_per_step']; print('$v $*', 'kvproj %.1f us' % (g['kv_proj']*1e3))"; }
for pr in 0 1 2 16 17 20 21 29; do run PARQ_KVPROJ_PROBE=$pr --attention-mode split; done
# the mode-4 kernel (the default forward): same bits through PARQ_KVPROJ_PROBE8
for pr in 0 1 2 4 16 17 21 29; do run PARQ_KVPROJ_PROBE8=$pr; done
