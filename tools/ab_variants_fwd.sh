#!/bin/bash
# GPU-box helper: default inference bench with the main library and each parq_amd/_C/variants/lib_*.so, alternating, 3 rounds
cd /root/repo
cp parq_amd/_C/libparq_hip.so /tmp/lib_base.so
for rep in 1 2 3; do
for lib in /tmp/lib_base.so parq_amd/_C/variants/lib_*.so; do
  cp $lib parq_amd/_C/libparq_hip.so
  echo -n "$(basename $lib): "
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_groups_ms_per_step'].items() if k in ('kv_proj','cross_attn','linear')})"
done
done
cp /tmp/lib_base.so parq_amd/_C/libparq_hip.so
