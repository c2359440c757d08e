#!/bin/bash
# GPU-box helper (development library: the product library reads no environment): A/B one knob on the default bench.  Usage: tools/ab_env.sh VAR val1 val2 ...  (3 alternating repeats)
cd /root/repo
var=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    echo -n "$var=$v: "
    env $var=$v python bench.py --dev-lib --steps 20 --warmup 3 --no-cpu-baseline --no-b32 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_groups_ms_per_step'].items() if k in ('kv_proj','cross_attn')})"
  done
done
