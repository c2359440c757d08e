#!/bin/bash
# GPU-box helper (development library): bench line under several environment settings.
# usage: tools/ab_bench.sh "NAME=VALUE ..." ...   (one quoted group per variant; "" = defaults)
cd /root/repo
for v in "$@"; do
  echo -n "[$v] "
  env $v python bench.py --dev-lib --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', {k: round(v,4) for k,v in d.get('kernel_groups_ms_per_step',{}).items()})"
done
