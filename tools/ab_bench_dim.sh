#!/bin/bash
# GPU-box helper (development library): bench line at a non-default decoder width under several environment settings.
# usage: tools/ab_bench_dim.sh DIM "NAME=VALUE ..." ...
cd /root/repo
dim=$1; shift
for v in "$@"; do
  echo -n "[dim $dim | $v] "
  env $v python bench.py --dev-lib --dim $dim --steps 20 --warmup 3 --no-b32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', {k: round(v,4) for k,v in d.get('kernel_groups_ms_per_step',{}).items()})"
done
