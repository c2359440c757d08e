#!/bin/bash
# GPU-box helper: A/B one environment switch on the shipped-size (d=1024) forward.  Usage: tools/ab_env_shipped.sh VAR v1 v2 ...
cd /root/repo
var=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    echo -n "$var=$v: "
    env $var=$v python tools/time_shipped_cfg.py 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_groups_ms_per_step'].items()})"
  done
done
