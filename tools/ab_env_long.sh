#!/bin/bash
# GPU-box helper (development library): A/B one knob on the default bench with LONG timed regions (the 20-step form has +-1.5 % noise).
# Usage: tools/ab_env_long.sh VAR val1 val2 ...  (4 alternating repeats of 300 steps)
cd /root/repo
var=$1; shift
for rep in 1 2 3 4; do
  for v in "$@"; do
    echo -n "$var=$v: "
    env $var=$v python bench.py --dev-lib --steps 300 --warmup 20 --no-cpu-baseline --no-b32 --no-peaked 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['step_ms_hipevents']['median'],4))"
  done
done
