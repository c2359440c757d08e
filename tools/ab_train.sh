#!/bin/bash
# GPU-box helper (development library): training-step bench line under several environment settings.
# usage: tools/ab_train.sh "NAME=VALUE ..." ...
cd /root/repo
for v in "$@"; do
  echo -n "[train | $v] "
  env $v python bench.py --dev-lib --train --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'steps/s', round(d['ms_per_step'],2), 'ms', 'loss', d['final_loss'])"
done
