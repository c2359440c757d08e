#!/bin/bash
# GPU-box helper (development library): stamped per-stage timeline of one decoder iteration under several PARQ_* settings.
# usage: tools/ab_chain.sh "NAME=VALUE ..." "NAME=VALUE ..." ...   (one quoted group per variant; "" = defaults)
cd /root/repo
for v in "$@"; do
  echo "=== variant: [$v]"
  env $v python tools/iter_timeline_stamps.py 2>/dev/null | grep -v "^# parq_hip\|^# one stamped\|^# body\|^# iteration 4:"
done
