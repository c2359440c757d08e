#!/bin/bash
# A/B of the training step (development library): dQ partial slots per group of key blocks, K/V projection weight gradient on a side stream
cd /root/repo
mkdir -p gpurun_out/r06g
for rep in 1 2; do
  for v in "1 0" "4 0" "1 1" "4 1" "2 1" "8 1"; do
    set -- $v
    echo -n "PARQ_BWD_KB_GROUP=$1 PARQ_BWD_SIDE_KV=$2: "
    env PARQ_BWD_KB_GROUP=$1 PARQ_BWD_SIDE_KV=$2 python bench.py --train --dev-lib --steps 12 --warmup 4 --phase-times 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['phase_ms'].items()})"
  done
done | tee gpurun_out/r06g/ab_train_bwd.txt
