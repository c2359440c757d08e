#!/bin/bash
# A/B of the one-pass ray-PE kernel's development variants (PARQ_RAYPE_PROBE bits 32 / 64 / 128: see raype.hip), alternating repeats
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in ${VARIANTS:-0 32 64 128 224}; do
    echo -n "PARQ_RAYPE_PROBE=$v  "; PARQ_RAYPE_PROBE=$v python tools/time_raype.py 2>&1 | grep "tokens cfg3"
  done
done
