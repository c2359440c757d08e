#!/bin/bash
# GPU-box helper: per-kernel timeline of one iteration (tools/gpu_check.sh) with the main library and each variant library
cd /root/repo
cp parq_amd/_C/libparq_hip.so /tmp/lib_base.so
for lib in /tmp/lib_base.so parq_amd/_C/variants/lib_*.so; do
  cp $lib parq_amd/_C/libparq_hip.so
  echo "== $(basename $lib)"
  tools/gpu_check.sh notest 2>&1 | grep -E "$1|iteration span"
done
cp /tmp/lib_base.so parq_amd/_C/libparq_hip.so
