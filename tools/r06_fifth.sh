#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06e
for cfg in "0 0" "1 0" "0 0" "1 0" "1 1"; do set -- $cfg; PARQ_KVPROJ_ROT=$1 PARQ_KVPROJ_PP=$2 timeout 300 python tools/r06_kvproj_pp.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ROT=$1 /"; done | tee gpurun_out/r06e/kvproj_rot.txt
