#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06f
PARQ_KVPROJ_PP=1 timeout 300 python tools/r06_kvproj_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee gpurun_out/r06f/kvproj_stamps_sp.txt
