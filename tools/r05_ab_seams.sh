#!/bin/bash
cd /root/repo
out=/root/repo/gpurun_out/${1:-r05_ab_seams}
mkdir -p $out
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_stages.py tests/test_gpu_decoder.py > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 1200 tools/ab_env_long.sh PARQ_FUSE_SEAMS 0 1 2>&1 | tee $out/ab.txt
