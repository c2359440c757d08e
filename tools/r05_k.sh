#!/bin/bash
# round 5: a pytest selection on the GPU box.  tools/r05_k.sh <tag> <pytest args...>
cd /root/repo
tag=$1; shift
out=/root/repo/gpurun_out/$tag
mkdir -p $out
timeout 2400 python -m pytest -m gpu -q "$@" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -30 $out/pytest.log
