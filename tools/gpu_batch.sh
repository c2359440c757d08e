#!/bin/bash
# GPU box: tests (log kept), default bench line, then whatever extra command is given.  tools/gpu_batch.sh <tag> [pytest -k expr]
tag=${1:-batch}; kexpr=${2:-}
cd /root/repo
out=/root/repo/gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
if [ -n "$kexpr" ]; then
  timeout 3000 python -m pytest tests -m gpu -x -q -k "$kexpr" > $out/pytest.log 2>&1
else
  timeout 3000 python -m pytest tests -m gpu -q --durations=15 > $out/pytest.log 2>&1
fi
echo "pytest rc=$?"
tail -30 $out/pytest.log
