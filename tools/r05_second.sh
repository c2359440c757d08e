#!/bin/bash
# round 5: the whole GPU suite (log kept), guard sweeps, default bench line
cd /root/repo
out=/root/repo/gpurun_out/${1:-r05_second}
mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q --durations=12 > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -40 $out/pytest.log
python bench.py > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","strict_fp16x3","attention_guard","peaked_workload","kernel_groups_ms_per_step"):
    print(k, json.dumps(d.get(k))[:900])
PY
