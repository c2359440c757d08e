#!/bin/bash
# round 5: the whole GPU suite, then every measurement artefact of the round (tools/collect_profiles.sh)
cd /root/repo
out=/root/repo/gpurun_out/r05_full
mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q --durations=10 > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $out/pytest.log
tools/collect_profiles.sh r05_collect > $out/collect.log 2>&1
tail -5 $out/collect.log
python - <<PY
import json
d=json.loads(open("/root/repo/gpurun_out/r05_collect/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","strict_fp16x3","attention_guard","peaked_workload","kernel_groups_ms_per_step"):
    print(k, json.dumps(d.get(k))[:500])
PY
