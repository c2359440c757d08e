#!/bin/bash
# round 5: fused LayerNorm seams — stage / decoder / headline parity, then the bench line
cd /root/repo
out=/root/repo/gpurun_out/${1:-r05_seams}
mkdir -p $out
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_stages.py tests/test_gpu_decoder.py tests/test_gpu_headline.py tests/test_gpu_tiers.py tests/test_gpu_reference_pins.py > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $out/pytest.log
python bench.py --no-b32 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","strict_fp16x3","attention_guard","peaked_workload","kernel_groups_ms_per_step"):
    print(k, json.dumps(d.get(k))[:600])
PY
