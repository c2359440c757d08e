#!/bin/bash
# round 6, first GPU call: the tests the policy / graph change touches, then the bench line
cd /root/repo
mkdir -p gpurun_out/r06a
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_streams.py tests/test_gpu_tiers.py tests/test_gpu_range.py tests/test_gpu_properties.py tests/test_gpu_reference_pins.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r06a/tests.log
cat gpurun_out/r06a/tests.log
timeout 900 python bench.py --no-b32 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r06a/bench.json'))
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],4))
print('policy', json.dumps(d.get('guard_policy_cost'), indent=1))
print('inflight', d.get('two_scenes_in_flight',{}).get('value'), d.get('two_scenes_in_flight',{}).get('outputs_bit_identical_to_one_at_a_time'))
print('strict', d.get('strict_fp16x3',{}).get('value'))
print('groups', d['kernel_groups_ms_per_step'])
print('traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_source'))
PY
tail -5 gpurun_out/r06a/bench.err
