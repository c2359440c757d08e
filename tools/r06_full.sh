#!/bin/bash
# the whole GPU tier + the default bench line (round 6)
cd /root/repo
mkdir -p gpurun_out/r06h
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06h/tests.log
cat gpurun_out/r06h/tests.log
timeout 900 python bench.py > gpurun_out/r06h/bench.json 2> gpurun_out/r06h/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r06h/bench.json'))
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'cpu', d.get('cpu_baseline',{}).get('value'))
g=d['guard_policy_cost']; print('sync', g['sync']['value'], 'lazy', g['lazy']['value'], 'cost', g['cost_of_the_default'], 'host', g['host_enqueue_ms'], g['host_enqueue_ms_without_captured_forward'])
print('inflight', d.get('two_scenes_in_flight',{}).get('value'), 'strict', d.get('strict_fp16x3',{}).get('value'), 'peaked', d.get('peaked_workload',{}).get('value'))
print('groups', d['kernel_groups_ms_per_step'])
print('kv', d['roofline_kv_proj']['avg_launch_ms'], d['roofline_kv_proj']['frac'], d['roofline_kv_proj']['traffic'])
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['traffic'])
PY
tail -3 gpurun_out/r06h/bench.err
