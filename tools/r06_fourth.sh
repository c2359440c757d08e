#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06d
for pp in 0 1 0 1; do PARQ_KVPROJ_PP=$pp timeout 300 python tools/r06_kvproj_pp.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06d/kvproj_pp.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split8.py tests/test_gpu_headline.py tests/test_gpu_reference_pins.py tests/test_gpu_tiers.py tests/test_gpu_decoder.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r06d/tests.log
python bench.py --no-b32 --no-cpu-baseline --no-peaked > gpurun_out/r06d/bench.json 2> gpurun_out/r06d/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r06d/bench.json'))
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],4))
g=d['guard_policy_cost']; print('sync', g['sync']['value'], 'lazy', g['lazy']['value'], 'cost', g['cost_of_the_default'], 'host', g['host_enqueue_ms'])
print('inflight', d.get('two_scenes_in_flight',{}).get('value'))
print('groups', d['kernel_groups_ms_per_step'])
print('kv', d['roofline_kv_proj']['avg_launch_ms'], d['roofline_kv_proj']['frac'], d['roofline_kv_proj']['traffic'])
PY
