#!/bin/bash
# round 4, fourth call: whole -m gpu tier (4-wave small-Lq attention, ray-PE weight cache), cfg2 / shipped lines, ray-PE time
cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04/gputests3.txt; tail -3 gpurun_out/r04/gputests3.txt
python bench.py --config cfg2 --steps 20 --warmup 3 > gpurun_out/r04/bench_cfg2_b.json 2> gpurun_out/r04/bench_cfg2_b.err
python bench.py --config shipped --steps 20 --warmup 3 > gpurun_out/r04/bench_shipped.json 2> gpurun_out/r04/bench_shipped.err
for c in cfg2_b shipped; do python -c "
import json
try:
    d=json.loads(open('gpurun_out/r04/bench_$c.json').read().strip().splitlines()[-1])
    print('$c', round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['kernel_groups_ms_per_step'], d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print('$c failed', e); print(open('gpurun_out/r04/bench_$c.err').read()[-1500:])
"; done
python tools/time_raype.py 2>&1 | grep AddRayPE | tee gpurun_out/r04/raype_final.txt
