#!/bin/bash
# round 4, first GPU call: the whole -m gpu tier, then the three configuration lines
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04/gputests.txt
python bench.py --steps 20 --warmup 3 > gpurun_out/r04/bench_cfg3.json 2> gpurun_out/r04/bench_cfg3.err
python bench.py --config cfg2 --steps 20 --warmup 3 > gpurun_out/r04/bench_cfg2.json 2> gpurun_out/r04/bench_cfg2.err
python bench.py --config cfg5 --steps 10 --warmup 2 > gpurun_out/r04/bench_cfg5.json 2> gpurun_out/r04/bench_cfg5.err
tail -3 gpurun_out/r04/gputests.txt
for c in 3 2 5; do python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/r04/bench_cfg$c.json').read().strip().splitlines()[-1])
    print('cfg$c', round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print('cfg$c failed', e); print(open('gpurun_out/r04/bench_cfg$c.err').read()[-1500:])
"; done
