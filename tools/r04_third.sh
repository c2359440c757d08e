#!/bin/bash
# round 4, third call: the -m gpu tier on the product library with the new attention step (VAR 27), the power-budget probe, bench lines
cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04/gputests2.txt; tail -3 gpurun_out/r04/gputests2.txt
bash tools/r04_energy.sh
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r04/bench_cfg3_b.json 2> gpurun_out/r04/bench_cfg3_b.err
python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_cfg5_b.json 2> gpurun_out/r04/bench_cfg5_b.err
for c in cfg3_b cfg5_b; do python -c "
import json
d=json.loads(open('gpurun_out/r04/bench_$c.json').read().strip().splitlines()[-1])
print('$c', round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['device_state_under_load'])"; done
