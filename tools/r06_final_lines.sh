#!/bin/bash
# round 6, final tree: the bench lines of every configuration (profiles/r06_bench*.json)
cd /root/repo
out=/root/repo/gpurun_out/r06z
rm -rf $out; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --scenes-per-gpu 8 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_8scenes.json 2>> $out/bench.err
python bench.py --config cfg2 --steps 20 --warmup 3 > $out/bench_cfg2.json 2>> $out/bench.err
python bench.py --config cfg5 --steps 10 --warmup 2 > $out/bench_cfg5.json 2>> $out/bench.err
python bench.py --config shipped --steps 20 --warmup 3 > $out/bench_shipped.json 2>> $out/bench.err
python bench.py --config shipped --scenes-per-gpu 4 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_shipped_4scenes.json 2>> $out/bench.err
python bench.py --train --steps 8 --warmup 2 > $out/bench_train.json 2>> $out/bench.err
python bench.py --train --steps 8 --warmup 2 --phase-times > $out/bench_train_phases.json 2>> $out/bench.err
python bench.py --gpus 2 --share-device --steps 5 --warmup 1 --no-cpu-baseline --no-b32 2>> $out/bench.err | grep "^{" > $out/bench_2ranks_shared.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r06z/bench*.json')):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f.split('/')[-1], round(d['value'],1), d['unit'], round(d['ms_per_step'],4), 'inflight', (d.get('two_scenes_in_flight') or {}).get('value'), 'host', d.get('host_enqueue_ms'), 'policy cost', (d.get('guard_policy_cost') or {}).get('cost_of_the_default'), 'roofline', (d.get('roofline') or {}).get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
tail -2 $out/bench.err
