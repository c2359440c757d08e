#!/bin/bash
# round 6: the whole GPU tier, the default bench line and the shipped-width line at the tree with the fp16 x 3 chain tile
cd /root/repo
out=/root/repo/gpurun_out/r06p2
rm -rf $out; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $out/tests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 900 python bench.py --config shipped --steps 20 --warmup 3 > $out/bench_shipped.json 2>> $out/bench.err
timeout 900 python bench.py --config shipped --scenes-per-gpu 4 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_shipped_4scenes.json 2>> $out/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r06p2/bench*.json')):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f.split('/')[-1], round(d['value'],1), d['unit'], round(d['ms_per_step'],4), 'inflight', (d.get('two_scenes_in_flight') or {}).get('value'), 'host', d.get('host_enqueue_ms'), 'policy cost', (d.get('guard_policy_cost') or {}).get('cost_of_the_default'), 'groups', d.get('kernel_groups_ms_per_step'))
PY
tail -3 $out/bench.err
