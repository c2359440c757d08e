#!/bin/bash
# round 6: the bench lines of the other configurations + the seam A/B (4 x 300 steps, development library)
cd /root/repo
out=/root/repo/gpurun_out/r06l
rm -rf $out; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --scenes-per-gpu 8 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_8scenes.json 2>> $out/bench.err
python bench.py --config cfg2 --steps 20 --warmup 3 > $out/bench_cfg2.json 2>> $out/bench.err
python bench.py --config cfg5 --steps 10 --warmup 2 > $out/bench_cfg5.json 2>> $out/bench.err
python bench.py --config shipped --steps 20 --warmup 3 > $out/bench_shipped.json 2>> $out/bench.err
python bench.py --train --steps 8 --warmup 2 > $out/bench_train.json 2>> $out/bench.err
python bench.py --train --steps 8 --warmup 2 --phase-times > $out/bench_train_phases.json 2>> $out/bench.err
python bench.py --gpus 2 --share-device --steps 5 --warmup 1 --no-cpu-baseline --no-b32 2>> $out/bench.err | grep "^{" > $out/bench_2ranks_shared.json
bash tools/ab_env_long.sh PARQ_FUSE_SEAMS 1 0 > $out/ab_seam.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r06l/bench*.json')):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f.split('/')[-1], round(d['value'],1), d['unit'], round(d['ms_per_step'],4), 'inflight', (d.get('two_scenes_in_flight') or {}).get('value'), 'host', d.get('host_enqueue_ms'), 'policy cost', (d.get('guard_policy_cost') or {}).get('cost_of_the_default'))
PY
cat $out/ab_seam.txt
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -m gpu -x -q 2>&1 | tail -3
