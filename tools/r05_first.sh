#!/bin/bash
# round 5, first GPU call: per-head tier tests, the guard's domain on noise and smooth features, a bench line
cd /root/repo
out=/root/repo/gpurun_out/r05_first
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_tiers.py tests/test_gpu_split8.py tests/test_gpu_range.py tests/test_gpu_view_sharding.py -m gpu -x -q > $out/pytest_tiers.log 2>&1
echo "pytest rc=$?"; tail -25 $out/pytest_tiers.log
timeout 600 python tools/split8_guard_sweep.py 1 2 2.5 3 4 6 > $out/guard_sweep_noise.txt 2>&1
SMOOTH=1 timeout 600 python tools/split8_guard_sweep.py 1 2 3 4 6 8 > $out/guard_sweep_smooth.txt 2>&1
cat $out/guard_sweep_noise.txt $out/guard_sweep_smooth.txt
python bench.py > $out/bench.json 2> $out/bench.err
tail -c 1500 $out/bench.json
