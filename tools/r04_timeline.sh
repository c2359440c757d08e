#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04; export TMPDIR=/tmp
rm -rf gpurun_out/prof_it
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_it -o it -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-b32 > /root/repo/gpurun_out/prof_it.log 2>&1)
python tools/iter_timeline.py $(find gpurun_out/prof_it -name "*kernel_trace.csv" | head -1) | tee gpurun_out/r04/iter_timeline.txt
cp $(find gpurun_out/prof_it -name "*kernel_stats.csv" | head -1) gpurun_out/r04/kernel_stats_forward.csv
rm -rf gpurun_out/prof_it
