#!/bin/bash
# GPU box: kernel trace of the default bench -> per-kernel durations and gaps of one steady-state iteration (tools/iter_timeline.py)
cd /root/repo
rm -rf gpurun_out/tl
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/tl -o tl -- python3 /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-b32 > /root/repo/gpurun_out/tl.log 2>&1)
python tools/iter_timeline.py $(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
find gpurun_out/tl -name "*kernel_trace.csv" -delete
