#!/bin/bash
# round 6: kernel trace (with timestamps) of the training step -> device idle time inside a step (tools/r06_train_gaps.py)
cd /root/repo; out=/root/repo/gpurun_out/r06t; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out/kt -o kt -- python3 /root/repo/bench.py --train --steps 4 --warmup 2 > $out/kt.log 2>&1)
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python tools/r06_train_gaps.py $f $out/window.txt | tee $out/gaps.txt
head -1 $f > $out/trace_head.csv
rm -rf $out/kt
