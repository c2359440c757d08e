#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats of the training step (development library) per PARQ_ATTN_BWD_PIPE setting; prints the
# average duration of the cross-attention backward kernel and of the forward attention.
cd /root/repo
for v in "$@"; do
  out=/root/repo/gpurun_out/prof_tv_$v
  rm -rf $out
  (cd /tmp && export TMPDIR=/tmp && PARQ_ATTN_BWD_PIPE=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 /root/repo/bench.py --dev-lib --train --steps 3 --warmup 1 > /dev/null 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "PIPE=$v: $(grep attn_bwd_split2 $f | awk -F, '{print "attn_bwd_split2 avg_ns", $4, "calls", $2}')"
  find $out -name "*kernel_trace.csv" -delete
done
