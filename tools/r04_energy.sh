#!/bin/bash
# round 4: the power-budget experiment (tools/flash_energy_probe.py) — per-kernel durations from rocprofv3 for three operand activities
cd /root/repo; mkdir -p gpurun_out/r04; export TMPDIR=/tmp
for m in random exact16 zeros random exact16; do
  out=/root/repo/gpurun_out/r04/energy_$m; rm -rf $out
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o e -- python3 /root/repo/tools/flash_energy_probe.py $m 2>/dev/null | grep "^mode")
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  grep -E "flash_split_pipe|kvsplit_convert|flash_merge" $f | awk -F'","' '{printf "   %-60s calls %s avg_ns %s\n", substr($1,2,60), $2, $4}'
  find $out -name "*kernel_trace.csv" -delete
done 2>&1 | tee gpurun_out/r04/energy_probe.txt
