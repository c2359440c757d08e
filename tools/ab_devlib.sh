#!/bin/bash
# GPU-box helper: A/B of two BUILDS of the development library (e.g. a compile-time switch): rocprofv3 kernel stats of the training
# step with each; prints the rows of the kernels matching $KERNELS.  The variant library replaces libparq_hip_dev.so for its runs.
# usage: tools/ab_devlib.sh <variant libparq_hip_dev.so> [rounds]
cd /root/repo
KERNELS=${KERNELS:-attn_bwd_split2|flash_split_pipe}
dev=parq_amd/_C/libparq_hip_dev.so
cp $dev /tmp/dev_main.so
for r in $(seq 1 ${2:-2}); do
  for which in main variant; do
    if [ $which = main ]; then cp /tmp/dev_main.so $dev; else cp $1 $dev; fi
    out=/root/repo/gpurun_out/prof_ab
    rm -rf $out; mkdir -p $out
    (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 /root/repo/bench.py --dev-lib --train --steps 3 --warmup 1 > /dev/null 2>&1)
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    echo "[$which] $(grep -E "$KERNELS" $f | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-200 | tr '\n' ';')"
    rm -rf $out
  done
done
cp /tmp/dev_main.so $dev
