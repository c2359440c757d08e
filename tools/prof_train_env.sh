#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats of the training step (development library) under several environment settings; prints
# the rows of the kernels named in $KERNELS (default: the TN GEMM of the dW products) and the step time of an unprofiled run.
# usage: tools/prof_train_env.sh "NAME=VALUE ..." ...
cd /root/repo
KERNELS=${KERNELS:-gemm_tn_kernel}
for v in "$@"; do
  out=/root/repo/gpurun_out/prof_te
  rm -rf $out; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && export $v && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 /root/repo/bench.py --dev-lib --train --steps 3 --warmup 1 $BENCH_ARGS > /dev/null 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "[$v] $(grep -E "$KERNELS" $f | awk -F'",' '{n=split($1,a,"::"); print a[n], $2, $3, $4}' | cut -c1-160 | tr '\n' ';')"
  echo -n "[$v] step: "
  (export $v; timeout 120 python bench.py --dev-lib --train --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms')")
  rm -rf $out
done
