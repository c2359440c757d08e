#!/bin/bash
# round 4: the one-pass ray-PE kernel — parity tests, then time against the two-kernel form (development switch) on one box
cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_backward.py -m gpu -x -q -k "ray_pe or module or raype or parq_module" 2>&1 | tail -5
cat > /tmp/tr.py <<'PY'
import sys, os, torch
sys.path.insert(0, "/root/repo")
from parq_amd import _lib
_lib.use_dev_library()
exec(open("/root/repo/tools/time_raype.py").read())
PY
for rep in 1 2; do
  python /tmp/tr.py 2>&1 | grep "AddRayPE" | sed 's/^/one-pass   /'
  RAYPE_GRAD=1 python /tmp/tr.py 2>&1 | grep "AddRayPE" | sed 's/^/one-pass, hidden kept (autograd forward) /'
  RAYPE_GRAD=1 PARQ_RAYPE_TWO_KERNELS=1 python /tmp/tr.py 2>&1 | grep "AddRayPE" | sed 's/^/two-kernel form of round 3 /'
done 2>&1 | tee gpurun_out/r04/raype_ab.txt
export TMPDIR=/tmp
out=/root/repo/gpurun_out/r04/raype_kt; rm -rf $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 /root/repo/tools/time_raype.py > /dev/null 2>&1)
grep -E "raype|split_f32" $(find $out -name "*kernel_stats.csv" | head -1) | cut -c1-150 | tee -a gpurun_out/r04/raype_ab.txt
for pass in FETCH_SIZE WRITE_SIZE; do
  o=/root/repo/gpurun_out/r04/raype_pmc_$pass; rm -rf $o
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $o -o p -- python3 /root/repo/tools/time_raype.py > /dev/null 2>&1)
  python tools/pmc_summary_any.py raype $(find $o -name "*counter_collection.csv") | tee -a gpurun_out/r04/raype_ab.txt
  rm -rf $o
done
find $out -name "*kernel_trace.csv" -delete
