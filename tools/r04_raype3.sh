#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_backward.py -m gpu -x -q -k "ray_pe or module or raype or parq_module" 2>&1 | tail -3
cat > /tmp/tr.py <<'PY'
import sys, os, torch
sys.path.insert(0, "/root/repo")
from parq_amd import _lib
_lib.use_dev_library()
exec(open("/root/repo/tools/time_raype.py").read())
PY
for p in 0 1 2 4 8 15 0; do echo -n "PARQ_RAYPE_PROBE=$p  "; PARQ_RAYPE_PROBE=$p python /tmp/tr.py 2>&1 | grep "tokens cfg3"; done | tee gpurun_out/r04/raype_probes2.txt
RAYPE_GRAD=1 PARQ_RAYPE_TWO_KERNELS=1 python /tmp/tr.py 2>&1 | grep "AddRayPE" | sed 's/^/two-kernel form of round 3: /' | tee -a gpurun_out/r04/raype_probes2.txt
export TMPDIR=/tmp
out=/root/repo/gpurun_out/r04/raype_kt; rm -rf $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 /root/repo/tools/time_raype.py > /dev/null 2>&1)
grep -E "raype|split_f32" $(find $out -name "*kernel_stats.csv" | head -1) | cut -c1-150 | tee -a gpurun_out/r04/raype_probes2.txt
find $out -name "*kernel_trace.csv" -delete
