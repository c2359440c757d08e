#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04
cat > /tmp/tr.py <<'PY'
import sys, os, torch
sys.path.insert(0, "/root/repo")
from parq_amd import _lib
_lib.use_dev_library()
exec(open("/root/repo/tools/time_raype.py").read())
PY
for p in 0 1 2 3 4 8 15 0; do echo -n "PARQ_RAYPE_PROBE=$p  "; PARQ_RAYPE_PROBE=$p python /tmp/tr.py 2>&1 | grep "tokens cfg3"; done | tee gpurun_out/r04/raype_probes.txt
