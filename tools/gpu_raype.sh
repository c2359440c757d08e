#!/bin/bash
# GPU-box helper: ray-PE tests, timing and per-kernel stats
cd /root/repo
python -m pytest tests/test_gpu_decoder.py -m gpu -q -k "ray_pe or parq_module" 2>&1 | tail -3
rm -rf gpurun_out/prof_rp
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_rp -o rp -- python3 /root/repo/tools/time_raype.py 2>&1 | grep AddRay)
head -4 $(find gpurun_out/prof_rp -name "*kernel_stats.csv" | head -1) | cut -c1-170
