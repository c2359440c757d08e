#!/bin/bash
# GPU-box helper: parity tests, per-kernel timeline of one iteration, bench line (B=1).  Usage: tools/gpu_check.sh [notest] [noprof]
cd /root/repo
if [[ "$*" != *notest* ]]; then python -m pytest tests -m gpu -x -q 2>&1 | tail -3; fi
if [[ "$*" != *noprof* ]]; then
  rm -rf gpurun_out/prof_it
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_it -o it -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/prof_it.log 2>&1)
  python tools/iter_timeline.py $(find gpurun_out/prof_it -name "*kernel_trace.csv" | head -1)
fi
python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_groups_ms_per_step'].items()})"
