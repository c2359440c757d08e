#!/bin/bash
# A/B of the K/V projection's slice shares on the default bench (development library), alternating repeats
cd /root/repo
mkdir -p gpurun_out/r06g
for rep in 1 2; do
  for v in 16 17 18 19; do
    echo -n "PARQ_KVPROJ_KSHARE=$v: "
    env PARQ_KVPROJ_KSHARE=$v python bench.py --dev-lib --steps 200 --warmup 20 --no-cpu-baseline --no-b32 --no-peaked --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_groups_ms_per_step'].items() if k in ('kv_proj','cross_attn')})"
  done
done | tee gpurun_out/r06g/ab_kvproj_kshare.txt
for v in 16 17; do PARQ_KVPROJ_KSHARE=$v timeout 300 python tools/r06_kvproj_pp.py 2>&1 | grep -v amdgpu.ids | grep "split8 safe 0b0" | sed "s/^/KSHARE=$v /"; done
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split8.py tests/test_gpu_headline.py tests/test_gpu_tiers.py -m gpu -x -q 2>&1 | tail -3
