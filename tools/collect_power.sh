#!/bin/bash
# GPU box: evidence for DESIGN.md §4 "power": streaming ceiling, K/V-projection ingredient probes, and socket power / shader
# clock (rocm-smi) while the K/V projection alone and the whole forward run back to back.  tools/collect_power.sh <tag>
tag=${1:-r02}
cd /root/repo
out=/root/repo/gpurun_out/power_$tag
mkdir -p $out
(cd tools/bench_src && [ -x hbm_stream ] || hipcc --offload-arch=gfx950 -O3 -w -o hbm_stream hbm_stream.hip)
timeout 120 tools/bench_src/hbm_stream > $out/hbm_stream.txt 2>&1
timeout 600 bash tools/kvproj_probe.sh > $out/kvproj_probe.txt 2>&1
{
  for p in 0 1 16 20; do PARQ_KVPROJ_PROBE=$p timeout 200 python tools/kvproj_power.py prepare 2>&1 | grep -v amdgpu.ids; done
  timeout 200 python tools/kvproj_power.py forward 2>&1 | grep -v amdgpu.ids
  PARQ_FLASH_PROBE=7 timeout 200 python tools/kvproj_power.py forward 2>&1 | grep -v amdgpu.ids
} > $out/power.txt 2>&1
echo "columns of the rocm-smi lines: device, fclk, level, mclk, level, sclk, level, socclk, level, socket power (W)" >> $out/power.txt
tail -n 60 $out/power.txt
