#!/bin/bash
# round 5 (development library): cross-attention stages handed out dynamically (PARQ_FLASH_DYN=1, default) against one range per workgroup (0)
cd /root/repo
out=/root/repo/gpurun_out/${1:-r05_ab_dyn}
mkdir -p $out
timeout 1200 python -m pytest -m gpu -q -x tests/test_gpu_split8.py tests/test_gpu_decoder.py tests/test_gpu_headline.py tests/test_gpu_tiers.py tests/test_gpu_properties.py tests/test_gpu_reference_pins.py > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout 1200 tools/ab_env_long.sh PARQ_FLASH_DYN 0 1 2>&1 | tee $out/ab.txt
timeout 300 python tools/iter_timeline_stamps.py > $out/iter_timeline_stamps.txt 2>&1
grep -n "flash_split\|cross-attention workgroups\|by XCD" $out/iter_timeline_stamps.txt | head -5
