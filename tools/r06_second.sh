#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06b
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_streams.py tests/test_gpu_tiers.py tests/test_gpu_range.py tests/test_gpu_properties.py tests/test_gpu_reference_pins.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r06b/tests.log
cat gpurun_out/r06b/tests.log
timeout 600 python tools/r06_host_path.py > gpurun_out/r06b/host_path.txt 2>&1
cat gpurun_out/r06b/host_path.txt
