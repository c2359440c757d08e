#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06c
timeout 600 python tools/r06_wait_latency.py > gpurun_out/r06c/wait_latency.txt 2>&1
cat gpurun_out/r06c/wait_latency.txt
