#!/bin/bash
# round 6: fp16 x 3 chain tile with branch-free operand requests — parity, A/B of the XCD map (PARQ_CHAIN_H3_MAP), phase marks
cd /root/repo
out=/root/repo/gpurun_out/r06m
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3 | tee $out/tests.txt
PARQ_CHAIN_H3=0 python tools/r06_h3.py 1 2>&1 | grep "H3="
for rep in 1 2 3; do
  for v in 0 1; do echo -n "MAP=$v "; PARQ_CHAIN_H3_MAP=$v python tools/r06_h3.py 1 2>&1 | grep "H3="; done
done | tee $out/ab.txt
PARQ_CHAIN_H3=0 python tools/r06_h3.py 4 2>&1 | grep "H3="
for v in 0 1; do echo -n "MAP=$v "; PARQ_CHAIN_H3_MAP=$v python tools/r06_h3.py 4 2>&1 | grep "H3="; done | tee -a $out/ab.txt
for v in 0 1; do echo "== PARQ_CHAIN_H3_MAP=$v"; PARQ_CHAIN_H3_MAP=$v python tools/r06_h3_phases.py 2>&1 | head -11; done | tee $out/phases.txt
