#!/bin/bash
# GPU box: stall attribution of the cross-attention kernel (and the other forward kernels) — SQ counters in separate
# rocprofv3 --pmc passes (8 SQ slots per pass, --kernel-trace only).  tools/pmc_stall.sh <tag> [bench args]
tag=${1:-r02_stall}; shift
cd /root/repo
out=/root/repo/gpurun_out/$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $out/counters_list.txt 2>&1)
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
            "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
            "SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/p$i -o pmc -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/p$i.log 2>&1)
done
python tools/pmc_summary.py $(find $out -name "*counter_collection.csv" | sort) > $out/summary.txt 2>&1
tail -40 $out/summary.txt
