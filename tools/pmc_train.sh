#!/bin/bash
# GPU-box helper: PMC counters of the training-step kernels (one rocprofv3 run per counter group, kernel-trace only).
cd /root/repo
out=/root/repo/gpurun_out/pmc_train
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
i=0
for pass in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/p$i -o pmc -- python3 /root/repo/bench.py --train --steps 1 --warmup 1 > $out/p$i.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
for path in sorted(glob.glob('/root/repo/gpurun_out/pmc_train/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in agg:
        if any(s in k for s in ("attn_bwd_split", "gemm_tn", "flash_split")):
            print(k, "n=%d dur_us=%.0f" % (len(dur[k]), sum(dur[k]) / len(dur[k]) / 1e3), {c: "%.4g" % (sum(v) / len(v)) for c, v in agg[k].items()})
PY
