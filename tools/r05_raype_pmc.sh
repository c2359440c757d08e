cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r05_raype_pmc; rm -rf $O; mkdir -p $O
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  name=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /root/repo/$O/$name -o pmc -- python3 /root/repo/tools/time_raype.py > /root/repo/$O/$name.log 2>&1)
done
python tools/pmc_summary.py $(find $O -name "*counter_collection.csv" | sort) | grep -i "onepass\|==" > $O/summary.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
cat $O/summary.txt
