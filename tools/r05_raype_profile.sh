#!/bin/bash
# ray-PE record of round 5: timing (product library), ingredient probes + phase stamps (development library), rocprofv3 kernel
# stats and the two PMC passes of the same tool
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05_raype; rm -rf $O; mkdir -p $O
for r in 1 2 3; do python tools/time_raype.py 2>&1 | grep cfg3; done > $O/time.txt
cp gpurun_out/r05_raype_probes_keep.txt $O/probes.txt 2>/dev/null || VARIANTS="0 1 2 4 8 15" tools/r05_raype_variants.sh > $O/probes.txt 2>&1
PARQ_RAYPE_PROBE=16 python tools/time_raype.py 2>&1 | grep stamps > $O/stamps.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 tools/time_raype.py > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 tools/time_raype.py > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 tools/time_raype.py > $O/pmc_w.log 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {}' > $O/kernel_stats_head.csv
python tools/pmc_summary.py $(find $O/pmc_f $O/pmc_w -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1 || true
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -size +2M -delete
du -sh $O
