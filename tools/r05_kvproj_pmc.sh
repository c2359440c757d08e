#!/bin/bash
# round 5 (development library): HBM fetch / write bytes of the K/V projection with one kind of epilogue store removed (PARQ_KVPROJ_PROBE8:
# 32 = no e4m3 pieces, 64 = no K hi16 plane, 128 = no V plane, 224 = no stores) — where does the read-back come from?
cd /root/repo
out=/root/repo/gpurun_out/${1:-r05_kvproj_pmc}
mkdir -p $out
export TMPDIR=/tmp
for probe in ${PROBES:-0 32 64 128 224}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && PARQ_KVPROJ_PROBE8=$probe timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/p${probe}_$ctr -o pmc -- python3 /root/repo/bench.py --dev-lib --steps 2 --warmup 1 --no-cpu-baseline --no-b32 --no-peaked > $out/p${probe}_$ctr.log 2>&1)
  done
done
python - <<PY
import csv, glob, collections
for probe in [int(x) for x in "${PROBES:-0 32 64 128 224}".split()]:
    row = []
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for path in glob.glob("$out/p%d_%s/**/*counter_collection.csv" % (probe, ctr), recursive=True):
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == ctr and "kvproj_dma_kernel" in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
        row.append(sum(vals) / max(1, len(vals)) / 1024.0)
    print("PARQ_KVPROJ_PROBE8=%3d: FETCH_SIZE %7.1f MB (x2 = %6.1f)  WRITE_SIZE %7.1f MB   [%d launches]" % (probe, row[0], 2 * row[0], row[1], len(vals)))
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
if [ -n "$AB" ]; then tools/ab_env_long.sh PARQ_KVPROJ_PROBE8 $AB 2>&1 | tee $out/ab.txt; fi
