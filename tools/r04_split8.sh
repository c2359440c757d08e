#!/bin/bash
# GPU box: headline bench in the fp16 x 3 mode and in mode 4 (fp8 cross terms), same box, back to back; per-kernel times from rocprofv3
mkdir -p gpurun_out/r04
for m in split split8 split split8; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-b32 --attention-mode $m 2>/dev/null | tail -1 > gpurun_out/r04/bench_$m.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r04/bench_$m.json"))
print("$m", "%.1f it/s" % d["value"], "%.4f ms/step" % d["ms_per_step"], "flash %.1f us" % (d["roofline"]["avg_launch_ms"] * 1e3), {k: round(v, 4) for k, v in d.get("profile_ms_per_forward", {}).items()} if "profile_ms_per_forward" in d else "")
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof8 -o s8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-b32 --attention-mode split8 > /dev/null 2>&1
f=$(find /tmp/prof8 -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r04/split8_kernel_stats.csv
