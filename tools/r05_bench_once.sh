#!/bin/bash
# round 5: the default bench line (with the live PMC passes), timed
cd /root/repo
out=/root/repo/gpurun_out/${1:-r05_bench_once}
mkdir -p $out
t0=$(date +%s); python bench.py > $out/bench.json 2> $out/bench.err; echo "bench.py wall: $(( $(date +%s) - t0 )) s"
tail -2 $out/bench.err
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
r=d["roofline"]; print({k: r.get(k) for k in ("achieved","frac","traffic","traffic_source","algorithmic_bytes_per_launch")})
print("kv", {k: d["roofline_kv_projection"].get(k) for k in ("traffic","traffic_source","algorithmic_bytes_per_launch")} if "roofline_kv_projection" in d else [k for k in d if "roof" in k])
PY
