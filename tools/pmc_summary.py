"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value and duration per kernel."""
import csv, sys, collections
for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("==", path)
    for k in agg:
        if not any(s in k for s in ("flash_split", "kvproj", "project_sample", "flash_f32", "merge", "linear", "self_attn", "box_decode", "raype")):
            continue
        print(k, "n=%d" % len(dur[k]), "dur_us=%.1f" % (sum(dur[k]) / len(dur[k]) / 1e3 ), {c: "%.4g" % (sum(v) / len(v)) for c, v in agg[k].items()})
