"""Mean counter values per kernel whose name contains argv[1], from rocprofv3 --pmc counter_collection CSVs (argv[2:])."""
import collections
import csv
import sys
pat = sys.argv[1]
for path in sys.argv[2:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k, {c: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in d.items()})
