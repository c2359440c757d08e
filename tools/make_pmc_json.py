"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs of tools/collect_profiles.sh -> the per-launch HBM byte counts bench.py
reports as roofline.traffic (profiles/r02_pmc.json).  FETCH_SIZE is doubled for the wide (16 B / lane) streaming kernels as
MI355X_MICROARCH.md prescribes for gfx950; the gather kernel's access width is uncalibrated (factor kept, flagged)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
KERNELS = {"flash_split8_kernel": 2.0, "flash_split_pipe_kernel": 2.0, "flash_split_kernel": 2.0, "kvproj_dma_kernel": 2.0, "project_sample_kernel": 2.0, "flash_merge_kernel": 2.0, "flash_merge_fixed_kernel": 2.0}


def means(pattern, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(out, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    acc[k].append(float(r["Counter_Value"]))
                    break
    return {k: sum(v) / len(v) for k, v in acc.items()}


res = {"source": "tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only, means per launch in KB",
       "fetch_correction_note": "gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced streaming reads (MI355X_MICROARCH.md, HBM section): x2; "
                                "the project+sample gather (1 KB rows) uses the same factor, uncalibrated; at one scene its tokens sit in the Infinity Cache",
       "by_scenes": {}}
for scenes, fp, wp in (("1", "pmc_FETCH_SIZE", "pmc_WRITE_SIZE"), ("32", "pmc32_FETCH_SIZE", "pmc32_WRITE_SIZE")):
    f, w = means(fp, "FETCH_SIZE"), means(wp, "WRITE_SIZE")
    res["by_scenes"][scenes] = {k: {"fetch_kb": f[k], "write_kb": w.get(k, 0.0), "fetch_correction": KERNELS[k]} for k in f}
print(json.dumps(res, indent=1))
