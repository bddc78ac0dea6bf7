"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_<tag>_<n>): median counter value per launch of kernels matching a substring."""
import csv, glob, sys, statistics
tag, pat = sys.argv[1], sys.argv[2]
out = {}
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/*/*_counter_collection.csv")):
    per = {}
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, dv in per.items():
        out[c] = statistics.median(dv.values())
for c, v in out.items():
    print(f"{c:32s} {v:16.0f}")
