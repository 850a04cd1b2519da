#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel name from the counter_collection CSV.
usage: pmc_sum.py <dir with *counter_collection.csv> <counter name> [...]      (PMC_FILTER=substr[,substr...]: only kernels
whose name contains one of them; PMC_TOP=n rows)"""
import csv, glob, os, sys, collections
d = sys.argv[1]
want = set(sys.argv[2:])
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        flt = [t for t in os.environ.get("PMC_FILTER", "").split(",") if t]
        if flt and not any(t in name for t in flt):
            continue
        name = name[:60]
        c = r.get("Counter_Name")
        if c in want:
            tot[name][c] += float(r.get("Counter_Value", 0))
            if c == sorted(want)[0]:
                cnt[name] += 1
for name in sorted(tot, key=lambda k: -sum(tot[k].values()))[:int(os.environ.get('PMC_TOP', '12'))]:
    print(f"{name:60s} launches {cnt[name]:6d}  " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(tot[name].items())))
print("total", "  ".join(f"{c}={sum(t[c] for t in tot.values()):.5g}" for c in sorted(want)))
