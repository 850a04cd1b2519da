#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel name from the counter_collection CSV.
usage: pmc_sum.py <dir with *counter_collection.csv> <counter name> [...]"""
import csv, glob, sys, collections
d = sys.argv[1]
want = set(sys.argv[2:])
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")[:60]
        c = r.get("Counter_Name")
        if c in want:
            tot[name][c] += float(r.get("Counter_Value", 0))
            if c == sorted(want)[0]:
                cnt[name] += 1
for name in sorted(tot, key=lambda k: -sum(tot[k].values()))[:12]:
    print(f"{name:60s} launches {cnt[name]:6d}  " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(tot[name].items())))
