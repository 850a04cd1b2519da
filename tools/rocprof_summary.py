#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel stats plus the fine-level SpMV launches
(the roofline kernel) separated from the smaller launches of the same template."""
import csv, sys, collections
path, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
with open(out, "w") as f:
    f.write("command: rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --light   (MI355X, 256^3)\n")
    f.write("%-118s %7s %12s %10s %9s %10s %6s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write("%-118s %7d %12.1f %10.2f %9.2f %10.2f %6.1f\n" % (k[:118], len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    sp = [d for k, v in agg.items() if "csr_stream_kernel<0" in k for d in v if d > 250.0]
    if sp:
        f.write("\nfine-level SpMV launches (csr_stream_kernel<0,...> = M_SPMV, duration > 250 us: the 16.7M-row operator): "
                "n=%d avg=%.2f us min=%.2f max=%.2f  -> %.0f GB/s of algorithmic bytes (1740111876 B)\n"
                % (len(sp), sum(sp) / len(sp), min(sp), max(sp), 1740111876 / (sum(sp) / len(sp) * 1e-6) / 1e9))
print(open(out).read())
