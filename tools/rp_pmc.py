#!/usr/bin/env python3
"""L2 requests of restriction and prolongation on the fine level of the 256^3 hierarchy, natural order vs the level-ordered
copies the cycle multiplies with (hardware counters behind tools/rp_probe.py's sector counts).

    AMGH_LEAN=0 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d DIR -- python tools/rp_pmc.py child [N]
    python tools/rp_pmc.py summarize DIR

child: with every copy kept (AMGH_LEAN=0) launches R natural x3, R level-ordered x4, P natural x5, P level-ordered x6 on
the fine level through amgh_bench_op (which = 2 / 7 / 1 / 6) and prints their HIP-event times; the launch counts tell the
four groups apart in the counter CSV (same kernel template, dispatch order).
"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PLAN = [("R natural", 2, 3), ("R level-ordered", 7, 4), ("P natural", 1, 5), ("P level-ordered", 6, 6)]


def child(N):
    import amg_amd as AMG
    A = AMG.poisson((N, N, N))
    ml = AMG.ruge_stuben(A, setup="gpu")
    dev = ml.device()
    print("MARK begin", flush=True)
    for name, which, reps in PLAN:
        ms = dev.bench_op(0, which, reps=reps, warmup=0)
        print(f"{name:18s} {ms:.4f} ms per launch ({reps} launches)", flush=True)


def summarize(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    # the bench_op launches are the LAST 18 big launches of csr_stream_kernel<0 (SPMV: R) / <2 (ADD is not used by bench_op:
    # P natural / level-ordered are SPMV launches too) -> take the last 18 SPMV dispatches of >= 1e6 rows in dispatch order
    disp = {}
    for r in rows:
        if "csr_stream_kernel<0" not in r.get("Kernel_Name", ""):
            continue
        k = int(r["Dispatch_Id"])
        disp.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        disp[k]["grid"] = int(r.get("Grid_Size", 0) or 0)
    ids = sorted(disp)[-sum(p[2] for p in PLAN):]
    pos = 0
    for name, _, reps in PLAN:
        grp = [disp[i] for i in ids[pos:pos + reps]]
        pos += reps
        keys = sorted(k for k in grp[0] if k != "grid")
        print(f"{name:18s} " + "  ".join(f"{k}={sum(g.get(k, 0.0) for g in grp) / len(grp):.4g}" for k in keys) + f"  (avg of {len(grp)} launches)")


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
    else:
        summarize(sys.argv[2])
