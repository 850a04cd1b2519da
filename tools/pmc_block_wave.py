#!/usr/bin/env python3
"""HBM traffic of one fine-level Gauss-Seidel sweep as a wavefront of blocks — one launch per depth (gs_bw_packed_kernel) and
ONE launch with the blocks chained by flags (gs_bw_chain_kernel) — from the PMC counters: two `rocprofv3 --pmc` passes
(FETCH_SIZE, WRITE_SIZE; counters + kernel trace only) over tools/block_wave_bench poisson N.  FETCH_SIZE x 2.0 on gfx950
(the half-count of wide coalesced reads, calibrated by tools/pmc_traffic.py in bench.py's own run).  What it answers: does
the polling of the chained kernel, or its agent-scope loads and write-through stores, cost HBM traffic?"""
import csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = sys.argv[1] if len(sys.argv) > 1 else "256"

def one(counter):
    tmp = tempfile.mkdtemp(prefix="amgh_pmc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "--",
               os.path.join(ROOT, "tools", "block_wave_bench"), "poisson", N]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=500)
        if r.returncode != 0:
            raise RuntimeError(r.stdout.decode(errors="replace")[-400:])
        tot, cnt = {}, {}
        for f in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != counter: continue
                name = row.get("Kernel_Name", "")
                key = "chained" if "gs_bw_chain_kernel" in name else "per depth" if "gs_bw_packed_kernel" in name else None
                if key:
                    tot[key] = tot.get(key, 0.0) + float(row["Counter_Value"]); cnt[key] = cnt.get(key, 0) + 1
        return tot, cnt
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

f, cf = one("FETCH_SIZE"); w, cw = one("WRITE_SIZE")
n = int(N) ** 3
alg = n * (80 + 8 + 8 + 8) + 0.727 * n * 12      # records (80 B per 7-point row) + b + x read + x written + external positions and values
print(f"poisson({N}^3): algorithmic bytes of one sweep ~ {alg / 1e9:.3f} GB (records 80 B/row, b, x in, x out, 0.727 external values per row)")
for key in ("per depth", "chained"):
    sweeps = 35                               # sweeps of each kind the tool executes (checks, timing, 20 alternating, stamps)
    fe, wr = f[key] * 1024 * 2.0 / sweeps, w[key] * 1024 / sweeps
    print(f"{key:>9}: {cf[key]} dispatches; per sweep fetched {fe / 1e9:.3f} GB + written {wr / 1e9:.3f} GB = {(fe + wr) / 1e9:.3f} GB = {(fe + wr) / alg:.2f} x algorithmic")
