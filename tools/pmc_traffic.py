#!/usr/bin/env python3
"""HBM traffic of ONE fine-level SpMV launch from the PMC counters, measured live (bench.py calls this in a
subprocess-per-pass; also usable by hand:  python tools/pmc_traffic.py [N=256]).

Method = MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC has 4
counter slots, FETCH_SIZE takes 3, WRITE_SIZE 2), so two `rocprofv3 --pmc` passes (counters only + kernel trace, no
other trace domain) over `tools/spmv_bench N 5 1 ship`, which launches the library's shipped csr_stream_kernel on
poisson((N,N,N)) and, in the same process, a 2 GiB 16-B/lane read and a 1 GiB copy of KNOWN size.  On gfx950
FETCH_SIZE tallies a wide coalesced read at half its bytes: the correction factor is calibrated on the known read of
the same run (expected 2.0) instead of being assumed; WRITE_SIZE is checked against the known copy the same way.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "tools", "spmv_bench")


def _pass(counter, N, timeout):
    tmp = tempfile.mkdtemp(prefix="amgh_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "--",
               BENCH, str(N), "5", "1", "ship"]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError(f"rocprofv3 rc={r.returncode}: {r.stdout.decode(errors='replace')[-300:]}")
        vals = {}
        for f in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "")
                key = ("spmv" if "csr_stream_kernel" in name else "read16" if "read16_kernel" in name
                       else "copy16" if "copy16_kernel" in name else None)
                if key:
                    vals.setdefault(key, []).append(float(row.get("Counter_Value", 0)))
        if "spmv" not in vals:
            raise RuntimeError("no csr_stream_kernel rows in the counter CSV")
        return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure(N=256, timeout=240):
    """-> dict with hbm_traffic_bytes_per_launch (corrected), the raw KiB averages and the calibration."""
    if not os.path.exists(BENCH):
        raise RuntimeError("tools/spmv_bench not built")
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    fetch, nf = _pass("FETCH_SIZE", N, timeout)
    write, _ = _pass("WRITE_SIZE", N, timeout)
    KiB = 1024.0
    # known sizes of the calibration kernels in spmv_bench: read16 = 2 GiB read, copy16 = 1 GiB read + 1 GiB written
    corr_f = (2 * 1024 ** 3) / (fetch["read16"] * KiB) if fetch.get("read16") else 2.0
    corr_w = (1024 ** 3) / (write["copy16"] * KiB) if write.get("copy16") else 1.0
    traffic = fetch["spmv"] * KiB * corr_f + write["spmv"] * KiB * corr_w
    return {"hbm_traffic_bytes_per_launch": traffic, "FETCH_SIZE_KiB_avg": fetch["spmv"],
            "WRITE_SIZE_KiB_avg": write["spmv"], "launches": nf.get("spmv", 0),
            "fetch_correction": corr_f, "write_correction": corr_w,
            "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- tools/spmv_bench N 5 1 ship; "
                      "corrections calibrated on a known 2 GiB read / 1 GiB copy in the same run"}


if __name__ == "__main__":
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 256), indent=1))
