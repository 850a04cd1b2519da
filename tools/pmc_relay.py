#!/usr/bin/env python3
"""HBM-side counters of the relayed dataflow sweep on an operator file, plain records against the dictionary layout
(tools/relay_bench, BW_RELAY_DICT = 0 / 1): one `rocprofv3 --kernel-trace --pmc <counter>` pass per counter and layout
(counters + kernel trace only), mean per dispatch of gs_bw_relay_kernel.  FETCH_SIZE doubled as in tools/pmc_flow.py
(gfx950, wide coalesced reads: MI355X_MICROARCH.md, HBM).   usage: python tools/pmc_relay.py OPERATOR.bin [target_rows max_rows]"""
import csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
binary = os.path.join(ROOT, "tools", "relay_bench")
for c in os.environ.get("PMC", "FETCH_SIZE WRITE_SIZE").split():
    out = []
    for d in (0, 1):
        tmp = tempfile.mkdtemp(prefix="amgh_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", BW_RELAY_ONLY="3", BW_RELAY_DICT=str(d))
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", tmp, "--", binary] + args,
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            vals = []
            for f in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == c and "gs_bw_relay_kernel" in row.get("Kernel_Name", ""):
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                out.append(f"{'dictionary' if d else 'plain'}: failed ({r.stdout.decode(errors='replace')[-200:]})"); continue
            per = sum(vals) / len(vals)
            gb = per * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9 if c in ("FETCH_SIZE", "WRITE_SIZE") else per
            out.append(f"{'dictionary' if d else 'plain'}: {gb:.3f} {'GB' if c in ('FETCH_SIZE', 'WRITE_SIZE') else ''} per sweep ({len(vals)} dispatches)")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    print(f"{c:>12}: " + " | ".join(out), flush=True)
