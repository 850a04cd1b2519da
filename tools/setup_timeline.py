#!/usr/bin/env python3
"""Where the seconds before the first V-cycle go (256^3, ruge_stuben(A, setup="gpu", device=0) — what bench.py's setup_s times):
the main thread's phases (AMG_SETUP_TIMING: strength / pattern download / host C/F splitting / interpolation / R*A*P per level,
summed), the library's own build lines (AMGH_VERBOSE: block partition, dataflow layout, merged-group search per level, with
wall-clock stamps), and the total, three times.   usage: AMG_SETUP_TIMING=1 AMGH_VERBOSE=1 python tools/setup_timeline.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
AMG.ruge_stuben(AMG.poisson((24, 24, 24)), setup="gpu", device=0)    # (runtime start-up, code objects)
print("---- timed from here", flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    A = AMG.poisson((N, N, N))
    t1 = time.perf_counter()
    ml = AMG.ruge_stuben(A, setup="gpu", device=0)
    t2 = time.perf_counter()
    dev = ml.device()
    t3 = time.perf_counter()
    print(f"==== rep {rep}: poisson {t1 - t0:.2f} s, ruge_stuben(setup='gpu', device=0) {t2 - t1:.2f} s, ml.device() {t3 - t2:.3f} s -> setup_s as bench.py counts it {t3 - t0:.2f} s", flush=True)
    del dev, ml, A
