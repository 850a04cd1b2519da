#!/usr/bin/env python3
"""smoothed_aggregation(poisson(N^3)) with the host library vs the GPU path (same hierarchy bit for bit).
usage: python tools/sa_setup_time.py [N=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = AMG.poisson((N, N, N))
AMG.smoothed_aggregation(AMG.poisson((16, 16, 16)), setup="gpu")   # load the library / create the HIP context
for mode in ("gpu", "host", "gpu"):
    t0 = time.perf_counter()
    ml = AMG.smoothed_aggregation(A, setup=mode)
    dt = time.perf_counter() - t0
    print(f"setup={mode:4s}: {dt:6.2f} s   levels {[l.A.m for l in ml.levels] + [ml.final_A.m]}  nnz {[l.A.nnz for l in ml.levels]}", flush=True)
    if mode == "gpu":
        g = ml
    else:
        h = ml
same = all(np.array_equal(a.A.nzval, b.A.nzval) and np.array_equal(a.A.rowval, b.A.rowval) and
           np.array_equal(a.P.nzval, b.P.nzval) and np.array_equal(a.R.rowval, b.R.rowval) for a, b in zip(g.levels, h.levels))
print("identical hierarchies:", same and len(g) == len(h))
