#!/usr/bin/env python3
"""C5 (lin_elastic_2d, SA with near-null-space, PCG) under rocprofv3 --kernel-trace --stats: which kernels a 208-row
V-cycle spends its time in.  Run with AMGH_USE_GRAPH=0 (rocprofv3 aborts on graph replays):

    AMGH_USE_GRAPH=0 rocprofv3 --kernel-trace --stats -d gpurun_out/c5prof -- python tools/c5_profile.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG  # noqa: E402

d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lin_elastic_2d.npz"))
A = AMG.SparseMatrixCSC.from_arrays(int(d["m"]), int(d["n"]), d["colptr"], d["rowval"], d["nzval"])
ml = AMG.smoothed_aggregation(A, B=d["B"])
print("levels", [l.A.m for l in ml.levels], ml.final_A.m, "nnz", [l.A.nnz for l in ml.levels])
dev = ml.device()
lib = dev.lib
print("dependency levels", [dev.gs_dependency_levels(l) for l in range(len(ml.levels))])
n = A.m
bd = AMG.DeviceBuffer(n, 0, d["b"])
zd = AMG.DeviceBuffer(n, 0)
for _ in range(50):
    lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
lib.amgh_dev_sync(0)
t0 = time.perf_counter()
for _ in range(200):
    lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
lib.amgh_dev_sync(0)
print(f"V-cycle {1e3 * (time.perf_counter() - t0) / 200:.3f} ms")
t0 = time.perf_counter()
x, log = AMG.cg(A, d["b"], Pl=AMG.aspreconditioner(ml), reltol=1e-10, log=True)
print(f"PCG {log['iters']} iterations in {1e3 * (time.perf_counter() - t0):.2f} ms")
