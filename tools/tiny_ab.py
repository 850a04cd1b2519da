#!/usr/bin/env python3
"""A/B of the small-operator sweep kernels on C1 and C5: gs_tiny = 1 (gs_wave_kernel where its record exists) vs
gs_tiny = 2 (gs_chain_tiny_kernel only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
from amg_amd._libs import hip_lib
from bench import uniform

def timed_cycles(dev, n, reps=20):
    lib = dev.lib
    bd = AMG.DeviceBuffer(n, 0, uniform(n, 0)); zd = AMG.DeviceBuffer(n, 0)
    best = 1e9
    for _ in range(5):
        for _ in range(200): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0); t0 = time.perf_counter()
        for _ in range(reps): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0)
        best = min(best, 1e3 * (time.perf_counter() - t0) / reps)
    return best

lib = hip_lib()
A1 = AMG.poisson(1000); ml1 = AMG.ruge_stuben(A1); dev1 = ml1.device()
d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lin_elastic_2d.npz"))
A5 = AMG.SparseMatrixCSC.from_arrays(int(d["m"]), int(d["n"]), d["colptr"], d["rowval"], d["nzval"])
ml5 = AMG.smoothed_aggregation(A5, B=d["B"]); dev5 = ml5.device()
print("C1 levels", [l.A.m for l in ml1.levels], "dependency levels", [dev1.gs_dependency_levels(l) for l in range(len(ml1.levels))])
print("C5 levels", [l.A.m for l in ml5.levels], "dependency levels", [dev5.gs_dependency_levels(l) for l in range(len(ml5.levels))])
for rnd in range(2):
    for tiny in (2, 1):
        lib.amgh_debug_set_tunable(b"gs_tiny", tiny)
        c1 = timed_cycles(dev1, 1000); c5 = timed_cycles(dev5, 208)
        AMG.cg(A5, d["b"], Pl=AMG.aspreconditioner(ml5), reltol=1e-10, log=True)
        t0 = time.perf_counter(); x, log = AMG.cg(A5, d["b"], Pl=AMG.aspreconditioner(ml5), reltol=1e-10, log=True); t = time.perf_counter() - t0
        print(f"gs_tiny = {tiny}: C1 V-cycle {c1:.3f} ms, C5 V-cycle {c5:.3f} ms, C5 PCG {log['iters']} iterations in {t * 1e3:.2f} ms", flush=True)
