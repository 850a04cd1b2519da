#!/usr/bin/env python3
"""V-cycle / solve timings of the smaller BASELINE.json configurations (C1, C2, C5) on the GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
from bench import uniform

def timed_cycles(ml, n, reps=20):
    dev = ml.device(); lib = dev.lib
    bd = AMG.DeviceBuffer(n, 0, uniform(n, 0)); zd = AMG.DeviceBuffer(n, 0)
    best = 1e9
    for _ in range(5):   # tiny cycles: the clocks only ramp up under sustained load, take the best of several rounds
        for _ in range(200 if n < 100000 else 3): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0); t0 = time.perf_counter()
        for _ in range(reps): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0)
        best = min(best, 1e3 * (time.perf_counter() - t0) / reps)
    return best, dev

A = AMG.poisson(1000); ml = AMG.ruge_stuben(A)
ms, dev = timed_cycles(ml, 1000)
print(f"C1 poisson(1000) RS sym-GS: V-cycle {ms:.3f} ms ({1000 / ms * 1e3:.3e} unknowns/s), levels {len(ml)}")
A = AMG.poisson((1024, 1024)); n = A.m
for om in (2 / 3, 0.5):
    jac = AMG.Jacobi(om)
    t0 = time.time(); ml = AMG.smoothed_aggregation(A, presmoother=jac, postsmoother=jac); ts = time.time() - t0
    ms, dev = timed_cycles(ml, n)
    x, hist = AMG._solve(ml, uniform(n, 0), reltol=1e-8, maxiter=500, log=True)
    sp = dev.bench_op(0, 0, 50, 5)
    print(f"C2 poisson((1024,1024)) SA Jacobi({om:.3f}): setup {ts:.1f}s levels {[l.A.m for l in ml.levels] + [ml.final_A.m]} "
          f"V-cycle {ms:.3f} ms ({n / ms * 1e3:.3e} unknowns/s), {len(hist) - 1} cycles to 1e-8, fine SpMV {sp * 1e3:.1f} us "
          f"({(A.nnz * 12 + 4 * (n + 1) + 16 * n) / sp / 1e6:.0f} GB/s, fits the 256 MB Infinity Cache)")
d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lin_elastic_2d.npz"))
A = AMG.SparseMatrixCSC.from_arrays(int(d["m"]), int(d["n"]), d["colptr"], d["rowval"], d["nzval"])
ml = AMG.smoothed_aggregation(A, B=d["B"])
t0 = time.perf_counter(); x, log = AMG.cg(A, d["b"], Pl=AMG.aspreconditioner(ml), reltol=1e-10, log=True); t = time.perf_counter() - t0
t0 = time.perf_counter(); x, log = AMG.cg(A, d["b"], Pl=AMG.aspreconditioner(ml), reltol=1e-10, log=True); t = time.perf_counter() - t0
ms, dev = timed_cycles(ml, 208)
print(f"C5 lin_elastic_2d SA(B) PCG: {log['iters']} iterations in {t * 1e3:.2f} ms; V-cycle {ms:.3f} ms (latency-bound, n=208)")
