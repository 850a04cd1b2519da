"""amgh_pcg: one launch per operation against the fused recurrence (tunable pcg_fused), C5 and Poisson grids."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import amg_amd as AMG
from bench import uniform
lib = AMG.hip_lib()
d = np.load("/root/repo/tests/golden/lin_elastic_2d.npz")
A = AMG.SparseMatrixCSC.from_arrays(int(d["m"]), int(d["n"]), d["colptr"], d["rowval"], d["nzval"])
ml = AMG.smoothed_aggregation(A, B=d["B"])
def t_cg(A, b, ml, reps=20, **kw):
    p = AMG.aspreconditioner(ml)
    for _ in range(3): x, log = AMG.cg(A, b, Pl=p, log=True, **kw)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); x, log = AMG.cg(A, b, Pl=p, log=True, **kw); best = min(best, time.perf_counter() - t0)
    return best * 1e3, log["iters"]
for plan in (0, 1):
    lib.amgh_debug_set_tunable(b"pcg_fused", plan)
    ms, it = t_cg(A, d["b"], ml, reltol=1e-10)
    print(f"C5 pcg_fused = {plan}: {it} iterations, {ms:.3f} ms", flush=True)
for shape in ((24, 24, 24), (64, 64, 64), (160, 160, 160)):
    A = AMG.poisson(shape); b = uniform(A.m, 3)
    ml = AMG.ruge_stuben(A, setup="gpu", device=0) if A.m > 1e6 else AMG.ruge_stuben(A)
    for plan in (0, 1):
        lib.amgh_debug_set_tunable(b"pcg_fused", plan)
        ms, it = t_cg(A, b, ml, reps=5, reltol=1e-8)
        print(f"poisson{shape} pcg_fused = {plan}: {it} iterations, {ms:.3f} ms = {ms / it:.3f} ms per iteration", flush=True)
