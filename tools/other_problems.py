"""V-cycle time on problem classes other than the 3-D Poisson headline (robustness of the schedule heuristics):
2-D Poisson 4096^2 (8190 dependency levels on the fine grid), 3-D Poisson with random diagonal shifts, a smoothed-
aggregation hierarchy with its default symmetric Gauss-Seidel."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy

lib = AMG.hip_lib()

def vcycle_ms(ml, n, reps=5):
    t0 = time.perf_counter(); dev = DeviceHierarchy(ml, 0, 1)
    bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n)); zd = AMG.DeviceBuffer(n, 0)
    for _ in range(2): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0); t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(reps): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    ms = 1e3 * (time.perf_counter() - t0) / reps
    gb = dev.device_bytes() / 1e9
    del dev; gc.collect()
    return ms, t_up, gb

cases = []
A = AMG.poisson((4096, 4096)); cases.append(("poisson((4096,4096)) 5-point", A))
M = AMG.poisson((160, 160, 160)).to_scipy().tocsr()
rng = np.random.default_rng(1)
M = (M + sp.diags(rng.random(M.shape[0]) * 2.0)).tocsc()      # variable coefficients on the diagonal
cases.append(("poisson((160,160,160)) + random diagonal in [0,2)", AMG.SparseMatrixCSC.from_scipy(M)))
cases.append(("smoothed_aggregation: poisson((160,160,160))", AMG.poisson((160, 160, 160))))
for name, A in cases:
    t0 = time.perf_counter()
    ml = AMG.smoothed_aggregation(A) if name.startswith("smoothed_aggregation") else AMG.ruge_stuben(A)
    ts = time.perf_counter() - t0
    n = A.m
    for merge in (1, 16):
        lib.amgh_debug_set_tunable(b"gs_merge", merge)
        ms, t_up, gb = vcycle_ms(ml, n)
        print(f"{name}: n={n} levels={len(ml.levels)} setup {ts:.1f}s | gs_merge<={merge}: upload+schedules {t_up:.1f}s, "
              f"{gb:.1f} GB, V-cycle {ms:.2f} ms ({n / ms / 1e3:.0f} M unknowns/s)", flush=True)
    x, hist = AMG._solve(ml, A @ np.ones(n), reltol=1e-8, log=True)
    print(f"   solve: {len(hist) - 1} cycles to 1e-8, error vs ones {np.abs(x - 1).max():.2e}")
