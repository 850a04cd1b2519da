"""27-point 3-D operator (26 on the diagonal, -1 to all 26 neighbours): merged groups vs wavefront of blocks on the fine
level, and a solve to 1e-8 as the correctness check of whichever schedule the cost model picks."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy

lib = AMG.hip_lib()
s = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = sp.diags([np.ones(s - 1), np.ones(s), np.ones(s - 1)], [-1, 0, 1])
M = (27.0 * sp.identity(s ** 3) - sp.kron(sp.kron(S, S), S)).tocsc()
A = AMG.SparseMatrixCSC.from_scipy(M); n = A.m
t0 = time.perf_counter(); ml = AMG.ruge_stuben(A); ts = time.perf_counter() - t0
print(f"27-point {s}^3 n={n} nnz={M.nnz} levels {[l.A.m for l in ml.levels]} setup {ts:.1f}s", flush=True)
lib.amgh_debug_set_tunable(b"gs_bw_min_rows", 1000000)
for bw in (0, 1, 2):
    lib.amgh_debug_set_tunable(b"gs_bw", bw)
    dev = DeviceHierarchy(ml, 0, 1)
    bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n)); zd = AMG.DeviceBuffer(n, 0)
    for _ in range(3): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    rounds = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0); rounds.append(2e2 * (time.perf_counter() - t0))
    st = dev.gs_sweep_stats(0, False)
    print(f"gs_bw={bw}: {dev.device_bytes() / 1e9:.1f} GB V-cycle {min(rounds):.2f} ms, fine sweep {dev.bench_op(0, 4, 3, 1):.3f} ms, "
          f"launches {st['launches']} tri {st['tri_entries']}", flush=True)
    del dev, bd, zd; gc.collect()
    x, hist = AMG._solve(ml, A @ np.ones(n), reltol=1e-8, log=True)
    print(f"   solve: {len(hist) - 1} cycles, error vs ones {np.abs(x - 1).max():.2e}", flush=True)
