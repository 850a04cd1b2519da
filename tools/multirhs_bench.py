"""V-cycle time for blocks of right-hand sides (workspace block size bs, multilevel.jl:28-59) on one hierarchy.
usage: python tools/multirhs_bench.py [N=256] [bs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sizes = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
# AMG_RHS_IL = 0: column by column | 1: interleaved restriction / prolongation (default) | 2: + block residual
if os.environ.get("AMG_RHS_IL"):
    AMG.hip_lib().amgh_debug_set_tunable(b"rhs_il", int(os.environ["AMG_RHS_IL"]))
    print("rhs_il =", os.environ["AMG_RHS_IL"])
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A)
n = A.m
rng = np.random.default_rng(0)
base = None
for bs in sizes:
    dev = ml.device(0, bs)
    lib = dev.lib
    bd = AMG.DeviceBuffer(n * bs, 0, rng.random(n * bs))
    zd = AMG.DeviceBuffer(n * bs, 0)
    for _ in range(2):
        assert lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0) == 0
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    ms = 1e3 * (time.perf_counter() - t0) / reps
    base = base or ms
    print(f"bs={bs}: V-cycle {ms:8.2f} ms  ({ms / bs:7.2f} ms per column, {n * bs / ms / 1e3:8.1f} M unknowns/s, "
          f"{bs * base / ms:4.2f}x vs column-by-column)", flush=True)
    del bd, zd
    ml._dev.pop((0, bs), None) if hasattr(ml, "_dev") else None
