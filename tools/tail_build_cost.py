"""What the collapsed coarse tail costs to build and what it saves per cycle (amghip.h: amgh_tail_dense_build), by the block of
right-hand sides the build runs (tunable tail_dense_batch).  python tools/tail_build_cost.py [N]   (N^3 Poisson, default 128)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amg_amd as AMG  # noqa: E402

lib = AMG.hip_lib()


def cycle_ms(ml, reps):
    dev = ml.device()
    n = ml.levels[0].A.m if ml.levels else ml.final_A.m
    bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n))
    zd = AMG.DeviceBuffer(n, 0)
    for _ in range(5):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    return 1e3 * (time.perf_counter() - t0) / reps, dev


def case(name, make, reps):
    lib.amgh_debug_set_tunable(b"tail_dense_rows", 0)
    base, _ = cycle_ms(make(), reps)
    lib.amgh_debug_set_tunable(b"tail_dense_rows", 6144)
    for batch in (17, 32, 48, 64):
        lib.amgh_debug_set_tunable(b"tail_dense_batch", batch)
        ms, dev = cycle_ms(make(), reps)
        lv, rows, bms = dev.tail_dense_info(0)
        be = bms / max(base - ms, 1e-9)
        print(f"{name}: batch {batch:2d}: tail level {lv} ({rows} rows) built in {bms:8.2f} ms; V-cycle {base:.4f} -> {ms:.4f} ms; pays after {be:7.0f} cycles", flush=True)
    lib.amgh_debug_set_tunable(b"tail_dense_batch", 32)


N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = np.load(os.path.join(ROOT, "tests", "golden", "lin_elastic_2d.npz"))
A5 = AMG.SparseMatrixCSC.from_arrays(int(d["m"]), int(d["n"]), d["colptr"], d["rowval"], d["nzval"])
case("C5 lin_elastic_2d", lambda: AMG.smoothed_aggregation(A5, B=d["B"]), 200)
A1 = AMG.poisson(1000)
case("C1 poisson(1000)", lambda: AMG.ruge_stuben(A1), 200)
A2 = AMG.poisson((1024, 1024))
jac = AMG.Jacobi(2.0 / 3.0)
case("C2 poisson(1024^2) SA Jacobi", lambda: AMG.smoothed_aggregation(A2, presmoother=jac, postsmoother=jac), 50)
A3 = AMG.poisson((N, N, N))
case(f"poisson({N}^3) ruge_stuben", lambda: AMG.ruge_stuben(A3), 20)
