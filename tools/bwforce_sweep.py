"""V-cycle of smaller 3-D Poisson hierarchies with the block layout where the cost model wants it (gs_bw = 1) against everywhere it
can be built (gs_bw = 2), per level the smoother time.   usage: python tools/bwforce_sweep.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from bench import uniform
Ns = [int(v) for v in sys.argv[1:]] or [64, 96, 128]
lib = AMG.hip_lib()
lib.amgh_debug_set_tunable(b"gs_bw_min_rows", 1500000)
for N in Ns:
    ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu")
    n = ml.levels[0].A.m
    b = uniform(n, 0)
    for bw in (1, 2):
        assert lib.amgh_debug_set_tunable(b"gs_bw", bw) == 0
        os.environ["AMGH_VERBOSE"] = "1" if bw == 1 else ""
        dev = AMG.DeviceHierarchy(ml, 0, 1)
        bd, zd = AMG.DeviceBuffer(n, 0, b), AMG.DeviceBuffer(n, 0)
        for _ in range(3): assert lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0) == 0
        lib.amgh_dev_sync(0)
        t0 = time.perf_counter()
        for _ in range(10): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        assert lib.amgh_dev_sync(0) == 0
        t = 1e3 * (time.perf_counter() - t0) / 10
        modes = [int(lib.amgh_debug_bw_mode(dev.h, l)) for l in range(min(4, len(ml.levels)))]
        pres = [dev.bench_op(l, 4, 3, 1) for l in range(min(3, len(ml.levels)))]
        print(f"N = {N} ({[l.A.m for l in ml.levels[:4]]} rows) gs_bw = {bw}: {t:7.3f} ms per V-cycle, block layouts on levels {modes}, pre-smoother of levels 0-2: " + " / ".join(f"{p:.3f}" for p in pres) + " ms", flush=True)
        del dev, bd, zd
lib.amgh_debug_set_tunable(b"gs_bw", 1); lib.amgh_debug_set_tunable(b"gs_bw_min_rows", 30000)
