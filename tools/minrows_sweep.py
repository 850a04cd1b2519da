"""V-cycle of smaller 3-D Poisson hierarchies by the size from which a level takes the wavefront-of-blocks layout (tunable
gs_bw_min_rows, read when a schedule is built; operators of at most 7 entries per row qualify from half of it).
usage: python tools/minrows_sweep.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from bench import uniform
Ns = [int(v) for v in sys.argv[1:]] or [96, 128, 160, 192]
lib = AMG.hip_lib()
for N in Ns:
    ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu")
    n = ml.levels[0].A.m
    b = uniform(n, 0)
    z0 = None
    for mr in (3000000, 1500000, 400000, 100000, 30000):
        assert lib.amgh_debug_set_tunable(b"gs_bw_min_rows", mr) == 0
        dev = AMG.DeviceHierarchy(ml, 0, 1)
        bd, zd = AMG.DeviceBuffer(n, 0, b), AMG.DeviceBuffer(n, 0)
        for _ in range(3): assert lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0) == 0
        lib.amgh_dev_sync(0)
        t0 = time.perf_counter()
        for _ in range(10): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        assert lib.amgh_dev_sync(0) == 0
        t = 1e3 * (time.perf_counter() - t0) / 10
        z = zd.download()
        if z0 is None: z0 = z
        modes = [int(lib.amgh_debug_bw_mode(dev.h, l)) for l in range(min(4, len(ml.levels)))]
        print(f"N = {N} ({[l.A.m for l in ml.levels[:4]]} rows) gs_bw_min_rows = {mr:8d}: {t:7.3f} ms per V-cycle, block layouts on levels {modes}, rel. diff to the first {np.linalg.norm(z - z0) / np.linalg.norm(z0):.1e}", flush=True)
        del dev, bd, zd
lib.amgh_debug_set_tunable(b"gs_bw_min_rows", 30000)
