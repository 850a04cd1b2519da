"""V-cycle of 2-D Poisson hierarchies (two offset classes: the wavefront of blocks is a line) by the size from which such a level takes
the block layout (tunable gs_bw_two_min_rows; 0 = never).   usage: python tools/two_dir_sweep.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from bench import uniform
Ns = [int(v) for v in sys.argv[1:]] or [512, 1024, 2048, 4096]
lib = AMG.hip_lib()
for N in Ns:
    ml = AMG.ruge_stuben(AMG.poisson((N, N)), setup="gpu")
    n = ml.levels[0].A.m
    b = uniform(n, 0)
    z0 = None
    for mr in (0, 6000000, 1000000, 200000, 30000):
        assert lib.amgh_debug_set_tunable(b"gs_bw_two_min_rows", mr) == 0
        dev = AMG.DeviceHierarchy(ml, 0, 1)
        bd, zd = AMG.DeviceBuffer(n, 0, b), AMG.DeviceBuffer(n, 0)
        for _ in range(3): assert lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0) == 0
        lib.amgh_dev_sync(0)
        t0 = time.perf_counter()
        for _ in range(6): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        assert lib.amgh_dev_sync(0) == 0
        t = 1e3 * (time.perf_counter() - t0) / 6
        z = zd.download()
        if z0 is None: z0 = z
        modes = [int(lib.amgh_debug_bw_mode(dev.h, l)) for l in range(min(5, len(ml.levels)))]
        print(f"N = {N}^2 ({[l.A.m for l in ml.levels[:5]]} rows) gs_bw_two_min_rows = {mr:8d}: {t:7.3f} ms per V-cycle, block layouts on levels {modes}, rel. diff to the first {np.linalg.norm(z - z0) / np.linalg.norm(z0):.1e}", flush=True)
        del dev, bd, zd
lib.amgh_debug_set_tunable(b"gs_bw_two_min_rows", 200000)
