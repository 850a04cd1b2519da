"""V-cycle (and bs = 8 cycle) of the 256^3 hierarchy by the rows per block the wavefront-of-blocks plan aims at (tunable gs_bw_rows, read
when a schedule is built).   usage: python tools/rows_sweep.py [N=256] [rows ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from bench import uniform
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows = [int(v) for v in sys.argv[2:]] or [512, 640, 729, 768, 900, 1000]
lib = AMG.hip_lib()
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu", device=0)
n = ml.levels[0].A.m
b = uniform(n, 0)
z0 = None
for r in rows:
    assert lib.amgh_debug_set_tunable(b"gs_bw_rows", r) == 0
    for bs in (1, 8):
        t0 = time.perf_counter()
        dev = AMG.DeviceHierarchy(ml, 0, bs)
        tb = time.perf_counter() - t0
        Bh = np.stack([b] + [uniform(n, 100 + c) for c in range(1, bs)], axis=1)
        Bd = AMG.DeviceBuffer(n * bs, 0, np.asfortranarray(Bh).ravel(order="F")); Zd = AMG.DeviceBuffer(n * bs, 0)
        for _ in range(3): assert lib.amgh_precond_apply_d(dev.h, Bd.ptr, Zd.ptr, 0) == 0
        lib.amgh_dev_sync(0)
        reps = 10 if bs == 1 else 5
        t0 = time.perf_counter()
        for _ in range(reps): lib.amgh_precond_apply_d(dev.h, Bd.ptr, Zd.ptr, 0)
        assert lib.amgh_dev_sync(0) == 0
        t = 1e3 * (time.perf_counter() - t0) / reps
        z = Zd.download()[:n]
        if z0 is None: z0 = z
        extra = f"L0 / L1 presmooth {dev.bench_op(0, 4, 3, 1):.3f} / {dev.bench_op(1, 4, 3, 1):.3f} ms, sweep steps {dev.gs_sweep_steps(0)} / {dev.gs_sweep_steps(1)}" if bs == 1 else ""
        print(f"gs_bw_rows = {r:5d} bs = {bs}: {t:7.3f} ms per cycle (layout {tb:.1f} s), first column bitwise the first run's {bool((z == z0).all())} {extra}", flush=True)
        del dev, Bd, Zd
lib.amgh_debug_set_tunable(b"gs_bw_rows", 512)
