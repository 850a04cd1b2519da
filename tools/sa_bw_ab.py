"""A/B of the fine-level Gauss-Seidel schedule (merged groups vs wavefront of blocks) on smoothed-aggregation hierarchies,
whose default smoother is the symmetric sweep (forward + backward, one launch sequence each)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy

lib = AMG.hip_lib()
sizes = [int(s) for s in sys.argv[1:]] or [160, 256]
for s in sizes:
    A = AMG.poisson((s, s, s)); n = A.m
    t0 = time.perf_counter(); ml = AMG.smoothed_aggregation(A); ts = time.perf_counter() - t0
    for bw in (0, 1, 2):
        lib.amgh_debug_set_tunable(b"gs_bw", bw)
        t0 = time.perf_counter(); dev = DeviceHierarchy(ml, 0, 1)
        bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n)); zd = AMG.DeviceBuffer(n, 0)
        for _ in range(2): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0); t_up = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(10): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0); ms = 1e2 * (time.perf_counter() - t0)
        print(f"SA poisson({s}^3) n={n} levels={len(ml.levels)} setup {ts:.1f}s gs_bw={bw}: build {t_up:.1f}s "
              f"{dev.device_bytes() / 1e9:.1f} GB V-cycle {ms:.2f} ms", flush=True)
        del dev, bd, zd; gc.collect()
    lib.amgh_debug_set_tunable(b"gs_bw", 1)
