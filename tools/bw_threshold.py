"""Smallest operator worth laying out as a (chained) wavefront of blocks: V-cycle time of ruge_stuben(poisson(N^3)) by the
tunable gs_bw_min_rows (levels with fewer rows keep the merged dependency-level groups)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy

lib = AMG.hip_lib()
sizes = [int(s) for s in sys.argv[1:]] or [96, 128, 160, 192]
for s in sizes:
    A = AMG.poisson((s, s, s)); n = A.m
    ml = AMG.ruge_stuben(A, setup="gpu")
    rows = [l.A.m for l in ml.levels]
    for thr in (1 << 30, 3000000, 1000000, 250000, 60000):
        lib.amgh_debug_set_tunable(b"gs_bw_min_rows", thr)
        dev = DeviceHierarchy(ml, 0, 1)
        bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n)); zd = AMG.DeviceBuffer(n, 0)
        for _ in range(3): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0)
        rounds = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(5): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
            lib.amgh_dev_sync(0); rounds.append(2e2 * (time.perf_counter() - t0))
        used = [l for l in range(min(4, len(rows))) if dev.gs_sweep_stats(l, False)["launches"] == 1 and dev.gs_sweep_stats(l, False)["tri_entries"] == 0 and rows[l] > 100000]
        print(f"N={s} rows {rows[:4]} gs_bw_min_rows={thr}: V-cycle {min(rounds):.3f} ms, {dev.device_bytes() / 1e9:.2f} GB, block levels {used}", flush=True)
        del dev, bd, zd; gc.collect()
lib.amgh_debug_set_tunable(b"gs_bw_min_rows", 3000000)
