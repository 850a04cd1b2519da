"""V-cycle time and per-level smoother-pass time for several settings of the zoned-group cost constants
(gs_zone, gs_zone_t0_ns, gs_zone_floor_ns: read when a schedule is built).
usage: python tools/zone_sweep.py [N=256] cfg cfg ...     cfg = name=v[,name=v...]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfgs = sys.argv[2:] or ["gs_zone=0", "gs_zone=1"]
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A, setup="gpu")
lib = AMG.hip_lib()
n = A.m
b = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n)); z = AMG.DeviceBuffer(n, 0)
lv = [l for l in range(len(ml.levels)) if ml.levels[l].A.m >= 4096]
for cfg in cfgs:
    for kv in cfg.split(","):
        k, v = kv.split("=")
        lib.amgh_debug_set_tunable(k.encode(), int(v))
    dev = DeviceHierarchy(ml, 0, 1)
    for _ in range(3):
        lib.amgh_precond_apply_d(dev.h, b.ptr, z.ptr, 0)
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(10):
        lib.amgh_precond_apply_d(dev.h, b.ptr, z.ptr, 0)
    lib.amgh_dev_sync(0)
    ms = (time.perf_counter() - t0) * 100
    st = [dev.gs_sweep_stats(l) for l in lv]
    ts = [dev.bench_op(l, 4, 3, 1) for l in lv]
    print(f"{cfg:44s} V-cycle {ms:6.2f} ms | launches fwd " + " ".join(str(s['launches']) for s in st) +
          " | entries(M) " + " ".join(f"{s['entries'] / 1e6:.0f}" for s in st) +
          " | pass ms " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    del dev
    gc.collect()
