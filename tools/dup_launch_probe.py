"""What could a prefetch of the NEXT merged group's arrays win?  Every group launch of a sweep issued twice (tunable
gs_dup_launch; a group's launch is idempotent): the repeat finds its composite rows in L2 / the memory-side cache.
t_warm = (T_doubled - T) / launches, against t_cold = T / launches.   usage: python tools/dup_launch_probe.py [N=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A, setup="gpu", device=0)
dev = ml.device(0, 1)
lib = dev.lib
n = A.m
bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n)); zd = AMG.DeviceBuffer(n, 0)
res = {}
for dup in (0, 1, 2):
    assert lib.amgh_debug_set_tunable(b"gs_dup_launch", dup) == 0
    for _ in range(3): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    dev.profile(True)
    for _ in range(4): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    prof = dev.profile_read(); dev.profile(False)
    pre = [prof[k] for k in prof if k.lower().startswith("pre")][0]; post = [prof[k] for k in prof if k.lower().startswith("post")][0]
    res[dup] = [(pre[l] + post[l]) / 4 for l in range(len(ml.levels))]
lib.amgh_debug_set_tunable(b"gs_dup_launch", 0)
for l in range(min(6, len(ml.levels))):
    la = dev.gs_sweep_stats(l, False)["launches"]
    if la <= 2: continue
    t0, t1, t2 = res[0][l], res[1][l], res[2][l]
    print(f"level {l} ({ml.levels[l].A.m} rows, {la} launches per sweep): smoothers {t0:.3f} ms per cycle, doubled {t1:.3f}, tripled {t2:.3f} -> "
          f"cold launch {1e3 * t0 / (4 * la):.2f} us, repeat {1e3 * (t1 - t0) / (4 * la):.2f} / {1e3 * (t2 - t1) / (4 * la):.2f} us")
