import os, sys, time
os.environ["AMGH_VERBOSE"]="1"
sys.path.insert(0, "/root/repo")
import numpy as np
import amg_amd as AMG
A = AMG.poisson((4096, 4096))
ml = AMG.ruge_stuben(A, setup="gpu")
dev = ml.device()
lib = dev.lib
n = A.m
b = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n)); z = AMG.DeviceBuffer(n, 0)
for _ in range(3): lib.amgh_precond_apply_d(dev.h, b.ptr, z.ptr, 0)
lib.amgh_dev_sync(0)
t0 = time.perf_counter()
for _ in range(10): lib.amgh_precond_apply_d(dev.h, b.ptr, z.ptr, 0)
lib.amgh_dev_sync(0)
print("V-cycle %.2f ms" % ((time.perf_counter() - t0) * 100))
dev.profile(True)
lib.amgh_precond_apply_d(dev.h, b.ptr, z.ptr, 0)
pr = dev.profile_read()
L1 = len(ml.levels) + 1
print("levels", [l.A.m for l in ml.levels], "deps", [dev.gs_dependency_levels(l) for l in range(len(ml.levels))])
for lab, v in pr.items(): print("%-14s" % lab, " ".join("%7.3f" % x for x in v))
print("launches fwd", [dev.gs_sweep_stats(l)["launches"] for l in range(len(ml.levels))])
