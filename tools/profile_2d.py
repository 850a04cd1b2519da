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

# wavefront of blocks forced on (tunable gs_bw = 2: the cost model otherwise insists on three offset classes)
reps, L = 10, len(ml.levels)
if os.environ.get("AMG_2D_FORCE_BW"):
    AMG.hip_lib().amgh_debug_set_tunable(b"gs_bw", 2)
    if os.environ.get("AMG_2D_BW_ROWS"):
        AMG.hip_lib().amgh_debug_set_tunable(b"gs_bw_rows", int(os.environ["AMG_2D_BW_ROWS"]))
    ml2 = AMG.ruge_stuben(A, setup="gpu", device=0) if os.environ.get("AMG_2D_GPU_SETUP") else AMG.ruge_stuben(A)
    dev2 = ml2.device(0, 1)
    bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n))
    zd = AMG.DeviceBuffer(n, 0)
    for _ in range(3): lib.amgh_precond_apply_d(dev2.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(reps): lib.amgh_precond_apply_d(dev2.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    print(f"gs_bw = 2 (rows per block {os.environ.get('AMG_2D_BW_ROWS', 'default')}): V-cycle {1e3 * (time.perf_counter() - t0) / reps:.2f} ms; "
          f"modes by level {[lib.amgh_debug_bw_mode(dev2.h, l) for l in range(min(L, 6))]}", flush=True)
    dev2.profile(True)
    for _ in range(3): lib.amgh_precond_apply_d(dev2.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    prof2 = dev2.profile_read()
    for l in range(min(L, 6)):
        print("%-3d %10d %6d | %s" % (l, ml2.levels[l].A.m, dev2.gs_dependency_levels(l), "  ".join("%9.3f ms" % (prof2[k][l] / 3) for k in prof2)))
