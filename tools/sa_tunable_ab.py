"""Which schedule tunable moves the smoothed-aggregation V-cycle (symmetric sweeps, wide coarse rows)?"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy

lib = AMG.hip_lib()
s = int(sys.argv[1]) if len(sys.argv) > 1 else 160
A = AMG.poisson((s, s, s)); n = A.m
ml = AMG.smoothed_aggregation(A)
print("levels", [l.A.m for l in ml.levels], ml.final_A.m, "nnz", [l.A.nnz for l in ml.levels])
def run(tag, sets):
    for k, v in sets: lib.amgh_debug_set_tunable(k, v)
    dev = DeviceHierarchy(ml, 0, 1)
    bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n)); zd = AMG.DeviceBuffer(n, 0)
    for _ in range(2): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(10): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0); ms = 1e2 * (time.perf_counter() - t0)
    print(f"{tag}: {dev.device_bytes() / 1e9:.1f} GB V-cycle {ms:.2f} ms", flush=True)
    del dev, bd, zd; gc.collect()
run("default", [])
run("gs_bw=0", [(b"gs_bw", 0)])
run("gs_bw=0 lean=0", [(b"gs_lean", 0)])
run("gs_bw=0 lean=0 xcd=0", [(b"gs_xcd_map", 0)])
run("gs_bw=0 lean=0 xcd=1 tiny=0", [(b"gs_xcd_map", 1), (b"gs_tiny", 0)])
run("gs_bw=0 lean=0 tiny=2", [(b"gs_tiny", 2)])
run("gs_bw=0 lean=0 tiny=1 lpr=1", [(b"gs_tiny", 1), (b"gs_lpr", 1)])
