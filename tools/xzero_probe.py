"""The dataflow sweep that starts a smooth! call on x = 0 without reading x (tunable gs_flow_xzero): V-cycle and pre-smoother
times off / on, results compared bitwise.   usage: python tools/xzero_probe.py [N=256] [bs=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A, setup="gpu", device=0)
dev = ml.device(0, bs)
lib = dev.lib
n = A.m
bd = AMG.DeviceBuffer(n * bs, 0, np.random.default_rng(0).random(n * bs)); zd = AMG.DeviceBuffer(n * bs, 0)
z = {}
for rnd in range(3):
    for on in (0, 1):
        assert lib.amgh_debug_set_tunable(b"gs_flow_xzero", on) == 0
        for _ in range(3): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0)
        t0 = time.perf_counter()
        for _ in range(20): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0)
        ms = 50 * (time.perf_counter() - t0)
        z[on] = zd.download()
        dev.profile(True)
        for _ in range(4): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0)
        prof = dev.profile_read(); dev.profile(False)
        pre = [prof[k] for k in prof if k.lower().startswith("pre")][0]
        print(f"gs_flow_xzero = {on}: V-cycle {ms:.3f} ms; pre-smoothers of levels 0 / 1: {pre[0] / 4:.3f} / {pre[1] / 4:.3f} ms", flush=True)
print("results bitwise equal:", bool(np.array_equal(z[0], z[1])))
lib.amgh_debug_set_tunable(b"gs_flow_xzero", 1)
