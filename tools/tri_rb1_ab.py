"""Dense triangle inverses under a single column: rows per workgroup (tunable gs_tri_rb1: 0 = tri_gemv_kernel, 2 / 4 / 8 =
tri_gemm_kernel<1, RB>, bitwise the same).  python tools/tri_rb1_ab.py [N]  — the V-cycle of the N^3 ruge_stuben hierarchy."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amg_amd as AMG  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = AMG.hip_lib()
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A, setup="gpu", device=0)
dev = ml.device()
n = A.m
bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n))
zd = AMG.DeviceBuffer(n, 0)
lib.amgh_profile_enable.argtypes = None
ref = None
for rep in range(2):
    TUN = sys.argv[2].encode() if len(sys.argv) > 2 else b"gs_tri_rb1"
    VALS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else (0, 2, 4, 8)
    for rb in VALS:
        lib.amgh_debug_set_tunable(TUN, rb)
        for _ in range(3):
            lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0)
        t0 = time.perf_counter()
        for _ in range(20):
            lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0)
        ms = 1e3 * (time.perf_counter() - t0) / 20
        z = zd.download()
        if ref is None:
            ref = z
        print(f"{TUN.decode()} = {rb}: V-cycle {ms:.3f} ms; bitwise the one-row kernel: {bool(np.array_equal(z, ref))}", flush=True)
lib.amgh_debug_set_tunable(TUN, 0)
