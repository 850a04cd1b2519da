#!/usr/bin/env python3
"""V-cycle time of the Jacobi(2/3)-smoothed ruge_stuben hierarchy on poisson(N^3) — the hierarchy that shards without the
dependency-level limit — with the x = 0 shortcut of the pre-smoothers on and off.
usage: python tools/jacobi_cycle.py [N=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
from bench import uniform

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = AMG.poisson((N, N, N))
jac = AMG.Jacobi(2.0 / 3.0)
ml = AMG.ruge_stuben(A, setup="gpu", presmoother=jac, postsmoother=jac)
n = A.m
dev = ml.device()
lib = dev.lib
bd, zd = AMG.DeviceBuffer(n, 0, uniform(n, 0)), AMG.DeviceBuffer(n, 0)
for flag in (1, 0, 1, 0):
    lib.amgh_debug_set_tunable(b"jacobi_zero", flag)
    for _ in range(3):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(20):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    ms = 1e3 * (time.perf_counter() - t0) / 20
    print(f"jacobi_zero={flag}: V-cycle {ms:.3f} ms = {n / ms / 1e3:.0f} M unknowns/s  (HBM {dev.device_bytes() / 1e9:.1f} GB)", flush=True)
lib.amgh_debug_set_tunable(b"jacobi_zero", 1)
