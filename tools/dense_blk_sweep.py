#!/usr/bin/env python3
"""Rows per dense diagonal block of the dense triangular sweeps (tunable gs_dense_blk, read at schedule build): V-cycle
time and the 38 k-row level's smoother time on the 256^3 hierarchy.   usage: python tools/dense_blk_sweep.py [blk ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy
from bench import uniform
sizes = [int(a) for a in sys.argv[1:]] or [4096, 2048, 1024]
A = AMG.poisson((256, 256, 256))
ml = AMG.ruge_stuben(A, setup="gpu")
lib = AMG.hip_lib()
n = A.m
bd = AMG.DeviceBuffer(n, 0, uniform(n, 0)); zd = AMG.DeviceBuffer(n, 0)
for blk in sizes:
    lib.amgh_debug_set_tunable(b"gs_dense_blk", blk)
    dev = DeviceHierarchy(ml, 0, 1)
    for _ in range(2): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(10): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    ms = (time.perf_counter() - t0) * 100
    dev.profile(True)
    for _ in range(3): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    pr = dev.profile_read(); dev.profile(False)
    print(f"gs_dense_blk={blk}: V-cycle {ms:.2f} ms; smoothers of levels 3..6 (pre+post, ms):",
          [round(float(pr['Presmoother'][l] + pr['Postsmoother'][l]) / 3, 3) for l in range(3, 7)],
          "checksum", float(np.abs(zd.download()).sum()), flush=True)
    del dev
lib.amgh_debug_set_tunable(b"gs_dense_blk", 4096)
