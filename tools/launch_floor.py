#!/usr/bin/env python3
"""How long does a dependent chain of trivial kernels take on this box? (floor for per-level launches)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
lib = AMG.hip_lib()
idx = AMG.DeviceBuffer(4, 0, np.zeros(4)); src = AMG.DeviceBuffer(4, 0, np.ones(4)); dst = AMG.DeviceBuffer(4, 0)
for n in (1, 256, 65536):
    i32 = np.zeros(max(n, 4), dtype=np.int32)
    ib = AMG.DeviceBuffer((max(n, 4) + 1) // 2, 0); lib.amgh_dev_upload(0, ib.ptr, i32.ctypes.data, 4 * max(n, 4))
    s = AMG.DeviceBuffer(max(n, 4), 0, np.ones(max(n, 4))); d = AMG.DeviceBuffer(max(n, 4), 0)
    for reps in (2000,):
        for _ in range(100): lib.amgh_gather_d(0, n, ib.ptr, s.ptr, d.ptr, None)
        lib.amgh_dev_sync(0); t0 = time.perf_counter()
        for _ in range(reps): lib.amgh_gather_d(0, n, ib.ptr, s.ptr, d.ptr, None)
        lib.amgh_dev_sync(0); t = time.perf_counter() - t0
        print(f"gather kernel n={n}: {1e6 * t / reps:.2f} us per dependent launch (null stream, host-driven)")
