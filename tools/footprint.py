#!/usr/bin/env python3
"""HBM footprint of the 256^3 hierarchy by category, default vs memory-lean, with the V-cycle time of each.

    python tools/footprint.py [size=256] [mode ...]     modes: default lean   (AMGH_LEAN is read at schedule build)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG  # noqa: E402
from bench import uniform  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
modes = sys.argv[2:] or ["default", "lean"]
A = AMG.poisson((N, N, N))
n = A.m
for mode in modes:
    lib = AMG.hip_lib()
    lib.amgh_debug_set_tunable(b"gs_lean", {"default": -1, "lean": 1, "full": 0}[mode])
    t0 = time.perf_counter()
    ml = AMG.ruge_stuben(A, setup="gpu", device=0)
    dev = ml.device()
    t_setup = time.perf_counter() - t0
    bd = AMG.DeviceBuffer(n, 0, uniform(n, 0))
    zd = AMG.DeviceBuffer(n, 0)
    for _ in range(3):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(10):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    ms = 1e2 * (time.perf_counter() - t0)
    det = dev.device_bytes_detail()
    print(json.dumps({"mode": mode, "size": N, "hbm_GB": round(dev.device_bytes() / 1e9, 2), "vcycle_ms": round(ms, 3),
                      "setup_s": round(t_setup, 2), "GB_by_category": {k: round(v / 1e9, 2) for k, v in det.items()}}), flush=True)
    z = zd.download()
    print("   checksum", float(z.sum()), float(abs(z).max()), flush=True)
    del dev, ml, bd, zd
