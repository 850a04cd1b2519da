#!/usr/bin/env python3
"""Per-level / per-label time breakdown of one V-cycle (hipEvent labels = the reference's
TimerOutputs labels, multilevel.jl:180,216-236).  Measurement tool.

    python tools/vcycle_profile.py [size=256] [cycles=3]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG  # noqa: E402
from bench import uniform  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.time()
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A)
print(f"setup {time.time() - t0:.1f}s levels", [l.A.m for l in ml.levels], ml.final_A.m)
dev = ml.device()
n = A.m
bd = AMG.DeviceBuffer(n, 0, uniform(n, 0))
zd = AMG.DeviceBuffer(n, 0)
lib = dev.lib
for graph in (0, 1):
    lib.amgh_set_use_graph(dev.h, graph)
    t0 = time.perf_counter()
    for _ in range(2):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(cycles):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    print(f"V-cycle ({'hipGraph' if graph else 'eager'}): {1e3 * (time.perf_counter() - t0) / cycles:.2f} ms "
          f"(2 warm-up cycles incl. capture: {t_warm:.2f} s)")
dev.profile(True)
for _ in range(cycles):
    lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
lib.amgh_dev_sync(0)
prof = dev.profile_read()
L = len(ml.levels)
deps = [dev.gs_dependency_levels(l) for l in range(L)]
print("%-3s %10s %10s %6s | %s" % ("lvl", "rows", "nnz", "deps", "  ".join("%-13s" % k[:13] for k in prof)))
tot = 0.0
for l in range(L + 1):
    rows = ml.levels[l].A.m if l < L else ml.final_A.m
    nnz = ml.levels[l].A.nnz if l < L else ml.final_A.nnz
    vals = [prof[k][l] / cycles for k in prof]
    tot += sum(vals)
    print("%-3d %10d %10d %6s | %s" % (l, rows, nnz, deps[l] if l < L else "-", "  ".join("%10.3f ms" % v for v in vals)))
print(f"sum of labelled steps: {tot:.2f} ms per V-cycle")
for l in range(min(L, 4)):
    print(f"level {l}: SpMV {dev.bench_op(l, 0, 20, 3):.4f} ms  residual {dev.bench_op(l, 3, 20, 3):.4f} ms  "
          f"R {dev.bench_op(l, 2, 20, 3):.4f} ms  P {dev.bench_op(l, 1, 20, 3):.4f} ms  presmooth {dev.bench_op(l, 4, 3, 1):.3f} ms")

# chain-kernel phase timing (shader cycles) per hierarchy level
import ctypes as C
out = (C.c_ulonglong * 8)()
lib.amgh_debug_chain_timing(1, None)
xb = AMG.DeviceBuffer(n, 0, uniform(n, 3))
print("chain kernel phases, avg shader cycles per dependency level: issue | gather+stage | barrier1 | rowsum+store | fence+barrier2 | levels launches")
for l in range(L):
    nl = ml.levels[l].A.m
    xl = AMG.DeviceBuffer(nl, 0, uniform(nl, 3)); bl = AMG.DeviceBuffer(nl, 0, uniform(nl, 4))
    lib.amgh_debug_chain_timing(1, out)
    lib.amgh_level_smooth_d(dev.h, l, 0, xl.ptr, bl.ptr)
    lib.amgh_debug_chain_timing(1, out)
    v = list(out)
    if v[5]:
        print(f"level {l}: " + " | ".join(f"{v[q] / v[5]:8.0f}" for q in range(5)) + f" | {v[5]} {v[6]} | of issue: wait+gather-issue {v[7] / v[5]:6.0f}")
lib.amgh_debug_chain_timing(0, None)
