#!/usr/bin/env python3
"""A hierarchy well beyond the headline size on ONE GPU, memory-lean schedules (AMGH_LEAN=1): build, footprint by
category, V-cycle time, one cycle against the oracle, and the size-independent properties of the full-size tests.
usage: python tools/big_problem.py [N=384] [lean=1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
os.environ["AMGH_LEAN"] = sys.argv[2] if len(sys.argv) > 2 else "1"
import amg_amd as AMG  # noqa: E402
from bench import uniform  # noqa: E402

t0 = time.time()
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A)
print(f"N={N}: n={A.m} nnz={A.nnz} levels {[l.A.m for l in ml.levels]} setup {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
dev = ml.device()
print(f"upload + schedules {time.time() - t0:.1f} s   lean={os.environ['AMGH_LEAN']}", flush=True)
det = dev.device_bytes_detail()
print("HBM bytes:", dev.device_bytes(), {k: round(v / 1e9, 2) for k, v in det.items()}, "GB", flush=True)
n = A.m
lib = dev.lib
b = uniform(n, 0)
bd, zd = AMG.DeviceBuffer(n, 0, b), AMG.DeviceBuffer(n, 0)
for _ in range(2):
    lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
lib.amgh_dev_sync(0)
t0 = time.perf_counter()
for _ in range(5):
    lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
lib.amgh_dev_sync(0)
ms = 1e3 * (time.perf_counter() - t0) / 5
print(f"V-cycle {ms:.2f} ms = {n / ms / 1e3:.1f} M unknowns/s", flush=True)
p = AMG.aspreconditioner(ml)
z = p.ldiv(b)
r2 = uniform(n, 2) - 0.5
z2 = p.ldiv(r2)
rel = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)  # noqa: E731
print("linearity  :", rel(p.ldiv(b - 2.0 * r2), z - 2.0 * z2))
print("symmetry   :", abs(z @ r2 - b @ z2) / abs(z @ r2))
print("bitwise rerun:", bool(np.array_equal(p.ldiv(b), z)))
x, hist = AMG._solve(ml, b, reltol=1e-8, log=True)
print(f"_solve: {len(hist) - 1} cycles, residual {hist[-1] / hist[0]:.2e}, monotone {bool(np.all(np.diff(hist) < 0))}")
if "--oracle" in sys.argv:
    from oracle import oracle as O
    t0 = time.time()
    zo = O.OracleHierarchy(ml).precond(b)
    print(f"one V-cycle vs the oracle: rel.err {rel(z, zo):.2e}   (oracle cycle {time.time() - t0:.1f} s)")
