#!/usr/bin/env python3
"""Sweep the launch-shape tunables of the per-level Gauss-Seidel launches on the 256^3 hierarchy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = AMG.poisson((N, N, N)); ml = AMG.ruge_stuben(A); dev = ml.device(); lib = dev.lib
print("levels", [l.A.m for l in ml.levels])
for xm in (1, 1):
    lib.amgh_debug_set_tunable(b"gs_block_pipe", xm)
    ts = [dev.bench_op(l, 4, 3, 1) for l in range(7)]
    print(f"block_pipe={xm}: " + "  ".join(f"L{l} {t:7.3f}" for l, t in enumerate(ts)) + f"   sum {sum(ts):7.3f} ms", flush=True)
