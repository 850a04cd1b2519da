#!/usr/bin/env python3
"""Time one presmoother application per hierarchy level (256^3 RS hierarchy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = AMG.poisson((N, N, N)); ml = AMG.ruge_stuben(A); dev = ml.device()
ts = [dev.bench_op(l, 4, 3, 1) for l in range(len(ml.levels))]
print(os.environ.get("HIP_FORCE_DEV_KERNARG"), " ".join(f"L{l} {t:7.3f}" for l, t in enumerate(ts)), f"sum {sum(ts):.3f} ms")
