"""One smoothed-aggregation hierarchy, 10 timed V-cycles (run under rocprofv3 --kernel-trace --stats for the kernel split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy

lib = AMG.hip_lib()
s = int(sys.argv[1]) if len(sys.argv) > 1 else 160
for kv in sys.argv[2:]:
    k, v = kv.split("="); lib.amgh_debug_set_tunable(k.encode(), int(v))
A = AMG.poisson((s, s, s)); n = A.m
ml = AMG.smoothed_aggregation(A)
dev = DeviceHierarchy(ml, 0, 1)
bd = AMG.DeviceBuffer(n, 0, np.random.default_rng(0).random(n)); zd = AMG.DeviceBuffer(n, 0)
for _ in range(3): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
lib.amgh_dev_sync(0)
if os.environ.get("SA_FIRST_CYCLES"):      # the cycles right after the build, one by one (a one-off 60 ms stall was seen there)
    one = []
    for _ in range(16):
        t0 = time.perf_counter(); lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0); lib.amgh_dev_sync(0)
        one.append(1e3 * (time.perf_counter() - t0))
    print("first cycles, one by one (ms):", " ".join(f"{t:.1f}" for t in one))
rounds = []
for _ in range(4):
    t0 = time.perf_counter()
    for _ in range(10): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0); rounds.append(1e2 * (time.perf_counter() - t0))
ms = min(rounds)
print("rounds of 10 cycles (ms per cycle):", " ".join(f"{r:.2f}" for r in rounds))
print(f"SA poisson({s}^3) {sys.argv[2:]} levels {[l.A.m for l in ml.levels]} V-cycle {ms:.2f} ms", flush=True)
