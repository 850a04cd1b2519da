#!/usr/bin/env python3
"""The row-sharded V-cycle with N virtual ranks of ONE process on one GPU (LOCAL transport: threads, peer copies between the
ranks' streams): ms per cycle by Gauss-Seidel mode, per number of sharded levels.  A functional measurement of the pipelined
exact sweep without the cost of several processes sharing a device.
usage: python tools/dist_local_bench.py [N=256] [nranks=4] [shard_min_rows=2000000]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # (virtual ranks: every rank's stream on a hardware queue of its own, tests/conftest.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from amg_amd import sharded as SH
from bench import uniform

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
smr = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu")
n = ml.levels[0].A.m
b = uniform(n, 0)
dev = ml.device()
bd, zd = AMG.DeviceBuffer(n, 0, b), AMG.DeviceBuffer(n, 0)
lib = dev.lib
for _ in range(2):
    lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
lib.amgh_dev_sync(0)
t0 = time.perf_counter()
for _ in range(5):
    lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
lib.amgh_dev_sync(0)
print(f"single handle: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per V-cycle", flush=True)
z_ref = zd.download()
for mode in ("exact", "exact-turns", "hybrid"):
    def work(rank, group):
        sh = SH.ShardedHierarchy.from_multilevel(ml, rank, nranks, 0, ("local", group), smr, gs_mode=mode)
        sh.set_rhs(b[sh.r0:sh.r1])
        for _ in range(2):
            sh.precond_apply_d(0)
        sh.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            sh.precond_apply_d(0)
        sh.barrier()
        dt = 1e3 * (time.perf_counter() - t0) / 5
        z = sh._down(sh._x)
        res = (dt, sh.lc, sh.gs_pipelined(), sh.r0, z)
        sh.close()
        return res
    res = SH.run_local_ranks(nranks, work)
    z = np.concatenate([r[4] for r in res])
    err = np.linalg.norm(z - z_ref) / np.linalg.norm(z_ref)
    print(f"{nranks} virtual ranks, gs_mode {mode:12s}: {max(r[0] for r in res):7.2f} ms per V-cycle, {res[0][1]} sharded levels, pipelined {res[0][2]}, "
          f"rel. diff to the single-handle cycle {err:.2e}", flush=True)
