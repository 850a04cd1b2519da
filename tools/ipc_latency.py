#!/usr/bin/env python3
"""Cost of ONE halo exchange of the row-sharded path: 2 ranks on one GPU, fine-level SpMV of a small problem (the launch is
microseconds: the call is dominated by the exchange), IPC transport (two PROCESSES, stream-written / stream-awaited flags)
vs LOCAL transport (two threads of one process, host rendezvous).

    python tools/ipc_latency.py            (spawns its own second process for the IPC case)
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amg_amd as AMG  # noqa: E402
from amg_amd import sharded as SH  # noqa: E402

N, REPS = 48, 400


def problem():
    A = AMG.poisson((N, N, N))
    jac = AMG.Jacobi(2.0 / 3.0)
    return AMG.ruge_stuben(A, presmoother=jac, postsmoother=jac, max_levels=3)


def timed(sh):
    y = AMG.DeviceBuffer(max(sh.nloc, 1), 0)
    for _ in range(20):
        sh.lib.amgh_dist_spmv_d(sh.h, 0, None, y.ptr)
    sh.barrier()
    t0 = time.perf_counter()
    for _ in range(REPS):
        sh.lib.amgh_dist_spmv_d(sh.h, 0, None, y.ptr)
    sh.barrier()
    t_spmv = (time.perf_counter() - t0) / REPS
    sh.set_rhs(np.ones(sh.nloc))
    for _ in range(5):
        sh.precond_apply_d(0)
    sh.barrier()
    sh.stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(50):
        sh.precond_apply_d(0)
    sh.barrier()
    t_cyc = (time.perf_counter() - t0) / 50
    ex = sh.stats()["halo_exchanges"] / 50
    return t_spmv, t_cyc, ex


if len(sys.argv) > 1 and sys.argv[1] == "worker":
    rank, name = int(sys.argv[2]), sys.argv[3]
    sh = SH.ShardedHierarchy.from_multilevel(problem(), rank, 2, 0, ("ipc", name), 2000)
    t_spmv, t_cyc, ex = timed(sh)
    if rank == 0:
        print(f"IPC   (2 processes): SpMV + exchange {1e6 * t_spmv:7.1f} us per call; V-cycle {1e3 * t_cyc:.3f} ms with {ex:.0f} exchanges", flush=True)
    sh.close()
else:
    ml = problem()

    def work(rank, group):
        sh = SH.ShardedHierarchy.from_multilevel(ml, rank, 2, 0, ("local", group), 2000)
        return timed(sh)
    r = SH.run_local_ranks(2, work)
    print(f"LOCAL (2 threads)  : SpMV + exchange {1e6 * r[0][0]:7.1f} us per call; V-cycle {1e3 * r[0][1]:.3f} ms with {r[0][2]:.0f} exchanges", flush=True)
    one = SH.run_local_ranks(1, lambda rank, group: timed(SH.ShardedHierarchy.from_multilevel(ml, 0, 1, 0, ("local", group), 2000)))
    print(f"one rank (no exchange): SpMV {1e6 * one[0][0]:7.1f} us per call; V-cycle {1e3 * one[0][1]:.3f} ms", flush=True)
    name = "/amgh_lat_%d" % os.getpid()
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r_), name], cwd=ROOT) for r_ in range(2)]
    [p.wait(timeout=300) for p in ps]
