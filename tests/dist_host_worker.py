"""Worker of tests/test_distributed_cpu.py::test_gloo_drives_the_library_sharded_cycle_on_the_host: one rank of a gloo world
(CPU, launched by torch.distributed.run).  The row-sharded V-cycle runs through libamghip's OWN `amgh_dist_*` code — halo
plans, exchange ordering, the turns of the exact Gauss-Seidel, collapse onto rank 0, all-reduces — executed in host memory
(device = -1 + amgh_dist_set_host_tail: the operators are plain loops, the IPC transport's shared-memory rendezvous carries
the exchanges), and is checked against the single-process oracle.  gloo launches the ranks, hands out the segment name and
gathers the results; the collapsed levels are the oracle's cycle on rank 0 (test infrastructure on the test side)."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import amg_amd as AMG  # noqa: E402
from amg_amd import sharded as SH  # noqa: E402
from conftest import uniform  # noqa: E402
from oracle import oracle as O  # noqa: E402
from sharded_emulation import emulate_sharded_cycles  # noqa: E402


def gather(x_local):
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, x_local)
    return np.concatenate(parts)


def rel(x, y):
    return np.linalg.norm(x - y) / np.linalg.norm(y)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    A = AMG.poisson((14, 12, 10))
    b = uniform(A.m, 5)
    tail_of = lambda tail: (lambda bb: O.OracleHierarchy(tail).precond(bb))   # noqa: E731  (one visit of the collapsed levels from x = 0)
    seq = [0]

    def sharded(ml, gs_mode):
        box = ["/amgh_h_%d_%d_%s" % (os.getppid(), seq[0], os.urandom(3).hex())] if rank == 0 else [None]
        seq[0] += 1
        dist.broadcast_object_list(box, src=0)
        return SH.ShardedHierarchy.from_multilevel(ml, rank, world, -1, ("ipc", box[0]), 100, gs_mode=gs_mode, host_tail=tail_of)

    # 1. Jacobi smoothers: exact across shards — cycles, the solve with its residual history, ldiv!, the sharded SpMV
    jac = AMG.Jacobi(2.0 / 3.0, iter=2)
    ml = AMG.ruge_stuben(A, presmoother=jac, postsmoother=jac)
    oh = O.OracleHierarchy(ml)
    sh = sharded(ml, "exact")
    assert sh.lc >= 2 and sh.host_exec
    bl = b[sh.r0:sh.r1]
    x_loc, hist = sh.solve(bl, reltol=1e-8, maxiter=60)
    xo, ho, _ = oh.solve(b, reltol=1e-8, maxiter=60)
    assert len(hist) == len(ho) and np.allclose(hist, ho, rtol=1e-9), (len(hist), len(ho))
    assert rel(gather(x_loc), xo) <= 1e-10
    assert rel(gather(sh.precond_apply(bl)), oh.precond(b)) <= 1e-10
    assert rel(gather(sh.spmv(0, bl)), A @ b) <= 1e-13
    st = sh.stats()
    assert st["halo_exchanges"] > 0 and st["halo_bytes_sent"] > 0
    sh.close()
    # 2. the default smoother, exact order across the shards (the ranks in turn): every cycle is the oracle's
    ml = AMG.ruge_stuben(A)
    oh = O.OracleHierarchy(ml)
    sh = sharded(ml, "exact")
    for k in (1, 2, 3):
        x_loc, _ = sh.solve(bl, maxiter=k, calculate_residual=False)
        xo, _, _ = oh.solve(b, maxiter=k, calculate_residual=False)
        assert rel(gather(x_loc), xo) <= 1e-10, k
    x_loc, hist = sh.solve(bl, reltol=1e-10, maxiter=60)
    xo, ho, _ = oh.solve(b, reltol=1e-10, maxiter=60)
    assert len(hist) == len(ho) and np.allclose(hist, ho, rtol=1e-8) and rel(gather(x_loc), xo) <= 1e-10
    lc = sh.lc
    sh.close()
    # 3. the hybrid (every shard at once, halo frozen per directional sweep) against its host emulation; SOR forward / backward
    sh = sharded(ml, "hybrid")
    want = emulate_sharded_cycles(ml, b, world, lc, 2)
    for k in (1, 2):
        x_loc, _ = sh.solve(bl, maxiter=k, calculate_residual=False)
        assert rel(gather(x_loc), want[k - 1]) <= 1e-10, k
    sh.close()
    ml = AMG.ruge_stuben(A, presmoother=AMG.SOR(1.2, AMG.ForwardSweep()), postsmoother=AMG.SOR(0.9, AMG.BackwardSweep()))
    sh = sharded(ml, "exact")
    x_loc, _ = sh.solve(bl, maxiter=2, calculate_residual=False)
    xo, _, _ = O.OracleHierarchy(ml).solve(b, maxiter=2, calculate_residual=False)
    assert rel(gather(x_loc), xo) <= 1e-10
    sh.close()
    dist.barrier()
    if rank == 0:
        print("DIST_HOST_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
