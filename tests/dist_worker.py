"""Worker of tests/test_distributed_cpu.py: one rank of a gloo world (CPU).  Runs the row-sharded
driver with the CPU test backend and checks it against the single-process oracle."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import amg_amd as AMG  # noqa: E402
import dist_mirror as D  # noqa: E402
from conftest import uniform  # noqa: E402
from dist_backends import CpuOps  # noqa: E402
from oracle import oracle as O  # noqa: E402


def gather_x(comm, x_local):
    parts = [None] * comm.world_size
    dist.all_gather_object(parts, x_local)
    return np.concatenate(parts)


def main():
    dist.init_process_group("gloo")
    comm = D.TorchComm()
    rank = comm.rank
    ops = CpuOps()
    A = AMG.poisson((14, 12, 10))
    n = A.m
    b = uniform(n, 5)
    # 1. exact-parity configuration: Jacobi smoothers (no cross-shard ordering), V / W / F cycles
    jac = AMG.Jacobi(2.0 / 3.0, iter=2)
    ml = AMG.ruge_stuben(A, presmoother=jac, postsmoother=jac)
    oh = O.OracleHierarchy(ml)
    dml = D.DistMultiLevel(ml, comm, ops, shard_min_rows=100)
    assert dml.lc >= 2, dml.lc            # at least two sharded levels + a collapsed tail
    r0, r1 = dml.local_range(0)
    for cyc in (0, 1, 2):
        x_loc, hist = dml.solve(b[r0:r1], cyc=cyc, reltol=1e-8, maxiter=60)
        x = gather_x(comm, x_loc)
        xo, ho, _ = oh.solve(b, cycle=cyc, reltol=1e-8, maxiter=60)
        assert len(hist) == len(ho), (cyc, len(hist), len(ho))
        err = np.linalg.norm(x - xo) / np.linalg.norm(xo)
        assert err < 1e-12, (cyc, err)
        assert np.allclose(hist, ho, rtol=1e-9)
    # ldiv! semantics
    dml.set_rhs(b[r0:r1])
    dml.precond_apply(0)
    z = gather_x(comm, ops.download(dml.x[0], r1 - r0))
    zo = oh.precond(b)
    assert np.linalg.norm(z - zo) / np.linalg.norm(zo) < 1e-12
    # 2. default smoother (symmetric Gauss-Seidel): processor-block hybrid across shards — not the
    #    same iterates, the same solution: both converge to A^-1 b within the solver tolerance
    ml = AMG.ruge_stuben(A)
    dml = D.DistMultiLevel(ml, comm, ops, shard_min_rows=100)
    x_loc, hist = dml.solve(b[r0:r1], reltol=1e-10, maxiter=60)
    x = gather_x(comm, x_loc)
    xo, ho, _ = O.OracleHierarchy(ml).solve(b, reltol=1e-10, maxiter=60)
    assert hist[-1] <= 1e-10 * hist[0]
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-8
    assert abs(len(hist) - len(ho)) <= 2, (len(hist), len(ho))
    # 3. everything collapsed onto rank 0 (problem smaller than the shard threshold)
    dml = D.DistMultiLevel(ml, comm, ops, shard_min_rows=10 ** 9)
    assert dml.lc == 0
    x_loc, hist = dml.solve(b[:n] if rank == 0 else b[:0], reltol=1e-8)
    if rank == 0:
        assert np.linalg.norm(x_loc - O.OracleHierarchy(ml).solve(b, reltol=1e-8)[0]) < 1e-10 * np.linalg.norm(x_loc)
    dist.barrier()
    if rank == 0:
        print("DIST_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
