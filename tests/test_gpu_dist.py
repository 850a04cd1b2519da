"""Row-sharded driver with the REAL HipOps backend: N virtual ranks (threads) sharing the one GPU a
gpurun box has; the all-gather is emulated by device-to-device copies (tests/dist_backends.py).
The RCCL path itself needs a multi-GPU node (driver's SCALE run)."""
import numpy as np
import pytest

import amg_amd as AMG
import dist_mirror as D  # noqa: E402
from conftest import uniform
from dist_backends import run_virtual_ranks
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _solve_sharded(ml, b, nranks, shard_min_rows, cyc=0, **kw):
    def work(comm):
        ops = D.HipOps(0)
        dml = D.DistMultiLevel(ml, comm, ops, shard_min_rows=shard_min_rows)
        r0, r1 = dml.local_range(0)
        x, hist = dml.solve(b[r0:r1], cyc=cyc, **kw)
        return x, hist, dml.lc
    res = run_virtual_ranks(nranks, work)
    return np.concatenate([r[0] for r in res]), res[0][1], res[0][2]


@pytest.mark.parametrize("nranks", [2, 4])
def test_sharded_jacobi_is_exactly_the_single_gpu_cycle(nranks):
    A = AMG.poisson((32, 24, 20))
    b = uniform(A.m, 5)
    jac = AMG.Jacobi(2.0 / 3.0, iter=2)
    ml = AMG.ruge_stuben(A, presmoother=jac, postsmoother=jac)
    oh = O.OracleHierarchy(ml)
    for cyc in (0, 1, 2):
        x, hist, lc = _solve_sharded(ml, b, nranks, 500, cyc=cyc, reltol=1e-8, maxiter=60)
        assert lc >= 2
        xo, ho, _ = oh.solve(b, cycle=cyc, reltol=1e-8, maxiter=60)
        assert len(hist) == len(ho)
        assert np.linalg.norm(x - xo) <= 1e-10 * np.linalg.norm(xo)


def test_sharded_hybrid_gauss_seidel_converges_to_the_same_solution():
    A = AMG.poisson((32, 32, 32))
    b = uniform(A.m, 6)
    ml = AMG.ruge_stuben(A)
    x, hist, lc = _solve_sharded(ml, b, 4, 1000, reltol=1e-10, maxiter=60)
    xo, ho, _ = O.OracleHierarchy(ml).solve(b, reltol=1e-10, maxiter=60)
    assert lc >= 2 and hist[-1] <= 1e-10 * hist[0]
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)
    assert abs(len(hist) - len(ho)) <= 2


def test_eight_shards_as_on_an_8_gpu_node():
    """The driver's largest configuration (N = 8): 8 row shards, merged-level schedules on the local blocks, levels
    below the threshold collapsed onto rank 0."""
    A = AMG.poisson((64, 64, 64))
    b = uniform(A.m, 8)
    ml = AMG.ruge_stuben(A)
    x, hist, lc = _solve_sharded(ml, b, 8, 8000, reltol=1e-8, maxiter=60)
    xo, ho, _ = O.OracleHierarchy(ml).solve(b, reltol=1e-8, maxiter=60)
    assert lc >= 2 and hist[-1] <= 1e-8 * hist[0]
    assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
    assert abs(len(hist) - len(ho)) <= 2


def test_single_rank_sharded_path_equals_plain_solve():
    A = AMG.poisson((20, 20, 20))
    b = uniform(A.m, 7)
    ml = AMG.ruge_stuben(A)
    ops = D.HipOps(0)
    dml = D.DistMultiLevel(ml, D.SingleComm(), ops, shard_min_rows=500)
    x, hist = dml.solve(b, reltol=1e-8)
    xo, ho = AMG._solve(ml, b, reltol=1e-8, log=True)
    assert len(hist) == len(ho) and np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)
