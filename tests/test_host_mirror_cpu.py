"""Host-side pieces added in round 2 that need no GPU: the row partition / level export of the sharded binding, the
LinearSolveWrapper coarse solvers (through the oracle's callback protocol), the CommonSolve / precs shapes, the int32
guard of SparseMatrixCSC, and the split of the C/F splitting into patterns + sweep (amgs_rs_cf_splitting_patterns)."""
import ctypes as C
import os

import numpy as np
import pytest

import amg_amd as AMG
from amg_amd import sharded as SH
from conftest import uniform
from oracle import oracle as O


def test_row_cuts_and_sharded_level_count():
    assert SH.row_cuts(10, 3).tolist() == [0, 3, 6, 10]
    assert SH.row_cuts(7, 1).tolist() == [0, 7]
    sizes = [16777216, 8388608, 1398103, 228538, 38260, 5195, 800, 181, 51, 15]
    assert SH.num_sharded_levels(sizes, 8) == 4          # 16.7M / 8.4M / 1.4M / 229k rows sharded, the rest on rank 0
    assert SH.num_sharded_levels(sizes, 8, shard_min_rows=10 ** 9) == 0
    assert SH.num_sharded_levels([100, 40, 9], 2, shard_min_rows=10) == 2   # the coarsest level is never sharded
    assert SH.num_sharded_levels([100, 40, 9], 16, shard_min_rows=10) == 0  # fewer than 8 rows per rank


def test_level_export_roundtrip(tmp_path):
    A = AMG.poisson((12, 10, 8))
    ml = AMG.ruge_stuben(A, presmoother=AMG.Jacobi(0.6, iter=2), postsmoother=AMG.SOR(1.1, AMG.BackwardSweep()))
    levels = SH.level_arrays(ml, 2)
    SH.export_levels(levels, str(tmp_path / "lv"))
    back = SH.load_levels(str(tmp_path / "lv"))
    assert len(back) == 2
    for a, b in zip(levels, back):
        assert (a["n"], a["nc"], a["pre"], a["post"]) == (b["n"], b["nc"], b["pre"], b["post"])
        for key in ("A", "P", "R"):
            for x, y in zip(a[key], b[key]):
                assert np.array_equal(np.asarray(x), np.asarray(y))
        assert (a["S"] is None) == (b["S"] is None)
    # a rank's slice: rows [r0, r1) with global column indices, rowptr starting at 0
    rp, ci, va = SH._rows(back[0]["A"], 100, 300)
    Arp, Aci, Ava = A.csr_arrays()
    assert rp[0] == 0 and rp[-1] == Arp[300] - Arp[100]
    assert np.array_equal(ci, Aci[Arp[100]:Arp[300]]) and np.array_equal(va, Ava[Arp[100]:Arp[300]])


def test_linear_solve_wrapper_through_the_oracle_callback():
    A = AMG.poisson((14, 14, 14))
    b = uniform(A.m, 4)
    ref = O.OracleHierarchy(AMG.ruge_stuben(A, max_levels=2)).solve(b, reltol=1e-10)[0]
    for alg in (AMG.SuperLUFactorization(), AMG.DenseLUFactorization()):
        ml = AMG.ruge_stuben(A, max_levels=2, coarse_solver=AMG.LinearSolveWrapper(alg))
        assert not ml.coarse_solver.uses_dense() and repr(ml.coarse_solver) == repr(alg)
        x = O.OracleHierarchy(ml).solve(b, reltol=1e-10)[0]
        assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
        B = np.column_stack([b, 2.0 * b])
        X = ml.coarse_solver.host_solve(np.column_stack([np.ones(ml.final_A.m), np.arange(ml.final_A.m, dtype=float)]))
        assert X.shape == (ml.final_A.m, 2) and B.shape[1] == 2
    with pytest.raises(AMG.AMGError):
        AMG.LinearSolveWrapper(object())


def test_commonsolve_and_precs_shapes_without_a_gpu():
    A = AMG.poisson((20, 20))
    solt = AMG.init(AMG.SmoothedAggregationAMG(), A, np.ones(A.m))
    assert isinstance(solt, AMG.AMGSolver) and solt.ml.method == "sa"
    with pytest.raises(AMG.AMGError):
        AMG.init(object(), A, np.ones(A.m))
    with pytest.raises(AMG.AMGError):
        AMG.solve_(object())
    bld = AMG.RugeStubenPreconBuilder(blocksize=1, max_levels=3)
    assert bld.blocksize == 1 and bld.kwargs == {"max_levels": 3}
    I = AMG.Identity()
    v = np.arange(4.0)
    assert np.array_equal(I.ldiv(v), v) and I.ldiv(v) is not v


def test_int32_guard():
    with pytest.raises(AMG.AMGError):
        AMG.SparseMatrixCSC.from_arrays(2 ** 31, 1, np.zeros(2, dtype=np.int64), np.zeros(0), np.zeros(0))
    with pytest.raises(AMG.AMGError):
        AMG.SparseMatrixCSC.from_arrays(2, 2, np.array([0, 1, 2 ** 31 + 5], dtype=np.int64), np.zeros(0), np.zeros(0))


@pytest.mark.parametrize("dims", [(30,), (11, 9), (6, 7, 5)])
def test_splitting_on_patterns_equals_rs_splitting(dims):
    """amgs_rs_cf_splitting_patterns (what the GPU setup path calls with device-built patterns) on S without its
    diagonal and its transpose = amgs_rs_splitting on S."""
    L = AMG.setup_lib()
    A = AMG.poisson(dims)
    S, T = AMG.Classical(0.25)(A)
    n = A.m
    want = np.zeros(n, dtype=np.int32)
    S2 = AMG.SparseMatrixCSC.from_arrays(S.m, S.n, S.colptr.copy(), S.rowval.copy(), S.nzval.copy())
    assert L.amgs_rs_splitting(S2._h, want.ctypes.data) == 0
    Ssp = S.to_scipy().tolil()
    Ssp.setdiag(0)
    Sn = Ssp.tocsc()
    Sn.eliminate_zeros()
    Sn.sort_indices()
    Tn = Sn.T.tocsc()
    Tn.sort_indices()
    got = np.zeros(n, dtype=np.int32)
    sp_, sj, tp_, tj = (np.ascontiguousarray(a, dtype=np.int32) for a in (Sn.indptr, Sn.indices, Tn.indptr, Tn.indices))
    assert L.amgs_rs_cf_splitting_patterns(n, sp_.ctypes.data, sj.ctypes.data, tp_.ctypes.data, tj.ctypes.data,
                                           got.ctypes.data) == 0
    assert np.array_equal(got, want)


def test_qrsolver_rank_deficient_is_the_basic_solution():
    """coarse_solver.jl:66-81: `qr(A) \\ b` of a sparse A is SuiteSparseQR — on a rank-deficient coarse matrix the BASIC solution
    (dead columns zero), not pinv's minimum-norm one.  Checked: consistent right-hand sides are solved, dead columns stay zero,
    the difference to the minimum-norm solution lies in the null space, full-rank input is the inverse."""
    import sys
    H = sys.modules[AMG.QRSolver.__module__]
    n = 9
    M = 2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
    M[0, 0] = M[-1, -1] = 1.0                                   # 1-D Neumann Laplacian: null space = constants
    X = AMG.QRSolver(AMG.SparseMatrixCSC.from_dense(M)).dense_operator()
    assert np.array_equal(X, H._qr_basic_operator(M))
    rng = np.random.default_rng(5)
    b = rng.random(n)
    b -= b.mean()                                               # consistent
    x = X @ b
    assert np.linalg.norm(M @ x - b) <= 1e-13 * np.linalg.norm(b)
    assert x[-1] == 0.0 and np.all(X[-1] == 0.0)                # the last column depends on the others: dead
    d = x - np.linalg.lstsq(M, b, rcond=None)[0]
    assert np.linalg.norm(M @ d) <= 1e-13 and np.ptp(d) <= 1e-13 and abs(d[0]) > 1e-3   # a (non-zero) constant apart
    # two dependent columns in the middle / at the end
    A = rng.random((6, 6))
    A[:, 3] = 2 * A[:, 1]
    A[:, 5] = A[:, 0] - A[:, 2]
    X3 = H._qr_basic_operator(A)
    bb = A @ rng.random(6)
    x3 = X3 @ bb
    assert np.linalg.norm(A @ x3 - bb) <= 1e-13 * np.linalg.norm(bb) and x3[3] == 0.0 and x3[5] == 0.0
    # full rank: the inverse
    F = M + 0.1 * np.eye(n)
    assert np.abs(H._qr_basic_operator(F) - np.linalg.inv(F)).max() <= 1e-13
