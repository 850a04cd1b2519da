"""The collapsed coarse tail (amghip.h: amgh_tail_dense_build; the library's default — the rest of the suite pins bits of the
per-level cycle and runs with tail_dense_rows = 0): from the first level of at most `tail_dense_rows` rows down, the recursion
of __solve! (multilevel.jl:214-239) is applied as ONE dense operator per cycle type, built from that very recursion on the
columns of the identity.  The same linear map: cycles / solves / cg within the suite's 1e-10 of the oracle and within 1e-12 of
the per-level cycle of the same handle (tunable tail_dense = 0), V / W / F, Gauss-Seidel / SOR / Jacobi tails, ruge_stuben and
smoothed_aggregation, blocks of right-hand sides, Float32, hierarchies that are a tail altogether (C1, C5)."""
import numpy as np
import pytest

import amg_amd as AMG
from conftest import load_csc, load_npz, uniform
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-10
TIGHT = 1e-12


def rel(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y)) / max(np.linalg.norm(y), 1e-300)


@pytest.fixture()
def tail():
    libs = [AMG.hip_lib(), AMG.hip_lib("float32")]
    for lib in libs:
        assert lib.amgh_debug_set_tunable(b"tail_dense_rows", 6144) == 0
        assert lib.amgh_debug_set_tunable(b"tail_dense", 1) == 0
    yield libs[0]
    for lib in libs:
        lib.amgh_debug_set_tunable(b"tail_dense_rows", 0)
        lib.amgh_debug_set_tunable(b"tail_dense", 1)


def _both(lib, fn):
    """fn() through the dense tail and through the per-level cycle of the same handle."""
    a = fn()
    lib.amgh_debug_set_tunable(b"tail_dense", 0)
    try:
        b = fn()
    finally:
        lib.amgh_debug_set_tunable(b"tail_dense", 1)
    return a, b


def test_tail_of_a_grid_hierarchy_v_w_f(tail):
    A = AMG.poisson((40, 40, 40))
    ml = AMG.ruge_stuben(A)
    sizes = [lv.A.m for lv in ml.levels] + [ml.final_A.m]
    dev = ml.device()
    lv, rows, ms = dev.tail_dense_info(0)
    assert lv >= 1 and rows == sizes[lv] <= 6144 and sizes[lv - 1] > 6144 and ms > 0.0   # built inside the setup, for V
    assert dev.tail_dense_info(1)[0] == -1                                                # W: at its first cycle
    oh = O.OracleHierarchy(ml)
    b = uniform(A.m, 3)
    z, zl = _both(tail, lambda: AMG.aspreconditioner(ml).ldiv(b))
    assert rel(z, oh.precond(b)) <= TOL and rel(z, zl) <= TIGHT and not np.array_equal(z, zl)   # (the operator really ran)
    assert np.array_equal(z, AMG.aspreconditioner(ml).ldiv(b))                                   # deterministic
    for cyc in (AMG.W(), AMG.F()):
        x, xl = _both(tail, lambda: AMG._solve(ml, b, cyc, maxiter=3, calculate_residual=False))
        xo, _, _ = oh.solve(b, cycle=cyc.code, maxiter=3, calculate_residual=False)
        assert rel(x, xo) <= TOL and rel(x, xl) <= TIGHT
        assert dev.tail_dense_info(cyc.code)[0] == lv
    x, hist = AMG._solve(ml, b, reltol=1e-9, log=True)
    xo, ho, _ = oh.solve(b, reltol=1e-9)
    assert len(hist) == len(ho) and rel(x, xo) <= TOL and np.abs(np.asarray(hist) - np.asarray(ho)).max() <= TOL * ho[0]


@pytest.mark.parametrize("kind", ["sor", "jacobi", "fwd_bwd"])
def test_tail_with_other_smoothers(tail, kind):
    A = AMG.poisson((48, 48, 16))
    pre, post = {"sor": (AMG.SOR(1.2), AMG.SOR(0.9, iter=2)), "jacobi": (AMG.Jacobi(2.0 / 3.0), AMG.Jacobi(0.6, iter=2)),
                 "fwd_bwd": (AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()))}[kind]
    ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=post)
    assert ml.device().tail_dense_info(0)[0] >= 1
    b = uniform(A.m, 11) - 0.5
    z, zl = _both(tail, lambda: AMG.aspreconditioner(ml).ldiv(b))
    assert rel(z, O.OracleHierarchy(ml).precond(b)) <= TOL and rel(z, zl) <= TIGHT


def test_tail_smoothed_aggregation_c2_shape(tail):
    A = AMG.poisson((192, 192))
    ml = AMG.smoothed_aggregation(A, presmoother=AMG.Jacobi(2.0 / 3.0), postsmoother=AMG.Jacobi(2.0 / 3.0))
    assert ml.device().tail_dense_info(0)[0] >= 1
    b = uniform(A.m, 2)
    z, zl = _both(tail, lambda: AMG.aspreconditioner(ml).ldiv(b))
    assert rel(z, O.OracleHierarchy(ml).precond(b)) <= TOL and rel(z, zl) <= TIGHT


def test_hierarchy_that_is_a_tail_altogether_c1_c5(tail):
    # C1: poisson(1000), ruge_stuben — the whole hierarchy is one operator; a solve iterates x += M (b - A x)
    A = AMG.poisson(1000)
    ml = AMG.ruge_stuben(A)
    assert ml.device().tail_dense_info(0)[:2] == (0, 1000)
    oh = O.OracleHierarchy(ml)
    b = uniform(1000, 0)
    x, hist = AMG._solve(ml, b, reltol=1e-8, log=True)
    xo, ho, _ = oh.solve(b, reltol=1e-8)
    assert len(hist) == len(ho) and rel(x, xo) <= TOL
    z, zl = _both(tail, lambda: AMG.aspreconditioner(ml).ldiv(b))
    assert rel(z, oh.precond(b)) <= TOL and rel(z, zl) <= TIGHT
    # C5: lin_elastic_2d, smoothed_aggregation with B, as preconditioner in cg: the reference's own counts (nns_test.jl:213-226)
    d = load_npz("lin_elastic_2d")
    A, b, B = load_csc("lin_elastic_2d"), d["b"], d["B"]
    ml = AMG.smoothed_aggregation(A, B=B)
    assert ml.device().tail_dense_info(0)[0] == 0
    oh = O.OracleHierarchy(ml)
    x, hist = AMG._solve(ml, b, reltol=1e-10, log=True)
    xo, ho, _ = oh.solve(b, reltol=1e-10)
    assert len(hist) - 1 == 27 == len(ho) - 1 and rel(x, xo) <= TOL
    xp, log = AMG.cg(A, b, Pl=AMG.aspreconditioner(ml), reltol=1e-10, log=True)
    xpo, _, itp = oh.pcg(b, reltol=1e-10)
    assert log["iters"] == itp == 13 and rel(xp, xpo) <= 1e-9
    for cyc in (AMG.W(), AMG.F()):
        z = AMG._solve(ml, b, cyc, maxiter=2, calculate_residual=False)
        zo, _, _ = oh.solve(b, cycle=cyc.code, maxiter=2, calculate_residual=False)
        assert rel(z, zo) <= TOL


def test_tail_blocks_of_right_hand_sides(tail):
    A = AMG.poisson((36, 36, 36))
    ml = AMG.ruge_stuben(A)
    oh = O.OracleHierarchy(ml)
    for bs in (3, 8):
        Bm = np.stack([uniform(A.m, 20 + c) - 0.3 * c for c in range(bs)], axis=1)
        Z = AMG.aspreconditioner(ml).ldiv(Bm)
        for c in range(bs):
            assert rel(Z[:, c], oh.precond(Bm[:, c])) <= TOL
    z1 = AMG.aspreconditioner(ml).ldiv(np.ascontiguousarray(Bm[:, 0]))
    assert rel(Z[:, 0], z1) <= TIGHT


def test_tail_float32(tail):
    A = AMG.poisson((32, 32, 32))
    A32 = AMG.SparseMatrixCSC.from_scipy(A.to_scipy().astype(np.float32))
    ml = AMG.ruge_stuben(A32)
    b = uniform(A.m, 4).astype(np.float32)
    z = AMG.aspreconditioner(ml).ldiv(b)
    zo = O.OracleHierarchy(ml, dtype=np.float32).precond(b)
    assert z.dtype == np.float32 and rel(z.astype(np.float64), zo.astype(np.float64)) <= 5e-5


def test_tail_off_and_host_coarse_solver(tail):
    A = AMG.poisson((24, 24, 24))
    b = uniform(A.m, 9)
    # a host coarse solver (the reference's `(cs)(x, b)` protocol): the tail stays a recursion
    ml = AMG.ruge_stuben(A, coarse_solver=AMG.LinearSolveWrapper(AMG.SuperLUFactorization()))
    assert ml.device().tail_dense_info(0)[0] == -1
    assert rel(AMG.aspreconditioner(ml).ldiv(b), O.OracleHierarchy(ml).precond(b)) <= TOL
    # tail_dense_rows = 0 at amgh_finalize: nothing is built, the bits are the per-level cycle's
    tail.amgh_debug_set_tunable(b"tail_dense_rows", 0)
    ml0 = AMG.ruge_stuben(A)
    assert ml0.device().tail_dense_info(0)[0] == -1
    z0 = AMG.aspreconditioner(ml0).ldiv(b)
    tail.amgh_debug_set_tunable(b"tail_dense_rows", 6144)
    ml1 = AMG.ruge_stuben(A)
    assert ml1.device().tail_dense_info(0)[0] >= 0
    tail.amgh_debug_set_tunable(b"tail_dense", 0)
    z1 = AMG.aspreconditioner(ml1).ldiv(b)
    tail.amgh_debug_set_tunable(b"tail_dense", 1)
    assert np.array_equal(z0, z1)
