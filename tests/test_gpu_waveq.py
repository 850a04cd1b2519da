"""The single-wave walk of small operators with FOUR lanes per row (gs_waveq_kernel, the library's default where rows have
at least 8 off-diagonal entries; the rest of the suite runs with gs_wave_quad = 0, the one-lane walk that reproduces the scalar
loop bit for bit).  Exact Gauss-Seidel / SOR in the reference's row order (smoother.jl:61-90, 193-221); a row's additions are
four interleaved partial sums: sweeps within 1e-13 of the oracle's scalar loop, cycles / cg within the suite's 1e-10, the
reference's own counts on lin_elastic_2d (nns_test.jl:213-226: 27 cycles, 13 cg iterations)."""
import numpy as np
import pytest

import amg_amd as AMG
from amg_amd.device import smooth_standalone
from conftest import load_csc, load_npz, uniform
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-10
TIGHT = 1e-13


def rel(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y)) / max(np.linalg.norm(y), 1e-300)


@pytest.fixture()
def quad():
    libs = [AMG.hip_lib(), AMG.hip_lib("float32")]
    for lib in libs:
        assert lib.amgh_debug_set_tunable(b"gs_wave_quad", 1) == 0
    yield libs[0]
    for lib in libs:
        lib.amgh_debug_set_tunable(b"gs_wave_quad", 0)


def _elastic():
    d = load_npz("lin_elastic_2d")
    return load_csc("lin_elastic_2d"), d["b"], d["B"]


@pytest.mark.parametrize("sm", ["fwd", "bwd", "sym", "sor", "sor_sym2"])
def test_quad_walk_sweeps_equal_the_scalar_loop(quad, sm):
    A, b, _ = _elastic()
    n = A.m
    smoother = {"fwd": AMG.GaussSeidel(AMG.ForwardSweep()), "bwd": AMG.GaussSeidel(AMG.BackwardSweep()), "sym": AMG.GaussSeidel(),
                "sor": AMG.SOR(1.3, sweep=AMG.ForwardSweep()), "sor_sym2": AMG.SOR(0.8, iter=2)}[sm]
    x0 = uniform(n, 7) - 0.5

    def gpu():
        x = x0.copy()
        smooth_standalone(smoother, A, x, b)
        return x

    x = gpu()
    xo = O.smooth(smoother, A, x0, b)
    assert rel(x, xo) <= TIGHT
    assert np.array_equal(x, gpu())   # deterministic
    # the four-lane walk really ran: the one-lane walk (same operator, laid out again) gives the scalar loop's bits, this one
    # differs from it somewhere — in rounding only
    quad.amgh_debug_set_tunable(b"gs_wave_quad", 0)
    x1 = gpu()
    quad.amgh_debug_set_tunable(b"gs_wave_quad", 1)
    assert np.array_equal(x1, xo)
    assert not np.array_equal(x, x1) and rel(x, x1) <= TIGHT


def test_quad_walk_c5_cycles_and_cg(quad):
    A, b, B = _elastic()
    ml = AMG.smoothed_aggregation(A, B=B)
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
    Bm = np.stack([b, uniform(A.m, 5), np.cos(np.arange(A.m))], axis=1)      # blocks of right-hand sides: a workgroup per column
    Z = AMG.aspreconditioner(ml).ldiv(Bm)
    for c in range(3):
        assert rel(Z[:, c], oh.precond(Bm[:, c])) <= TOL


def test_quad_walk_on_the_small_levels_of_a_grid_hierarchy(quad):
    A = AMG.poisson((20, 20, 20))
    ml = AMG.ruge_stuben(A)
    b = uniform(A.m, 3)
    oh = O.OracleHierarchy(ml)
    assert rel(AMG.aspreconditioner(ml).ldiv(b), oh.precond(b)) <= TOL
    ml = AMG.ruge_stuben(A, presmoother=AMG.SOR(1.2), postsmoother=AMG.SOR(1.2))
    assert rel(AMG.aspreconditioner(ml).ldiv(b), O.OracleHierarchy(ml).precond(b)) <= TOL


def test_quad_walk_float32(quad):
    A, b, B = _elastic()
    S = A.to_scipy().astype(np.float32)
    A32 = AMG.SparseMatrixCSC.from_scipy(S)
    ml = AMG.smoothed_aggregation(A32, B=B.astype(np.float32))
    z = AMG.aspreconditioner(ml).ldiv(b.astype(np.float32))
    zo = O.OracleHierarchy(ml, dtype=np.float32).precond(b.astype(np.float32))
    assert z.dtype == np.float32 and rel(z.astype(np.float64), zo.astype(np.float64)) <= 5e-5


def test_quad_walk_zero_diagonals_and_rows_that_must_divide(quad):
    """Rows with a zero diagonal keep their value (smoother.jl:87) and rows whose diagonal lies outside the range the record's
    reciprocal is trusted in take the division itself (the wave-uniform cold branch), with four lanes per row as with one."""
    import scipy.sparse as sp
    A0, b, _ = _elastic()
    M = A0.to_scipy().tolil()
    n = M.shape[0]
    for r in (0, 77, n - 1):
        M[r, r] = 0.0
    for r in (5, 123):
        M[r, r] = 6.0e120
    M[9, 9] = 3.0e-130
    A = AMG.SparseMatrixCSC.from_scipy(sp.csc_matrix(M.tocsr()))
    x0 = uniform(n, 61) - 0.5
    for sm in (AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(iter=2), AMG.SOR(1.2)):
        x = x0.copy()
        smooth_standalone(sm, A, x, b)
        xo = O.smooth(sm, A, x0, b)
        ok = np.isfinite(xo)
        assert np.array_equal(np.isfinite(x), ok), repr(sm)
        floor = 1e-3 * np.median(np.abs(xo[ok]))
        assert np.max(np.abs(x[ok] - xo[ok]) / np.maximum(np.abs(xo[ok]), floor)) <= 1e-12, repr(sm)
        for r in (0, 77, n - 1):
            assert x[r] == x0[r]
