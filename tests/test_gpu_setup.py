"""The GPU half of the setup phase (amgh_setup_*: strength.jl:7-37, classical.jl:57-189, R*A*P) against the host
library libamgsetup, which is pinned by the reference's setup goldens (tests/test_setup_goldens.py): every level's
A, P, R must come out with the same structure and the same values, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import amg_amd as AMG
from conftest import load_csc, uniform

pytestmark = pytest.mark.gpu


def same(X, Y):
    return (X.shape == Y.shape and np.array_equal(X.colptr, Y.colptr) and np.array_equal(X.rowval, Y.rowval)
            and np.array_equal(X.nzval, Y.nzval))


def check_hierarchy(A, **kw):
    h = AMG.ruge_stuben(A, setup="host", **kw)
    g = AMG.ruge_stuben(A, setup="gpu", **kw)
    assert len(h) == len(g)
    for l, (a, b) in enumerate(zip(h.levels, g.levels)):
        assert same(a.A, b.A), f"A differs on level {l}"
        assert same(a.P, b.P), f"P differs on level {l}"
        assert same(a.R, b.R), f"R differs on level {l}"
    assert same(h.final_A, g.final_A)
    return g


@pytest.mark.parametrize("dims", [(1000,), (50, 50), (24, 24, 24), (7,), (40, 3, 17)])
def test_gpu_setup_builds_the_host_hierarchy_bit_for_bit_on_poisson(dims):
    g = check_hierarchy(AMG.poisson(dims))
    if dims == (1000,):   # the reference's README / runtests.jl:77-88 hierarchy
        assert [l.A.m for l in g.levels] + [g.final_A.m] == [1000, 500, 250, 125, 62, 31, 15, 7]


@pytest.mark.parametrize("name", ["randlap", "test"])
def test_gpu_setup_on_the_references_irregular_graphs(name):
    check_hierarchy(load_csc(name))


def test_gpu_setup_other_theta_and_nonsymmetric_input():
    import scipy.sparse as sp
    A = AMG.poisson((30, 30))
    check_hierarchy(A, strength=AMG.Classical(0.5), max_levels=4)
    # a non-symmetric M-matrix: convection-diffusion-like, both symmetry conventions
    n = 400
    rng = np.random.default_rng(3)
    M = sp.diags([-1.0 - 0.3 * rng.random(n - 1), 4.0 + rng.random(n), -1.0 + 0.4 * rng.random(n - 1),
                  -0.5 * np.ones(n - 20)], [-1, 0, 1, 20], format="csc")
    check_hierarchy(AMG.SparseMatrixCSC.from_scipy(M))
    check_hierarchy(AMG.SparseMatrixCSC.from_scipy(M), symmetry=AMG.NoSymmetry())


def test_gpu_setup_primitives_transpose_and_spgemm():
    lib = AMG.hip_lib()
    from amg_amd.hierarchy import _DMat
    A = AMG.poisson((17, 13, 11))
    ml = AMG.ruge_stuben(A, max_levels=2)
    R, P = ml.levels[0].R, ml.levels[0].P
    dA, dR, dP = (_DMat.upload(M, lib) for M in (A, R, P))
    t = C.c_void_p()
    assert lib.amgh_setup_transpose(dR.h, C.byref(t)) == 0
    assert same(_DMat(t.value, lib).to_host(), R.transpose())
    c = C.c_void_p()
    assert lib.amgh_setup_spgemm(dR.h, dA.h, C.byref(c)) == 0
    RA = _DMat(c.value, lib)
    assert same(RA.to_host(), R @ A)
    c2 = C.c_void_p()
    assert lib.amgh_setup_spgemm(RA.h, dP.h, C.byref(c2)) == 0
    assert same(_DMat(c2.value, lib).to_host(), (R @ A) @ P)


def test_gpu_setup_solves_like_the_host_setup():
    A = AMG.poisson((32, 32, 32))
    b = uniform(A.m, 2)
    xg, hg = AMG._solve(AMG.ruge_stuben(A, setup="gpu"), b, log=True)
    xh, hh = AMG._solve(AMG.ruge_stuben(A, setup="host"), b, log=True)
    assert np.array_equal(xg, xh) and np.array_equal(hg, hh)


@pytest.mark.parametrize("dims", [(40, 40, 40), (300, 200), (5000,)])
def test_hierarchy_built_beside_the_setup_equals_the_one_built_after_it(dims):
    """ruge_stuben(setup="gpu", device=0): every level's upload + smoother schedules run on a worker thread while the
    host does that level's C/F splitting (amgh_push_level_begin / _end).  Same hierarchy, same cycle bit for bit."""
    A = AMG.poisson(dims)
    after = AMG.ruge_stuben(A, setup="gpu")
    beside = AMG.ruge_stuben(A, setup="gpu", device=0)
    assert (0, 1) in beside._dev and (0, 1) not in after._dev           # the handle is there before anyone asks
    assert len(after) == len(beside)
    for a, b in zip(after.levels, beside.levels):
        assert same(a.A, b.A) and same(a.P, b.P) and same(a.R, b.R)
    d1, d2 = after.device(), beside.device()
    assert d2 is beside._dev[(0, 1)]
    assert d1.lib.amgh_num_levels(d1.h) == d2.lib.amgh_num_levels(d2.h) == len(after.levels)
    r = uniform(A.m, 5)
    for cycle in (AMG.V(), AMG.W()):
        assert np.array_equal(AMG.aspreconditioner(after, cycle).ldiv(r), AMG.aspreconditioner(beside, cycle).ldiv(r))
    x, hist = AMG._solve(beside, r, log=True, reltol=1e-10)
    assert np.linalg.norm(A.to_scipy() @ x - r) <= 1e-9 * np.linalg.norm(r)


def test_built_beside_the_setup_when_coarsening_stops_early_or_never_starts():
    # max_levels = 1: no level at all; tiny operator: below max_coarse from the start
    for A, kw in ((AMG.poisson(30), dict(max_levels=1)), (AMG.poisson(8), {}), (AMG.poisson((12, 12)), dict(max_levels=2))):
        ml = AMG.ruge_stuben(A, setup="gpu", device=0, **kw)
        ref = AMG.ruge_stuben(A, setup="host", **kw)
        assert len(ml) == len(ref)
        r = uniform(A.m, 2)
        assert np.allclose(AMG.aspreconditioner(ml).ldiv(r), AMG.aspreconditioner(ref).ldiv(r), rtol=1e-12, atol=1e-14)


def check_sa_hierarchy(A, **kw):
    h = AMG.smoothed_aggregation(A, setup="host", **kw)
    g = AMG.smoothed_aggregation(A, setup="gpu", **kw)
    assert len(h) == len(g)
    for l, (a, b) in enumerate(zip(h.levels, g.levels)):
        assert same(a.A, b.A), f"A differs on level {l}"
        assert same(a.P, b.P), f"P differs on level {l}"
        assert same(a.R, b.R), f"R differs on level {l}"
    assert same(h.final_A, g.final_A)
    return g


@pytest.mark.parametrize("dims", [(1000,), (60, 50), (24, 24, 24), (9,), (40, 3, 17)])
def test_gpu_smoothed_aggregation_builds_the_host_hierarchy_bit_for_bit(dims):
    """smoothed_aggregation(setup="gpu"): prolongation smoothing (row sums, scaling, SpGEMM, subtraction) and R*A*P on
    the device (amgh_setup_jacobi_prolongation, aggregation.jl:30-59,147) — same sums in the same order as libamgsetup."""
    g = check_sa_hierarchy(AMG.poisson(dims))
    if dims == (1000,):     # the host library's level sizes for this matrix (aggregates of three)
        assert [l.A.m for l in g.levels] + [g.final_A.m] == [1000, 334, 112, 38, 13, 5]


def test_gpu_smoothed_aggregation_with_candidates_and_other_options():
    from conftest import load_npz
    d = load_npz("lin_elastic_2d")       # block-dof elasticity with three rigid-body modes: the QR path of fit_candidates
    A = load_csc("lin_elastic_2d")
    g = check_sa_hierarchy(A, B=d["B"])
    x, hist = AMG._solve(g, d["b"], log=True, reltol=1e-10)
    assert len(hist) - 1 == 27           # nns_test.jl:213-226
    check_sa_hierarchy(AMG.poisson((30, 30)), strength=AMG.SymmetricStrength(0.1), max_levels=3)
    check_sa_hierarchy(AMG.poisson((30, 30)), smooth=AMG.JacobiProlongation(1.0), improve_candidates=AMG.GaussSeidel(iter=2))
    check_sa_hierarchy(AMG.poisson(500), B=np.linspace(1.0, 2.0, 500))
    check_sa_hierarchy(load_csc("randlap"))
    check_sa_hierarchy(AMG.poisson((20, 20)), symmetry=AMG.NoSymmetry())


@pytest.mark.parametrize("theta", [0.0, 0.1, 0.25, 0.9])
def test_gpu_symmetric_strength_is_the_host_librarys_bit_for_bit(theta):
    """amgh_setup_symmetric_strength (strength.jl:77-122): entries with a_ij^2 < theta^2 |a_ii| |a_jj| and stored zeros
    dropped, |.| scaled by the column maxima — structure and values of the host library's S, on grids, on an irregular
    graph, on a non-symmetric operator with stored zeros and a zero diagonal, and the bsr_flag shortcut (:81-84)."""
    from amg_amd.hierarchy import _DMat
    import scipy.sparse as sp
    lib = AMG.hip_lib()
    rng = np.random.default_rng(11)
    n = 400
    M = sp.random(n, n, density=0.02, random_state=3, format="lil")
    M.setdiag(rng.random(n) + 0.5)
    M[5, 5] = 0.0                                   # a stored zero on the diagonal
    M[7, 9] = 0.0                                   # a stored zero off the diagonal
    M = M.tocsc()
    M.data[::17] *= -1.0
    cases = [AMG.poisson((30, 20)), AMG.poisson((12, 11, 10)), load_csc("randlap"), AMG.SparseMatrixCSC.from_scipy(M)]
    for A in cases:
        dA = _DMat.upload(A, lib)
        for bsr in (0, 1):
            s_ = C.c_void_p()
            assert lib.amgh_setup_symmetric_strength(dA.h, theta, bsr, C.byref(s_)) == 0
            S_dev = _DMat(s_.value, lib).to_host()
            S_host, _ = AMG.SymmetricStrength(theta)(A, bool(bsr))
            assert same(S_dev, S_host), (A.shape, theta, bsr)


def test_gpu_fit_candidates_vector_is_the_host_librarys_bit_for_bit():
    """amgh_setup_fit_candidates_vector (aggregation.jl:161-193): T = AggOp' carrying the candidate restricted to every
    aggregate, normalised, and the aggregates' norms — structure and values of the host library's, with aggregates of
    one node, nodes in no aggregate, and a candidate that vanishes on a whole aggregate (norm 0: column of zeros)."""
    from amg_amd.hierarchy import _DMat, fit_candidates
    lib = AMG.hip_lib()
    for A, seed in ((AMG.poisson((40, 30)), 1), (AMG.poisson((14, 12, 10)), 2), (load_csc("randlap"), 3)):
        S, _ = AMG.SymmetricStrength(0.0)(A)
        AggOp = AMG.StandardAggregation()(S)
        n = A.m
        B = uniform(n, seed) + 0.5
        first = AggOp.transpose().rowval[: AggOp.transpose().colptr[1]]     # the fine nodes of aggregate 0
        B[first] = 0.0
        T_host, Bc_host = fit_candidates(AggOp, B)
        dAgg = _DMat.upload(AggOp, lib)
        t_ = C.c_void_p()
        Bc = np.empty(AggOp.m)
        assert lib.amgh_setup_fit_candidates_vector(dAgg.h, B.ctypes.data, 1e-10, C.byref(t_), Bc.ctypes.data) == 0
        T_dev = _DMat(t_.value, lib).to_host()
        assert same(T_dev, T_host) and np.array_equal(Bc, Bc_host) and Bc[0] == 0.0
