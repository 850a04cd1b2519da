"""Pins the CPU oracle (oracle/amg_oracle.c) against every known-answer vector the
reference's own tests hold for the solve phase.  No GPU needed."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import amg_amd as AMG
from conftest import load_csc, load_npz, uniform
from oracle import oracle as O

def approx(x, y, rtol=np.sqrt(np.finfo(float).eps)):
    """Julia's isapprox for vectors: norm(x-y) <= rtol*max(norm(x), norm(y))."""
    x, y = np.asarray(x), np.asarray(y)
    return np.linalg.norm(x - y) <= rtol * max(np.linalg.norm(x), np.linalg.norm(y))


FWD = AMG.GaussSeidel(AMG.ForwardSweep())
BWD = AMG.GaussSeidel(AMG.BackwardSweep())


def tridiag(N):
    return AMG.SparseMatrixCSC.from_scipy(sp.diags([-np.ones(N - 1), 2 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1]))


def test_gauss_seidel_hand_values():  # sa_tests.jl:316-379
    A1, A3 = tridiag(1), tridiag(3)
    assert np.allclose(O.smooth(FWD, A1, [0.0], [0.0]), [0.0])
    assert np.array_equal(O.smooth(FWD, A3, [0, 1, 2.0], np.zeros(3)), [1 / 2, 5 / 4, 5 / 8])
    assert np.allclose(O.smooth(BWD, A1, [0.0], [0.0]), [0.0])
    assert np.array_equal(O.smooth(BWD, A3, [0, 1, 2.0], np.zeros(3)), [1 / 8, 1 / 4, 1 / 2])
    assert np.array_equal(O.smooth(FWD, A1, [0.0], [10.0]), [5.0])
    assert np.array_equal(O.smooth(FWD, A3, [0, 1, 2.0], [10, 20, 30.0]), [11 / 2, 55 / 4, 175 / 8])
    A = tridiag(100)
    x1 = O.smooth(AMG.GaussSeidel(AMG.ForwardSweep(), 200), A, np.ones(100), np.zeros(100))
    x2 = O.smooth(AMG.GaussSeidel(AMG.BackwardSweep(), 200), A, np.ones(100), np.zeros(100))
    r1, r2 = np.linalg.norm(A @ x1), np.linalg.norm(A @ x2)
    assert r1 < 0.01 and r2 < 0.01 and np.isclose(r1, r2)


def test_issue26_symmetric_gs_iter4():  # test_regression.jl:14-23
    x = O.smooth(AMG.GaussSeidel(AMG.SymmetricSweep(), 4), AMG.poisson(10), np.ones(10), np.zeros(10))
    ref = [0.176765, 0.353529, 0.497517, 0.598914, 0.653311, 0.659104, 0.615597, 0.52275, 0.382787, 0.203251]
    assert ((x - ref) ** 2).sum() < 1e-6


def test_fast_and_nosymmetry_smoothers_agree():  # test_smoothers.jl:29-45
    A = AMG.poisson(50)
    x0, b = uniform(50, 3), np.ones(50)
    for s in (AMG.Jacobi(4 / 5, iter=2), AMG.GaussSeidel(AMG.SymmetricSweep(), iter=2), AMG.SOR(0.5, iter=2)):
        xf = O.smooth(s, A, x0, b, hermitian=True)
        xg = O.smooth(s, A, x0, b, hermitian=False)
        assert np.allclose(xf, xg, rtol=1e-12, atol=0)


def test_nosymmetry_smoothers_converge():  # test_smoothers.jl:15-27 (own RNG)
    rng = np.random.default_rng(1)
    N = 50
    M = sp.random(N, N, 0.05, random_state=rng, format="csc") + 5 * sp.identity(N, format="csc")
    A = AMG.SparseMatrixCSC.from_scipy(M)
    x0, b = rng.random(N), np.ones(N)
    for s in (AMG.Jacobi(1 / 6, iter=500), AMG.GaussSeidel(AMG.ForwardSweep(), 100),
              AMG.GaussSeidel(AMG.BackwardSweep(), 100), AMG.GaussSeidel(AMG.SymmetricSweep(), 100),
              AMG.SOR(0.5, AMG.ForwardSweep(), 100), AMG.SOR(0.5, AMG.BackwardSweep(), 100),
              AMG.SOR(0.5, AMG.SymmetricSweep(), 100)):
        x = O.smooth(s, A, x0, b, hermitian=False)
        assert np.allclose(M @ x, b)


def test_nosymmetry_singular_exception():  # smoother.jl:239-241
    M = sp.csc_matrix(np.array([[1.0, 2.0], [3.0, 0.0]]))
    with pytest.raises(ArithmeticError):
        O.smooth(FWD, AMG.SparseMatrixCSC.from_scipy(M), np.ones(2), np.ones(2), hermitian=False)


def test_solver_poisson1000():  # runtests.jl:115-124 (config C1)
    A = AMG.poisson(1000)
    b = A @ np.ones(1000)
    x, hist, it = O.OracleHierarchy(AMG.ruge_stuben(A)).solve(b)
    assert ((x - 1) ** 2).sum() < 1e-8 and it == 6
    x, _, _ = O.OracleHierarchy(AMG.ruge_stuben(A, presmoother=FWD, postsmoother=FWD)).solve(b)
    assert ((x - 1) ** 2).sum() < 1e-8


def test_solver_randlap():  # runtests.jl:130-139
    A = load_csc("randlap")
    b = A @ np.ones(100)
    x, _, _ = O.OracleHierarchy(AMG.ruge_stuben(A, presmoother=FWD, postsmoother=FWD)).solve(b)
    assert (x ** 2).sum() < 1e-8
    x, _, _ = O.OracleHierarchy(AMG.ruge_stuben(A)).solve(b)
    assert (x ** 2).sum() < 1e-6


def test_thing_known_answer_vectors():  # runtests.jl:143-224
    A = load_csc("thing")
    g = load_npz("thing_solutions")
    n = 46
    b = np.zeros(n); b[0], b[1] = 1, -1
    oh = O.OracleHierarchy(AMG.ruge_stuben(A, presmoother=FWD, postsmoother=FWD, coarse_solver=AMG.Pinv))
    x, _, _ = oh.solve(A @ np.ones(n), maxiter=1, abstol=1e-12)
    assert ((x - g["solve_Aones_fwd_maxiter1"]) ** 2).sum() < 1e-8
    x, _, _ = oh.solve(b, maxiter=1, abstol=1e-12)
    assert ((x - g["solve_b_fwd_maxiter1"]) ** 2).sum() < 1e-8
    assert np.abs(x - g["solve_b_fwd_maxiter1"]).max() < 1e-7   # print precision of the golden
    x, _, _ = oh.pcg(b)
    assert ((x - g["cg_fwd"]) ** 2).sum() < 1e-8
    oh = O.OracleHierarchy(AMG.ruge_stuben(A, coarse_solver=AMG.Pinv))
    x, _, _ = oh.pcg(b, maxiter=100000, reltol=1e-6)
    assert ((x - g["cg_sym_reltol1e-6"]) ** 2).sum() < 1e-8
    x, _, _ = oh.solve(b, maxiter=1, reltol=1e-12)
    assert ((x - g["solve_b_sym_maxiter1"]) ** 2).sum() < 1e-8
    assert np.abs(x - g["solve_b_sym_maxiter1"]).max() < 5e-6


def test_cycles_poisson50x50():  # cycle_tests.jl:6-30
    A = AMG.poisson((50, 50))
    b = A @ np.ones(A.m)
    nb = np.linalg.norm(b)
    expect = {"ruge_stuben": 7, "smoothed_aggregation": 10}  # V-cycle counts
    for f in (AMG.ruge_stuben, AMG.smoothed_aggregation):
        oh = O.OracleHierarchy(f(A))
        for cyc in range(3):
            x, hist, it = oh.solve(b, cycle=cyc, reltol=1e-8)
            assert np.linalg.norm(b - A @ x) < 1e-8 * nb
            if cyc == 0:
                assert it == expect[f.__name__]
            x, _, _ = oh.pcg(b, cycle=cyc, reltol=1e-8)
            assert np.linalg.norm(b - A @ x) <= 1e-8 * nb


def test_issue46_bug_matrix():  # test_regression.jl:25-39
    a = load_csc("bug")
    b = np.zeros(4); b[0], b[1] = 1, -1
    for f in (AMG.smoothed_aggregation, AMG.ruge_stuben):
        oh = O.OracleHierarchy(f(a))
        x, _, _ = oh.solve(b)
        assert ((a @ x - b) ** 2).sum() < 1e-10
        x, _, _ = oh.pcg(b, maxiter=1000)
        assert ((a @ x - b) ** 2).sum() < 1e-10


def test_issue56_tight_tolerance():  # test_regression.jl:59-69 (b from own stream)
    X = AMG.SparseMatrixCSC.from_scipy(AMG.poisson(27000).to_scipy() + 24.0 * sp.identity(27000, format="csc"))
    b = uniform(27000, 56)
    ref = spla.spsolve(X.to_scipy(), b)
    x, _, _ = O.OracleHierarchy(AMG.ruge_stuben(X)).solve(b, reltol=1e-10)
    assert approx(x, ref, 1e-10)
    ml = AMG.smoothed_aggregation(X, strength=AMG.SymmetricStrength(0.05))
    x, _, _ = O.OracleHierarchy(ml).solve(b, reltol=1e-10)
    assert len(ml) == 1   # every node isolated at theta=0.05: the whole solve is the coarse solver
    assert approx(x, ref, 1e-10)


def test_issue95_nonsymmetric_nosymmetry():  # test_regression.jl:71-83 (own RNG)
    rng = np.random.default_rng(95)
    N = 10000
    M = sp.random(N, N, 0.001, random_state=rng, format="csc") + 5 * sp.identity(N, format="csc")
    b = np.ones(N)
    for f in (AMG.ruge_stuben, AMG.smoothed_aggregation):
        ml = f(M, symmetry=AMG.NoSymmetry())
        x, _, _ = O.OracleHierarchy(ml).solve(b)
        assert approx(M @ x, b, 1e-8)


def test_lin_elastic_2d():  # nns_test.jl:213-226 (config C5)
    d = load_npz("lin_elastic_2d")
    A = load_csc("lin_elastic_2d")
    oh = O.OracleHierarchy(AMG.smoothed_aggregation(A, B=d["B"]))
    x, hist, it = oh.solve(d["b"], reltol=1e-10)
    assert approx(A @ x, d["b"]) and it == 27
    x, _, itp = oh.pcg(d["b"], reltol=1e-10)
    assert approx(A @ x, d["b"]) and itp == 13
    x, hist, it = O.OracleHierarchy(AMG.smoothed_aggregation(A, coarse_solver=AMG.Pinv)).solve(d["b"], reltol=1e-10)
    assert not approx(A @ x, d["b"]) and hist[0] > hist[-1]


def test_near_null_space_argument_forms():  # nns_test.jl:6-24
    A = AMG.poisson(100)
    b = uniform(100, 7)
    xs = []
    for B in (None, np.ones(100), np.ones((100, 1))):
        x, _, _ = O.OracleHierarchy(AMG.smoothed_aggregation(A, B=B)).solve(b, maxiter=1, abstol=1e-6)
        xs.append(x)
    assert np.allclose(xs[0], xs[1]) and np.allclose(xs[0], xs[2])


def test_b_zero_and_log_semantics():  # multilevel.jl:170-178 / SURVEY §9
    A = AMG.poisson(100)
    oh = O.OracleHierarchy(AMG.ruge_stuben(A))
    x, hist, it = oh.solve(np.zeros(100))
    assert it == 0 and np.all(x == 0) and hist.tolist() == [0.0]
    b = A @ np.ones(100)
    x, hist, it = oh.solve(b, maxiter=3, calculate_residual=False)
    assert it == 3 and hist[0] == np.linalg.norm(b)
