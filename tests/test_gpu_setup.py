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
