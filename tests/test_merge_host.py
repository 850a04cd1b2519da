"""Merged dependency levels, host side (no GPU): the construction libamghip uses for its Gauss-Seidel schedules —
level order, groups of m levels made independent by substitution over one triangle, pre-pass over the other — is
run through `amgh_debug_merged_sweep_host` and compared with the oracle's scalar lexicographic sweep
(smoother.jl:78-88).  The device kernels then only have to apply the same composite rows (tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import scipy.sparse as sp

import amg_amd as AMG
from conftest import uniform
from oracle import oracle as O

FWD, BWD = AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep())


def _host_sweep(rowptr, col, val, nrows, ncols, m, backward, x, b, omega=1.0):
    lib = AMG.hip_lib()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    x = np.array(x, dtype=np.float64, copy=True)
    b = np.ascontiguousarray(b, dtype=np.float64)
    rc = lib.amgh_debug_merged_sweep_host(nrows, ncols, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, m,
                                          int(backward), float(omega), x.ctypes.data, b.ctypes.data)
    assert rc >= 0, rc
    return x, rc


def _rel(a, r):
    return np.linalg.norm(a - r) / np.linalg.norm(r)


def _irregular(n, half_bw, seed, zero_diag=()):
    rng = np.random.default_rng(seed)
    L = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=rng, format="lil")
    for i in range(n):
        for j in range(max(0, i - half_bw), i):
            if rng.random() < 0.5:
                L[i, j] = -rng.random()
    L = sp.tril(L.tocsr(), k=-1)
    M = L + L.T
    d = np.asarray(abs(M).sum(axis=1)).ravel() + 1.0
    for r in zero_diag:
        d[r] = 0.0
    M = (M + sp.diags(d)).tocsc()
    M.eliminate_zeros()
    return AMG.SparseMatrixCSC.from_scipy(M)


@pytest.mark.parametrize("m", [1, 2, 3, 5, 16])
def test_merged_sweep_equals_scalar_sweep(m):
    cases = [AMG.poisson((12, 10, 9)), AMG.ruge_stuben(AMG.poisson((20, 18, 16))).levels[1].A,
             _irregular(700, 9, 3, zero_diag=(0, 123, 699)), AMG.poisson(200)]
    for k, A in enumerate(cases):
        n = A.m
        rp, ci, va = A.csr_arrays()
        x0, b = uniform(n, 10 + k) - 0.5, uniform(n, 20 + k)
        for back, s in ((0, FWD), (1, BWD)):
            x, ngroups = _host_sweep(rp, ci, va, n, n, m, back, x0, b)
            ref = O.smooth(s, A, x0, b)
            assert _rel(x, ref) <= 1e-13, (m, k, back, _rel(x, ref))
            assert ngroups >= 1


@pytest.mark.parametrize("omega", [0.5, 1.2, 1.9])
def test_merged_sor_sweep_equals_scalar_sor(omega):
    """SOR = the same triangular solve with the diagonal D/omega and ((1-omega)/omega) D x_old on the right-hand side
    (smoother.jl:193-221): merged groups of that scaled system against the oracle's scalar SOR sweeps."""
    cases = [AMG.poisson((12, 10, 9)), _irregular(500, 8, 7, zero_diag=(3, 250)),
             AMG.ruge_stuben(AMG.poisson((20, 18, 16))).levels[1].A]
    for k, A in enumerate(cases):
        n = A.m
        rp, ci, va = A.csr_arrays()
        x0, b = uniform(n, 40 + k) - 0.5, uniform(n, 50 + k)
        for m in (1, 3, 8):
            for back, s in ((0, AMG.SOR(omega, AMG.ForwardSweep())), (1, AMG.SOR(omega, AMG.BackwardSweep()))):
                x, _ = _host_sweep(rp, ci, va, n, n, m, back, x0, b, omega)
                ref = O.smooth(s, A, x0, b)
                assert _rel(x, ref) <= 1e-12, (omega, k, m, back, _rel(x, ref))


def test_merged_sweep_with_halo_columns():
    """Local block of a row-sharded operator: columns >= nrows are frozen; they belong to the pre-pass."""
    A = AMG.poisson((10, 10, 12))
    M = A.to_scipy().tocsr()
    n, nloc = M.shape[0], 10 * 10 * 8
    top = M[:nloc, :].tocsr()
    ext = sp.vstack([top, sp.hstack([sp.csr_matrix((n - nloc, nloc)), sp.identity(n - nloc)])])
    ext_T = AMG.SparseMatrixCSC.from_scipy(ext.T.tocsc())   # the oracle's fast smoothers read column i as row i
    x0, b = uniform(n, 31) - 0.5, uniform(n, 32)
    b_ext = b.copy(); b_ext[nloc:] = x0[nloc:]
    for m in (2, 4):
        for back, s in ((0, FWD), (1, BWD)):
            x, _ = _host_sweep(top.indptr, top.indices, top.data, nloc, n, m, back, x0, b[:nloc])
            ref = O.smooth(s, ext_T, x0, b_ext)
            assert _rel(x[:nloc], ref[:nloc]) <= 1e-13 and np.array_equal(x[nloc:], x0[nloc:])


def test_group_count_is_ceil_levels_over_m():
    A = AMG.poisson((8, 8, 8))            # 8 + 8 + 8 - 2 = 22 dependency levels
    rp, ci, va = A.csr_arrays()
    x0, b = np.zeros(A.m), np.ones(A.m)
    for m, want in ((1, 22), (2, 11), (3, 8), (5, 5), (22, 1), (40, 1)):
        for back in (0, 1):
            assert _host_sweep(rp, ci, va, A.m, A.m, m, back, x0, b)[1] == want
