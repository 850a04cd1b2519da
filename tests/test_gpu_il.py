"""Blocks of 2 / 4 / 8 right-hand sides (`MultiLevelWorkspace{TX,bs}`, multilevel.jl:28-59) on merged dependency-level
groups: the groups gather from an INTERLEAVED copy of the sweep's vector (gs_slot_il_kernel / gs_sell_il_kernel, tunable
gs_il) — one sector per matrix entry for all columns.  Same Gauss-Seidel iterate as the column-by-column kernels (another
order of a row's additions: TIGHT), every column the oracle's single-column cycle (smoother.jl:61-90, TOL)."""
import numpy as np
import pytest

import amg_amd as AMG
from conftest import uniform
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-10
TIGHT = 1e-13


def rel(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y)) / max(np.linalg.norm(y), 1e-300)


def _with(lib, name, value, fn):
    assert lib.amgh_debug_set_tunable(name, value) == 0
    try:
        return fn()
    finally:
        lib.amgh_debug_set_tunable(name, 1)


@pytest.mark.parametrize("bs", [8, 4, 2])
@pytest.mark.parametrize("kind", ["gs", "sor"])
def test_interleaved_groups_equal_column_kernels_and_oracle(bs, kind):
    A = AMG.poisson((64, 64, 48))
    n = A.m
    kw = dict(presmoother=AMG.SOR(1.2), postsmoother=AMG.SOR(1.2)) if kind == "sor" else {}
    ml = AMG.ruge_stuben(A, **kw)
    B = np.stack([uniform(n, 300 + c) - 0.2 * c for c in range(bs)], axis=1)
    lib = AMG.hip_lib()
    p = AMG.aspreconditioner(ml)
    Z = p.ldiv(B)
    Z2 = p.ldiv(B)
    assert np.array_equal(Z, Z2)                      # deterministic
    Zc = _with(lib, b"gs_il", 0, lambda: AMG.aspreconditioner(ml).ldiv(B))
    assert rel(Z, Zc) <= TIGHT
    if bs >= 4:                                       # (the interleaved kernels really ran: their sums round differently; bs = 2: the
        assert not np.array_equal(Z, Zc)              #  slot kernel alone is interleaved, the SELL groups of this hierarchy keep the column kernel)
    oh = O.OracleHierarchy(ml)
    for c in (0, bs - 1):
        assert rel(Z[:, c], oh.precond(B[:, c])) <= TOL


def test_interleaved_groups_in_w_and_f_cycles_and_solve():
    A = AMG.poisson((48, 48, 40))
    n = A.m
    ml = AMG.ruge_stuben(A)
    B = np.stack([uniform(n, 400 + c) for c in range(8)], axis=1)
    oh = O.OracleHierarchy(ml)
    for cyc in (AMG.W(), AMG.F()):
        X = AMG._solve(ml, B, cyc, maxiter=2, calculate_residual=False)
        for c in (0, 5):
            xo, _, _ = oh.solve(B[:, c], cycle=cyc.code, maxiter=2, calculate_residual=False)
            assert rel(X[:, c], xo) <= TOL
    # a single-column call on the same hierarchy afterwards (its scratch was grown, the interleaved copy is not used)
    x1 = AMG._solve(ml, B[:, 3].copy(), maxiter=2, calculate_residual=False)
    xo, _, _ = oh.solve(B[:, 3], maxiter=2, calculate_residual=False)
    assert rel(x1, xo) <= TOL


def test_interleaved_groups_float32():
    S = AMG.poisson((48, 48, 48)).to_scipy().astype(np.float32)
    ml = AMG.ruge_stuben(AMG.SparseMatrixCSC.from_scipy(S))
    n = ml.levels[0].A.m
    B = np.stack([uniform(n, 500 + c) for c in range(8)], axis=1).astype(np.float32)
    lib = AMG.hip_lib("float32")
    Z = AMG.aspreconditioner(ml).ldiv(B)
    assert Z.dtype == np.float32
    Zc = _with(lib, b"gs_il", 0, lambda: AMG.aspreconditioner(ml).ldiv(B))
    assert rel(Z.astype(np.float64), Zc.astype(np.float64)) <= 5e-5
    oh = O.OracleHierarchy(ml, dtype=np.float32)
    assert rel(Z[:, 7].astype(np.float64), oh.precond(np.ascontiguousarray(B[:, 7])).astype(np.float64)) <= 5e-5

