"""The relayed Gauss-Seidel walk's DEPENDENCY-AWARE row sum (csrc/hip/gs_relay.hpp, template flag LATE; tunable gs_bw_inorder = 0,
the library's default): the half of smoother.jl:81-87's sum that multiplies the sweep's far side is added above the hand-over,
the near half below it.  Same iterate — exact lexicographic Gauss-Seidel, every product of the row, one division — one
reassociation per row: not the scalar loop's bits (the rest of the suite pins those with gs_bw_inorder = 1), the oracle's
values to rounding.  Tolerances: a single sweep 1e-14 relative (the reassociation moves a row sum by an ulp or two), cycles and
solves the suite's 1e-10; deterministic: a rerun gives the same bits."""
import numpy as np
import pytest

import amg_amd as AMG
from amg_amd.device import DeviceHierarchy
from conftest import uniform
from oracle import oracle as O
from test_gpu_flow import _irregular_long_rows, rel, tunables

pytestmark = pytest.mark.gpu

SMOOTHERS = [AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(), AMG.GaussSeidel(iter=3),
             AMG.SOR(1.3), AMG.SOR(0.7, AMG.BackwardSweep())]


def _galerkin_19_point():
    """the second level of a 3-D Poisson hierarchy as a fine operator: rows of up to 18 off-diagonal entries, 9 on either side"""
    ml = AMG.ruge_stuben(AMG.poisson((24, 22, 20)))
    A1 = ml.levels[1].A
    assert int(np.diff(A1.to_scipy().tocsr().indptr).max()) - 1 > 12
    return A1


@pytest.mark.parametrize("case", ["poisson3d", "poisson2d", "galerkin19"])
def test_split_row_sum_is_the_oracle_sweep_to_rounding(case):
    lib = AMG.hip_lib()
    A, rows = {"poisson3d": lambda: (AMG.poisson((20, 18, 16)), 64), "poisson2d": lambda: (AMG.poisson((48, 40)), 64),
               "galerkin19": lambda: (_galerkin_19_point(), 64)}[case]()
    x0, bb = uniform(A.m, 31) - 0.5, uniform(A.m, 32)
    for pre in SMOOTHERS:
        ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=pre)
        with tunables(lib, gs_bw=2, gs_bw_rows=rows, gs_lean=0):
            dev = DeviceHierarchy(ml, 0, 1)
            assert lib.amgh_debug_bw_mode(dev.h, 0) == 3, (case, repr(pre))
            x_in = dev.smooth(0, False, x0, bb)                       # stored-order sums (the suite's default)
            assert lib.amgh_debug_bw_late(dev.h, 0) == 0
            with tunables(lib, gs_bw_inorder=0):
                assert lib.amgh_debug_bw_late(dev.h, 0) == 1, (case, repr(pre))     # these records allow the split sum
                x_late = dev.smooth(0, False, x0, bb)
                for _ in range(2):
                    assert np.array_equal(dev.smooth(0, False, x0, bb), x_late)     # deterministic, epoch after epoch
                with tunables(lib, gs_bw_dict=0):                                    # plain records: the same split, the same bits
                    assert np.array_equal(dev.smooth(0, False, x0, bb), x_late), (case, repr(pre))
            assert np.array_equal(dev.smooth(0, False, x0, bb), x_in)               # and back
            assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0
        xo = O.smooth(pre, A, x0, bb, hermitian=True)
        assert rel(x_late, xo) <= 1e-14, (case, repr(pre), rel(x_late, xo))
        assert np.max(np.abs(x_late - xo)) <= 1e-13 * np.max(np.abs(xo)), (case, repr(pre))
        if not isinstance(pre, AMG.SOR):
            assert np.array_equal(x_in, xo)


def test_rows_too_unbalanced_for_the_split_keep_the_stored_order():
    """A row with more than maxk / 2 entries on one side of the diagonal cannot be split: the whole level keeps the stored-order
    sum (amgh_debug_bw_late = 0) and the relayed sweep stays the scalar loop bit for bit, whatever the tunable says."""
    lib = AMG.hip_lib()
    A = _irregular_long_rows(6000, 3)
    x0, bb = uniform(A.m, 41) - 0.5, uniform(A.m, 42)
    pre = AMG.GaussSeidel()
    ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=pre)
    with tunables(lib, gs_bw=2, gs_bw_rows=64, gs_lean=0, gs_bw_inorder=0):
        dev = DeviceHierarchy(ml, 0, 1)
        assert lib.amgh_debug_bw_mode(dev.h, 0) == 3
        late = lib.amgh_debug_bw_late(dev.h, 0)
        x = dev.smooth(0, False, x0, bb)
    xo = O.smooth(pre, A, x0, bb, hermitian=True)
    if late == 0:
        assert np.array_equal(x, xo)
    else:
        assert rel(x, xo) <= 1e-14


@pytest.mark.parametrize("cyc", [0, 1, 2])
def test_cycles_and_solves_with_the_split_row_sum_match_the_oracle(cyc):
    lib = AMG.hip_lib()
    A = AMG.poisson((40, 36, 32))
    b = uniform(A.m, 5) - 0.4
    ml = AMG.ruge_stuben(A)
    oh = O.OracleHierarchy(ml)
    with tunables(lib, gs_bw=2, gs_bw_rows=64, gs_bw_inorder=0):
        dev = DeviceHierarchy(ml, 0, 1)
        assert lib.amgh_debug_bw_late(dev.h, 0) == 1 and lib.amgh_debug_bw_late(dev.h, 1) == 1
        for k in (1, 3):
            x, _, _ = dev.solve(b, np.zeros(A.m), cyc, k, 0.0, 0.0, False, True)
            xo, _, _ = oh.solve(b, cycle=cyc, maxiter=k, calculate_residual=False)
            assert rel(x, xo) <= 1e-10, (cyc, k)
        x, hist, _ = dev.solve(b, np.zeros(A.m), cyc, 100, 0.0, 1e-9, True, True)
        xo, ho, _ = oh.solve(b, cycle=cyc, reltol=1e-9, maxiter=100)
        assert len(hist) == len(ho) and np.allclose(hist, ho, rtol=1e-8) and rel(x, xo) <= 1e-10
        if cyc == 0:
            assert rel(dev.precond_apply(b), oh.precond(b)) <= 1e-10


def test_c3_full_size_v_cycle_with_the_split_row_sum():
    """256^3, ruge_stuben defaults, the library's default row sum: levels 0 and 1 (7- and 19-point rows) run the LATE kernels; the
    V-cycle is the oracle's to 1e-12 (the north_star asks 1e-10), the fine-level symmetric sweep to 1e-14, run to run the same bits."""
    lib = AMG.hip_lib()
    A = AMG.poisson((256, 256, 256))
    ml = AMG.ruge_stuben(A, setup="gpu", device=0)
    n = A.m
    b = uniform(n, 0)
    oh = O.OracleHierarchy(ml)
    with tunables(lib, gs_bw_inorder=0):
        dev = ml.device(0, 1)
        assert lib.amgh_debug_bw_late(dev.h, 0) == 1 and lib.amgh_debug_bw_late(dev.h, 1) == 1
        z = dev.precond_apply(b)
        assert np.array_equal(dev.precond_apply(b), z)
        x0 = uniform(n, 6) - 0.5
        xs = dev.smooth(0, False, x0, b)
        assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0
    zo = oh.precond(b)
    assert rel(z, zo) <= 1e-12, rel(z, zo)
    xo = O.smooth(ml.levels[0].presmoother, A, x0, b, hermitian=True)
    assert rel(xs, xo) <= 1e-14
    z_in = dev.precond_apply(b)                      # the stored-order sum on the same handle
    assert rel(z_in, zo) <= 1e-12 and not np.array_equal(z_in, z)


def test_float32_instance_split_row_sum():
    """The Float32 instance of the LATE kernels (gs_relay.hpp with R = float: 4 values per chunk, the reciprocal in Float32): forced
    onto a small hierarchy, both levels really run it; sweep and cycle are the Float32 oracle's within Float32 rounding, rerun
    bitwise, and the stored-order sum on the same handle is the Float32 scalar loop bit for bit (smoother.jl:61-90)."""
    from test_gpu_float32 import F32, F32_TOL, as_f32_matrix
    lib = AMG.hip_lib(F32)
    A = as_f32_matrix(AMG.poisson((24, 20, 16)))
    ml = AMG.ruge_stuben(A)
    n = A.m
    b = (uniform(n, 10) - 0.3).astype(F32)
    x0 = (uniform(n, 11) - 0.5).astype(F32)
    with tunables(lib, gs_bw=2, gs_bw_rows=128, gs_lean=0):
        dev = DeviceHierarchy(ml, 0, 1, dtype=F32)
        assert lib.amgh_debug_bw_mode(dev.h, 0) == 3
        x_in = dev.smooth(0, False, x0, b)
        with tunables(lib, gs_bw_inorder=0):
            assert lib.amgh_debug_bw_late(dev.h, 0) == 1
            x_late = dev.smooth(0, False, x0, b)
            assert np.array_equal(dev.smooth(0, False, x0, b), x_late)
            z = dev.precond_apply(b)
        assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0
    xo = O.smooth(ml.levels[0].presmoother, A, x0, b, hermitian=True, dtype=F32)
    assert np.array_equal(x_in, xo)
    assert rel(x_late.astype(np.float64), xo.astype(np.float64)) <= 1e-6
    assert x_late.dtype == F32 and rel(x_late.astype(np.float64), x_in.astype(np.float64)) <= 1e-6
    zo = O.OracleHierarchy(ml, dtype=F32).precond(b)
    assert rel(z.astype(np.float64), zo.astype(np.float64)) <= F32_TOL


def test_split_row_sum_zero_diagonals_and_rows_that_must_divide():
    """Rows the pre-scaled form cannot take: a zero diagonal (the row keeps its value and still publishes it, smoother.jl:87) and a
    diagonal outside [1e-100, 1e100] (the record's reciprocal is 0 = "divide": the LATE kernel's wave-uniform cold branch sums the
    near half plainly and divides).  A 3-D grid whose records allow the split, with such rows planted in it: the sweep is the
    oracle's to rounding, in every direction."""
    import scipy.sparse as sp
    lib = AMG.hip_lib()
    M = AMG.poisson((20, 18, 16)).to_scipy().tolil()
    n = M.shape[0]
    for r in (0, 777, 3000, n - 1):
        M[r, r] = 0.0                     # zero diagonals (stored as explicit zeros would be dropped: rows without a diagonal)
    for r in (5, 1234, 4000):
        M[r, r] = 6.0e120                 # the reciprocal would leave the range the record trusts
    for r in (9, 2222):
        M[r, r] = 3.0e-130
    A = AMG.SparseMatrixCSC.from_scipy(sp.csc_matrix(M.tocsr()))
    x0, bb = uniform(n, 51) - 0.5, uniform(n, 52)
    for pre in (AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(iter=2), AMG.SOR(1.2)):
        ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=pre, max_levels=2)
        with tunables(lib, gs_bw=2, gs_bw_rows=64, gs_lean=0):
            dev = DeviceHierarchy(ml, 0, 1)
            assert lib.amgh_debug_bw_mode(dev.h, 0) == 3, repr(pre)
            x_in = dev.smooth(0, False, x0, bb)
            with tunables(lib, gs_bw_inorder=0):
                assert lib.amgh_debug_bw_late(dev.h, 0) == 1
                x_late = dev.smooth(0, False, x0, bb)
            assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0
        xo = O.smooth(pre, A, x0, bb, hermitian=True)
        ok = np.isfinite(xo)
        assert np.array_equal(np.isfinite(x_late), ok)
        if not isinstance(pre, AMG.SOR):
            assert np.array_equal(x_in[ok], xo[ok])
        # entry by entry, relative to the entry (rows with a 3e-130 diagonal hold values of 1e129: a norm would hide everything else)
        # with the typical magnitude of the ordinary entries as the floor
        floor = 1e-3 * np.median(np.abs(xo[ok]))
        assert np.max(np.abs(x_late[ok] - xo[ok]) / np.maximum(np.abs(xo[ok]), floor)) <= 1e-12, repr(pre)
        for r in (0, 777, 3000, n - 1):
            assert x_late[r] == x0[r]      # zero diagonal: untouched
