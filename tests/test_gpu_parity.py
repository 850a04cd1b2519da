"""HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs, plus the reference's
known-answer vectors evaluated on the GPU.  Tolerance: 1e-10 relative (BASELINE.json north_star);
kernels that reproduce the oracle's IEEE sequence are additionally held to 1e-13."""
import numpy as np
import pytest
import scipy.sparse as sp

import amg_amd as AMG
from conftest import load_csc, load_npz, uniform
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-10     # north_star tolerance
TIGHT = 1e-13   # same-operation-order kernels

FWD = AMG.GaussSeidel(AMG.ForwardSweep())
BWD = AMG.GaussSeidel(AMG.BackwardSweep())
SYM = AMG.GaussSeidel()


def rel(x, y):
    d = np.linalg.norm(np.asarray(x) - np.asarray(y))
    s = max(np.linalg.norm(y), 1e-300)
    return d / s


def test_libamghip_is_loaded_and_gpu_visible():
    assert AMG.gpu_available()
    assert AMG.hip_lib().amgh_device_count() >= 1


# ---- SpMV family -------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1000,), (70, 50), (24, 20, 16)])
def test_spmv_residual_add_vs_oracle(shape):
    A = AMG.poisson(shape)
    n = A.m
    op = AMG.DeviceCSR(n, n, *A.csr_arrays())
    x, b = uniform(n, 1) - 0.5, uniform(n, 2)
    y_ref = O.spmv(A, x)
    assert rel(op.spmv(x), y_ref) <= TIGHT
    assert rel(op.residual(x, b), b - y_ref) <= TIGHT
    assert rel(op.spmv_add(x, b), b + y_ref) <= TIGHT


def test_spmv_irregular_rows_empty_rows_long_rows():
    rng = np.random.default_rng(5)
    n = 5000
    M = sp.random(n, n, 0.002, random_state=rng, format="lil")
    M[17, :] = rng.random(n)            # one dense row (longer than the LDS stage)
    M[100:140, :] = 0                   # empty rows
    M = sp.csc_matrix(M)
    A = AMG.SparseMatrixCSC.from_scipy(M)
    x = uniform(n, 9) - 0.5
    op = AMG.DeviceCSR(n, n, *A.csr_arrays())
    ref = O.spmv(A, x)
    got = op.spmv(x)
    assert rel(got, ref) <= TIGHT
    assert np.all(got[100:140] == 0)


def test_rectangular_P_and_R():
    A = AMG.poisson((40, 40))
    ml = AMG.ruge_stuben(A)
    dev = ml.device()
    lev = ml.levels[0]
    r = uniform(lev.A.m, 3)
    e = uniform(lev.P.n, 4)
    assert rel(dev.spmv(0, 2, r), O.spmv(lev.R, r)) <= TIGHT          # restriction b_c = R r
    assert rel(dev.spmv(0, 1, e), O.spmv(lev.P, e)) <= TIGHT          # prolongation P e
    assert rel(dev.spmv(0, 0, r), O.spmv(lev.A, r)) <= TIGHT
    L = len(ml.levels)
    assert rel(dev.spmv(L, 0, np.ones(ml.final_A.m)), O.spmv(ml.final_A, np.ones(ml.final_A.m))) <= TIGHT


# ---- smoothers ---------------------------------------------------------------------------------
def test_gauss_seidel_hand_values_on_gpu():  # sa_tests.jl:316-379
    def tri(N):
        return AMG.SparseMatrixCSC.from_scipy(sp.diags([-np.ones(N - 1), 2 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1]))
    x = np.array([0.0]); FWD(tri(1), x, np.zeros(1)); assert x[0] == 0
    x = np.array([0, 1, 2.0]); FWD(tri(3), x, np.zeros(3)); assert np.array_equal(x, [1 / 2, 5 / 4, 5 / 8])
    x = np.array([0, 1, 2.0]); BWD(tri(3), x, np.zeros(3)); assert np.array_equal(x, [1 / 8, 1 / 4, 1 / 2])
    x = np.array([0.0]); FWD(tri(1), x, np.array([10.0])); assert x[0] == 5.0
    x = np.array([0, 1, 2.0]); FWD(tri(3), x, np.array([10, 20, 30.0]))
    assert np.array_equal(x, [11 / 2, 55 / 4, 175 / 8])
    x = np.ones(10); AMG.GaussSeidel(AMG.SymmetricSweep(), 4)(AMG.poisson(10), x, np.zeros(10))  # issue #26
    ref = [0.176765, 0.353529, 0.497517, 0.598914, 0.653311, 0.659104, 0.615597, 0.52275, 0.382787, 0.203251]
    assert ((x - ref) ** 2).sum() < 1e-6


SMOOTHERS = [AMG.Jacobi(2 / 3), AMG.Jacobi(0.5, iter=3), AMG.Jacobi(4 / 5, iter=2), FWD, BWD, SYM,
             AMG.GaussSeidel(AMG.SymmetricSweep(), 3), AMG.SOR(0.5, iter=2), AMG.SOR(1.2, AMG.ForwardSweep()),
             AMG.SOR(0.8, AMG.BackwardSweep(), 2)]


@pytest.mark.parametrize("shape", [(300,), (48, 40), (20, 18, 16)])
def test_fast_smoothers_vs_oracle(shape):
    A = AMG.poisson(shape)
    n = A.m
    x0, b = uniform(n, 11) - 0.5, uniform(n, 12)
    for s in SMOOTHERS:
        x = x0.copy()
        s(A, x, b)
        ref = O.smooth(s, A, x0, b, hermitian=True)
        assert rel(x, ref) <= TIGHT, repr(s)


def test_smoothers_wide_dependency_levels():
    """3-D grid large enough that dependency levels exceed the chain width (wide stream-kernel path)."""
    A = AMG.poisson((64, 64, 48))
    n = A.m
    x0, b = uniform(n, 21) - 0.5, uniform(n, 22)
    for s in (SYM, AMG.SOR(0.9)):
        x = x0.copy()
        s(A, x, b)
        assert rel(x, O.smooth(s, A, x0, b)) <= TIGHT


def test_smoothers_on_amg_coarse_operator_and_zero_diagonal():
    ml = AMG.ruge_stuben(AMG.poisson((30, 30, 30)))
    A2 = ml.levels[1].A                      # irregular Galerkin operator
    n = A2.m
    x0, b = uniform(n, 31), uniform(n, 32)
    for s in (SYM, AMG.Jacobi(2 / 3, iter=2), AMG.SOR(0.7)):
        x = x0.copy(); s(A2, x, b)
        assert rel(x, O.smooth(s, A2, x0, b)) <= TIGHT
    # rows with a zero / missing diagonal are skipped (smoother.jl:87,137,218)
    M = AMG.poisson(50).to_scipy().tolil()
    M[7, 7] = 0.0
    M = sp.csc_matrix(M); M.eliminate_zeros()
    A = AMG.SparseMatrixCSC.from_scipy(M)
    x0, b = uniform(50, 33), uniform(50, 34)
    for s in (SYM, AMG.Jacobi(0.5), AMG.SOR(0.5)):
        x = x0.copy(); s(A, x, b)
        ref = O.smooth(s, A, x0, b)
        assert rel(x, ref) <= TIGHT and x[7] == x0[7]


def test_small_random_operators_through_the_single_wave_walk():
    """Operators that fit LDS are walked by one wave from their packed record (gs_wave_kernel): random patterns —
    non-symmetric, rows without a diagonal entry, zero diagonals, rows of up to ~30 entries — against the oracle, and
    bitwise against the regular chain kernel (gs_tiny = 0), NoSymmetry convention (the operator's own rows)."""
    lib = AMG.hip_lib()
    rng = np.random.default_rng(2024)
    for trial in range(12):
        n = int(rng.integers(5, 400))
        dens = float(rng.choice([0.01, 0.03, 0.08])) if n > 60 else 0.3
        M = sp.random(n, n, density=min(0.9, dens), random_state=int(rng.integers(1 << 30)), format="lil")
        if trial % 3 != 2:
            M = (M + M.T).tolil()                       # symmetric pattern two times out of three
        for i in range(n):
            M[i, i] = 4.0 + n * dens + rng.random()      # diagonally dominant: the sweeps stay O(1)
        if trial % 3 != 2:
            for i in rng.integers(0, n, size=2):
                M[int(i), int(i)] = 0.0                  # rows that keep their x (smoother.jl:87); NoSymmetry refuses them
        M = sp.csc_matrix(M); M.eliminate_zeros()
        A = AMG.SparseMatrixCSC.from_scipy(M)
        x0, b = uniform(n, 100 + trial) - 0.5, uniform(n, 200 + trial)
        sym = None if trial % 3 != 2 else AMG.NoSymmetry()
        for s in (AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(iter=2),
                  AMG.SOR(1.2, AMG.SymmetricSweep()), AMG.SOR(0.7, AMG.BackwardSweep(), iter=2)):
            got = {}
            for tiny in (1, 0):
                assert lib.amgh_debug_set_tunable(b"gs_tiny", tiny) == 0
                assert lib.amgh_debug_set_tunable(b"gs_block_inverse", 0) == 0   # (else well-conditioned triangles are inverted densely)
                try:
                    x = x0.copy()
                    AMG.device.smooth_standalone(s, A, x, b, symmetry=sym)
                    got[tiny] = x
                finally:
                    lib.amgh_debug_set_tunable(b"gs_tiny", 1)
                    lib.amgh_debug_set_tunable(b"gs_block_inverse", 1)
            assert np.array_equal(got[1], got[0]), (trial, repr(s))
            ref = O.smooth(s, A, x0, b, hermitian=sym is None)
            assert rel(got[1], ref) <= TIGHT, (trial, repr(s))


def _dense_band_spd(n, half_bw, seed, zero_diag_rows=()):
    """Irregular, densely coupled SPD-like matrix (many dependency levels per row block): the kind of operator the
    block-inverse Gauss-Seidel sweeps are selected for."""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for i in range(n):
        for j in range(max(0, i - half_bw), i):
            if rng.random() < 0.6:
                v = -rng.random()
                rows += [i, j]; cols += [j, i]; vals += [v, v]
    M = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    d = np.asarray(abs(M).sum(axis=1)).ravel() + 1.0
    for r in zero_diag_rows:
        d[r] = 0.0
    M = (M + sp.diags(d)).tocsc()
    M.eliminate_zeros()
    return AMG.SparseMatrixCSC.from_scipy(M)


@pytest.mark.parametrize("super_blocks,pipe", [(8, 1), (2, 1), (0, 1), (3, 0), (0, 0)])
def test_block_inverse_sweeps_all_variants(super_blocks, pipe):
    """Block-inverse sweeps (gs_block_kernel / gs_block_pipe_kernel, with and without superblocks) against the
    oracle's scalar lexicographic sweeps: irregular band matrices whose block steps overflow the register-resident
    leading entries, rows without a usable diagonal, a last block shorter than 128 rows, several iterations."""
    lib = AMG.hip_lib()
    lib.amgh_debug_set_tunable(b"gs_super", super_blocks)
    lib.amgh_debug_set_tunable(b"gs_block_pipe", pipe)
    lib.amgh_debug_set_tunable(b"gs_dense_tri", 0)       # (the dense whole-triangle sweeps would take these operators over)
    try:
        cases = [_dense_band_spd(1000, 40, 1), _dense_band_spd(700, 150, 2, zero_diag_rows=(5, 300, 699)),
                 _dense_band_spd(257, 30, 3), _dense_band_spd(130, 129, 4)]
        for k, A in enumerate(cases):
            n = A.m
            x0, b = uniform(n, 60 + k) - 0.5, uniform(n, 70 + k)
            for s in (FWD, BWD, SYM, AMG.GaussSeidel(AMG.SymmetricSweep(), 3)):
                x = x0.copy()
                s(A, x, b)
                ref = O.smooth(s, A, x0, b)
                assert rel(x, ref) <= 1e-11, (k, repr(s), rel(x, ref))
    finally:
        lib.amgh_debug_set_tunable(b"gs_super", 8)
        lib.amgh_debug_set_tunable(b"gs_block_pipe", 1)
        lib.amgh_debug_set_tunable(b"gs_dense_tri", 1)


@pytest.mark.parametrize("blk", [4096, 300, 256])
def test_dense_triangular_sweeps(blk):
    """Small operators: the triangle of the whole matrix (one block) or of `blk`-row diagonal blocks inverted densely on
    the device; a sweep = per block, pre-pass over everything outside the block's triangle + one triangular GEMV
    (tri_inverse_kernel / tri_gemv_kernel).  Against the oracle's scalar lexicographic sweeps, and for blocks of
    right-hand sides through a hierarchy whose coarse levels take this path."""
    lib = AMG.hip_lib()
    lib.amgh_debug_set_tunable(b"gs_dense_blk", blk)
    try:
        cases = [_dense_band_spd(1000, 40, 1), _dense_band_spd(257, 30, 3), _dense_band_spd(130, 129, 4)]
        cases.append(_dense_band_spd(9000, 25, 7))          # > 8192 rows: several diagonal blocks, the last one shorter
        for k, A in enumerate(cases):
            n = A.m
            x0, b = uniform(n, 160 + k) - 0.5, uniform(n, 170 + k)
            for s in (FWD, BWD, SYM, AMG.GaussSeidel(AMG.SymmetricSweep(), 3)):
                x = x0.copy()
                s(A, x, b)
                ref = O.smooth(s, A, x0, b)
                assert rel(x, ref) <= 1e-11, (k, repr(s), rel(x, ref))
        # inside a cycle, single and blocked right-hand sides (the coarse levels of this hierarchy are dense-swept)
        A = AMG.poisson((20, 20, 20))
        ml = AMG.ruge_stuben(A)
        dev = ml.device()
        assert any(dev.gs_sweep_steps(l) <= 3 for l in range(1, len(ml.levels)))     # a level swept in <= 3 block steps
        oh = O.OracleHierarchy(ml)
        R = np.stack([uniform(A.m, 1), uniform(A.m, 2) - 0.5, np.cos(np.arange(A.m))], axis=1)
        Z = AMG.aspreconditioner(ml).ldiv(R)
        for j in range(3):
            assert rel(Z[:, j], oh.precond(np.ascontiguousarray(R[:, j]))) <= TOL
            assert np.array_equal(Z[:, j], AMG.aspreconditioner(ml).ldiv(np.ascontiguousarray(R[:, j])))
    finally:
        lib.amgh_debug_set_tunable(b"gs_dense_blk", 4096)


@pytest.mark.parametrize("bigslot", [0, 1, 2])   # 2: long-row slot layout everywhere
def test_merged_level_sweeps_irregular_operator(bigslot):
    """Merged dependency levels (substituted groups, pre-pass over the other triangle) on irregular operators with
    rows that keep their x (zero diagonal): with and without the long-row slot layout, GS forward / backward /
    symmetric / several iterations, against the oracle's scalar sweeps."""
    lib = AMG.hip_lib()
    lib.amgh_debug_set_tunable(b"gs_block_inverse", 0)   # read at schedule build: force the level-scheduled paths
    lib.amgh_debug_set_tunable(b"gs_bigslot", bigslot)
    try:
        cases = [_dense_band_spd(6000, 12, 21, zero_diag_rows=(0, 17, 3000, 5999)), _dense_band_spd(5000, 40, 22)]
        ml = AMG.ruge_stuben(AMG.poisson((40, 36, 30)))
        cases.append(ml.levels[1].A)             # Galerkin operator of a 3-D grid (19-point-like rows)
        cases.append(AMG.ruge_stuben(AMG.poisson((64, 64, 48))).levels[2].A)   # longer rows: deep groups outgrow 512-entry slots
        # nonsymmetric values AND pattern (the schedules symmetrise the pattern for the levels only; both sides sweep
        # CSC column i as row i, smoother.jl:78)
        M = cases[1].to_scipy().tolil()
        rng = np.random.default_rng(5)
        for i in rng.integers(50, 4900, 300):
            M[i, i - int(rng.integers(1, 40))] *= 1.7
            M[i, i - 45] = -0.05
        cases.append(AMG.SparseMatrixCSC.from_scipy(sp.csc_matrix(M)))
        for k, A in enumerate(cases):
            n = A.m
            x0, b = uniform(n, 90 + k) - 0.5, uniform(n, 95 + k)
            for s in (FWD, BWD, SYM, AMG.GaussSeidel(AMG.SymmetricSweep(), 2)):
                x = x0.copy()
                s(A, x, b)
                ref = O.smooth(s, A, x0, b)
                assert rel(x, ref) <= 1e-11, (bigslot, k, repr(s), rel(x, ref))
    finally:
        lib.amgh_debug_set_tunable(b"gs_block_inverse", 1)
        lib.amgh_debug_set_tunable(b"gs_bigslot", 1)


def _halo_case(A_sq, nloc):
    """Rows [0, nloc) of a square operator as a rectangular local block (columns >= nloc are frozen halo values, as
    on a shard) + the equivalent square system for the oracle: identity rows keep the halo entries."""
    M = A_sq.to_scipy().tocsr()
    n = M.shape[0]
    top = M[:nloc, :].tocsr()
    ext = sp.vstack([top, sp.hstack([sp.csr_matrix((n - nloc, nloc)), sp.identity(n - nloc)])])
    # the oracle's fast smoothers read CSC column i as row i (smoother.jl:78): hand it the transpose
    return top, AMG.SparseMatrixCSC.from_scipy(ext.T.tocsc())


@pytest.mark.parametrize("which", ["block-inverse", "merged-levels"])
def test_sweeps_on_rectangular_shard_block_with_halo_columns(which):
    """Local block of a row-sharded operator (dist.py): Gauss-Seidel over the local rows, halo columns frozen.
    Covers the block-inverse path and the merged-level path on operators with columns >= nrows."""
    from amg_amd.device import DeviceCSR
    if which == "block-inverse":
        A_sq, nloc = _dense_band_spd(900, 60, 9), 640
    else:
        A_sq, nloc = AMG.poisson((24, 24, 24)), 24 * 24 * 20   # 66 dependency levels: merged groups are built
    top, ext = _halo_case(A_sq, nloc)
    n = A_sq.m
    x0, b = uniform(n, 81) - 0.5, uniform(n, 82)
    b_ext = b.copy(); b_ext[nloc:] = x0[nloc:]           # identity rows: x_halo = b_halo = x0_halo
    op = DeviceCSR(nloc, n, top.indptr.astype(np.int32), top.indices.astype(np.int32), top.data)
    for s in (FWD, BWD, SYM, AMG.GaussSeidel(AMG.SymmetricSweep(), 2)):
        x = op.smooth(s, x0.copy(), b[:nloc])
        ref = O.smooth(s, ext, x0, b_ext)
        assert rel(x, ref[:nloc]) <= 1e-11, (which, repr(s), rel(x, ref[:nloc]))


def test_hermitian_flag_on_nonsymmetric_matrix_sweeps_the_transpose():
    """SURVEY §7 hard part 4: with the default HermitianSymmetry the fast smoothers read CSC columns as rows."""
    rng = np.random.default_rng(3)
    n = 400
    M = sp.random(n, n, 0.02, random_state=rng, format="csc") + 6 * sp.identity(n, format="csc")
    A = AMG.SparseMatrixCSC.from_scipy(M)
    x0, b = rng.random(n), np.ones(n)
    for s in (SYM, AMG.Jacobi(0.6, iter=2), AMG.SOR(0.5, iter=2)):
        x = x0.copy(); s(A, x, b)                                     # Hermitian default
        assert rel(x, O.smooth(s, A, x0, b, hermitian=True)) <= TIGHT
        x = x0.copy(); s(A, x, b, AMG.NoSymmetry())                   # true rows
        assert rel(x, O.smooth(s, A, x0, b, hermitian=False)) <= 1e-12


def test_nosymmetry_singular_exception():
    M = sp.csc_matrix(np.array([[1.0, 2.0], [3.0, 0.0]]))
    with pytest.raises(AMG.SingularException):
        FWD(M, np.ones(2), np.ones(2), AMG.NoSymmetry())


# ---- cycles, _solve, ldiv!, cg -------------------------------------------------------------------
def _check_solve(ml, b, **kw):
    oh = O.OracleHierarchy(ml)
    x, hist = AMG._solve(ml, b, log=True, **kw)
    okw = {k: v for k, v in kw.items() if k in ("maxiter", "abstol", "reltol", "calculate_residual")}
    cyc = kw.get("cycle")
    xo, ho, it = oh.solve(b, cycle=cyc.code if cyc is not None else 0, **okw)
    assert len(hist) == len(ho), (len(hist), len(ho))
    assert rel(x, xo) <= TOL
    if len(ho) > 1:
        assert np.allclose(hist, ho, rtol=1e-7, atol=1e-14 * ho[0])
    return x


def test_solve_poisson1000_c1():  # runtests.jl:115-124
    A = AMG.poisson(1000)
    b = A @ np.ones(1000)
    x = _check_solve(AMG.ruge_stuben(A), b)
    assert ((x - 1) ** 2).sum() < 1e-8
    x = _check_solve(AMG.ruge_stuben(A, presmoother=FWD, postsmoother=FWD), b)
    assert ((x - 1) ** 2).sum() < 1e-8


@pytest.mark.parametrize("method", ["ruge_stuben", "smoothed_aggregation"])
def test_cycles_v_w_f(method):  # cycle_tests.jl:6-30
    A = AMG.poisson((50, 50))
    b = A @ np.ones(A.m)
    nb = np.linalg.norm(b)
    ml = getattr(AMG, method)(A)
    oh = O.OracleHierarchy(ml)
    for cyc in (AMG.V(), AMG.W(), AMG.F()):
        x = _check_solve(ml, b, cycle=cyc, reltol=1e-8)
        assert np.linalg.norm(b - A @ x) < 1e-8 * nb
        p = AMG.aspreconditioner(ml, cyc)
        x, log = AMG.cg(A, b, Pl=p, reltol=1e-8, log=True)
        xo, ho, ito = oh.pcg(b, cycle=cyc.code, reltol=1e-8)
        assert np.linalg.norm(b - A @ x) <= 1e-8 * nb
        assert log["iters"] == ito and rel(x, xo) <= 1e-9
        z = p.ldiv(b)
        assert rel(z, oh.precond(b, cyc.code)) <= TOL


def test_thing_known_answer_vectors_on_gpu():  # runtests.jl:143-224
    A = load_csc("thing")
    g = load_npz("thing_solutions")
    n = 46
    b = np.zeros(n); b[0], b[1] = 1, -1
    ml = AMG.ruge_stuben(A, presmoother=FWD, postsmoother=FWD, coarse_solver=AMG.Pinv)
    x = AMG._solve(ml, A @ np.ones(n), maxiter=1, abstol=1e-12)
    assert ((x - g["solve_Aones_fwd_maxiter1"]) ** 2).sum() < 1e-8
    x = AMG.solve(A, b, AMG.RugeStubenAMG(), presmoother=FWD, postsmoother=FWD, maxiter=1, abstol=1e-12,
                  coarse_solver=AMG.Pinv)
    assert ((x - g["solve_b_fwd_maxiter1"]) ** 2).sum() < 1e-8
    x = AMG.cg(A, b, Pl=AMG.aspreconditioner(ml))
    assert ((x - g["cg_fwd"]) ** 2).sum() < 1e-8
    ml = AMG.ruge_stuben(A, coarse_solver=AMG.Pinv)
    x = AMG.cg(A, b, Pl=AMG.aspreconditioner(ml), maxiter=100000, reltol=1e-6)
    assert ((x - g["cg_sym_reltol1e-6"]) ** 2).sum() < 1e-8
    x = AMG._solve(ml, b, maxiter=1, reltol=1e-12)
    assert ((x - g["solve_b_sym_maxiter1"]) ** 2).sum() < 1e-8


def test_sa_jacobi_config_c2_small():
    """Config C2 at reduced size: smoothed_aggregation + Jacobi(2/3) pre/post."""
    A = AMG.poisson((96, 96))
    jac = AMG.Jacobi(2 / 3)
    ml = AMG.smoothed_aggregation(A, presmoother=jac, postsmoother=jac)
    b = A @ np.ones(A.m)
    x = _check_solve(ml, b, reltol=1e-8, maxiter=200)
    assert np.linalg.norm(b - A @ x) < 1e-8 * np.linalg.norm(b)
    b2 = uniform(A.m, 0)
    _check_solve(ml, b2, reltol=1e-8, maxiter=200)


def test_rs_gs_config_c3_small():
    """Config C3 at reduced size: 3-D Poisson, ruge_stuben defaults (symmetric GS)."""
    A = AMG.poisson((40, 40, 40))
    ml = AMG.ruge_stuben(A)
    b = uniform(A.m, 0)
    x = _check_solve(ml, b)
    assert np.linalg.norm(b - A @ x) < 1.5e-8 * np.linalg.norm(b)


def test_lin_elastic_c5():  # nns_test.jl:213-226
    d = load_npz("lin_elastic_2d")
    A = load_csc("lin_elastic_2d")
    ml = AMG.smoothed_aggregation(A, B=d["B"])
    x, hist = AMG.solve(A, d["b"], AMG.SmoothedAggregationAMG(), log=True, reltol=1e-10, B=d["B"])
    assert len(hist) - 1 == 27
    assert np.linalg.norm(A @ x - d["b"]) <= 1.5e-8 * np.linalg.norm(d["b"])
    xo, _, _ = O.OracleHierarchy(ml).solve(d["b"], reltol=1e-10)
    assert rel(x, xo) <= TOL
    xp, log = AMG.cg(A, d["b"], Pl=AMG.aspreconditioner(ml), reltol=1e-10, log=True)
    xpo, _, itp = O.OracleHierarchy(ml).pcg(d["b"], reltol=1e-10)
    assert log["iters"] == itp == 13 and rel(xp, xpo) <= 1e-9


@pytest.mark.parametrize("case", ["c5", "small", "large"])
def test_pcg_launch_plans_are_bitwise_the_same_recurrence(case):
    """amgh_pcg between two cycles: one launch per operation (pcg_fused = 0, 15 launches), or the dot products' second stages
    fused with the scalar steps and the updates with the norm (1, 8 launches).  The same sums, products and quotients in the
    same order: identical iterates, residual histories and counts (IterativeSolvers' cg recurrence, runtests.jl:186,204)."""
    if case == "c5":
        d = load_npz("lin_elastic_2d")
        A, b = load_csc("lin_elastic_2d"), d["b"]
        ml = AMG.smoothed_aggregation(A, B=d["B"])
    else:
        A = AMG.poisson((20, 20, 20)) if case == "small" else AMG.poisson((30, 30, 30))
        b = uniform(A.m, 91) - 0.4
        ml = AMG.ruge_stuben(A)
    lib = AMG.hip_lib()
    out = {}
    try:
        for plan in (0, 1):
            assert lib.amgh_debug_set_tunable(b"pcg_fused", plan) == 0
            out[plan] = AMG.cg(A, b, Pl=AMG.aspreconditioner(ml), reltol=1e-10, log=True)
    finally:
        lib.amgh_debug_set_tunable(b"pcg_fused", 1)
    for plan in (1,):
        assert out[plan][1]["iters"] == out[0][1]["iters"]
        assert np.array_equal(out[plan][1]["resnorm"], out[0][1]["resnorm"])
        assert np.array_equal(out[plan][0], out[0][0])
    xo, _, ito = O.OracleHierarchy(ml).pcg(b, reltol=1e-10)
    assert out[1][1]["iters"] == ito and rel(out[1][0], xo) <= 1e-9


def test_degenerate_hierarchies():
    # no levels: every "cycle" is the coarse solve (multilevel.jl:179-180)
    for sz in (10, 5, 2):
        A = AMG.poisson(sz)
        ml = AMG.ruge_stuben(A)
        b = A @ np.ones(sz)
        x = _check_solve(ml, b)
        assert np.allclose(x, 1.0)
    # b = 0: zero iterations, x = 0, residuals == [0]
    ml = AMG.ruge_stuben(AMG.poisson(100))
    x, hist = AMG._solve(ml, np.zeros(100), log=True)
    assert np.all(x == 0) and hist.tolist() == [0.0]
    # calculate_residual = false: exactly maxiter cycles
    b = AMG.poisson(100) @ np.ones(100)
    _check_solve(ml, b, maxiter=3, calculate_residual=False)
    # big single-level problem through the pluggable host coarse solver (issue #56, SA branch)
    X = AMG.SparseMatrixCSC.from_scipy(AMG.poisson(27000).to_scipy() + 24.0 * sp.identity(27000, format="csc"))
    ml = AMG.smoothed_aggregation(X, strength=AMG.SymmetricStrength(0.05))
    assert len(ml) == 1
    b = uniform(27000, 56)
    x = AMG._solve(ml, b, reltol=1e-10)
    assert np.linalg.norm(X @ x - b) <= 1e-10 * np.linalg.norm(b)


def test_issue56_tight_tolerance_rs():  # test_regression.jl:59-66
    X = AMG.SparseMatrixCSC.from_scipy(AMG.poisson(27000).to_scipy() + 24.0 * sp.identity(27000, format="csc"))
    b = uniform(27000, 56)
    ml = AMG.ruge_stuben(X)
    x = _check_solve(ml, b, reltol=1e-10)
    import scipy.sparse.linalg as spla
    ref = spla.spsolve(X.to_scipy(), b)
    assert rel(x, ref) <= 1e-10


def test_issue95_nosymmetry_hierarchy():  # test_regression.jl:71-83
    rng = np.random.default_rng(95)
    N = 10000
    M = sp.random(N, N, 0.001, random_state=rng, format="csc") + 5 * sp.identity(N, format="csc")
    b = np.ones(N)
    for f in (AMG.ruge_stuben, AMG.smoothed_aggregation):
        ml = f(M, symmetry=AMG.NoSymmetry())
        x = AMG._solve(ml, b)
        xo, _, _ = O.OracleHierarchy(ml).solve(b)
        assert np.linalg.norm(M @ x - b) <= 1e-8 * np.linalg.norm(b)
        assert rel(x, xo) <= TOL


def test_user_assembled_geometric_hierarchy():  # test/gmg.jl:1-49 via the public Level/MultiLevel constructors
    def extend(A):
        nF = A.m
        nC = (nF - 1) // 2 + 1 if nF % 2 == 0 else (nF - 1) // 2
        I, J, V = [], [], []
        for k in range(1, nC + 1):
            I.append(2 * k); J.append(k); V.append(1.0)
        for k in range(1, nC):
            I += [2 * k + 1, 2 * k + 1]; J += [k, k + 1]; V += [0.5, 0.5]
        P = sp.csc_matrix((V, (np.array(I) - 1, np.array(J) - 1)), shape=(nF, nC))
        return P, sp.csc_matrix(P.T)
    A = AMG.poisson(4000)
    levels = []
    while len(levels) + 1 < 10 and A.m > 10:
        P, R = extend(A)
        levels.append(AMG.Level(A, P, R, SYM, SYM))
        A = AMG.SparseMatrixCSC.from_scipy(R @ A.to_scipy() @ P)
    ml = AMG.MultiLevel(levels, A, AMG.Pinv(A), SYM, SYM)
    assert len(ml) == 10
    b = levels[0].A @ np.ones(4000)
    x, hist = AMG._solve(ml, b, log=True, maxiter=20)
    xo, ho, _ = O.OracleHierarchy(ml).solve(b, maxiter=20)
    assert len(hist) == len(ho) and rel(x, xo) <= TOL


def test_full_size_properties_spmv_linearity_and_symmetry():
    """Size-independent properties at a larger size than the oracle comparison: linearity of A x,
    <Ax,y> = <x,Ay> (symmetric operator), residual(x=0) = b."""
    A = AMG.poisson((128, 128, 64))
    n = A.m
    op = AMG.DeviceCSR(n, n, *A.csr_arrays())
    x, y = uniform(n, 1), uniform(n, 2)
    Ax, Ay = op.spmv(x), op.spmv(y)
    assert rel(op.spmv(2.0 * x - 3.0 * y), 2.0 * Ax - 3.0 * Ay) <= 1e-13
    assert abs(Ax @ y - x @ Ay) <= 1e-12 * abs(Ax @ y)
    assert np.array_equal(op.residual(np.zeros(n), y), y)
    # interior rows of the 7-point stencil annihilate constants
    r = op.spmv(np.ones(n)).reshape((64, 128, 128))
    assert np.all(r[1:-1, 1:-1, 1:-1] == 0)


def test_multiple_right_hand_sides_block_workspace():
    """bs > 1 (`MultiLevelWorkspace{TX,bs}`, multilevel.jl:28-59): b, x are n x bs; the smoothers and
    operators act column by column, the stopping test uses the norm of the whole block."""
    A = AMG.poisson((30, 30))
    n = A.m
    B = np.stack([uniform(n, 41), A @ np.ones(n), uniform(n, 43) - 0.5], axis=1)
    for ml in (AMG.ruge_stuben(A), AMG.smoothed_aggregation(A, presmoother=AMG.Jacobi(2 / 3), postsmoother=AMG.Jacobi(2 / 3))):
        oh = O.OracleHierarchy(ml)
        # fixed number of cycles: every column equals the single-RHS oracle run
        X = AMG._solve(ml, B, maxiter=4, calculate_residual=False)
        assert X.shape == B.shape
        for c in range(3):
            xo, _, _ = oh.solve(B[:, c], maxiter=4, calculate_residual=False)
            assert rel(X[:, c], xo) <= TOL
        # tolerance-driven: joint Frobenius-norm test, same cycle count for all columns
        X, hist = AMG._solve(ml, B, reltol=1e-8, log=True, maxiter=100)
        its = len(hist) - 1
        Xo = np.stack([oh.solve(B[:, c], maxiter=its, calculate_residual=False)[0] for c in range(3)], axis=1)
        assert rel(X, Xo) <= TOL
        R = B - A.to_scipy() @ X
        assert np.isclose(hist[-1], np.linalg.norm(R), rtol=1e-6) and hist[-1] <= 1e-8 * np.linalg.norm(B)
        assert hist[-2] > 1e-8 * np.linalg.norm(B)
        Z = AMG.aspreconditioner(ml).ldiv(B)
        for c in range(3):
            assert rel(Z[:, c], oh.precond(B[:, c])) <= TOL


@pytest.mark.parametrize("bs,sor", [(8, False), (6, False), (3, False), (4, True)])
def test_block_of_right_hand_sides_on_long_row_groups_equals_single_columns_bitwise(bs, sor):
    """Groups with long composite rows run from the SELL-like copy; a block of right-hand sides goes through the same
    kernel with 8 / 2 / 1 columns per launch (the matrix entries read once for all of them) and every column's
    additions in the single-column order: bit for bit the column-by-column result, and the oracle's within TOL."""
    A = AMG.poisson((64, 64, 64))
    n = A.m
    ml = AMG.ruge_stuben(A, presmoother=AMG.SOR(1.2), postsmoother=AMG.SOR(1.2)) if sor else AMG.ruge_stuben(A)
    B = np.stack([uniform(n, 70 + c) - 0.25 * c for c in range(bs)], axis=1)
    lib = AMG.hip_lib()
    # (single right-hand sides sum the 50-100-entry rows of the slot launches with 8 / 16 lanes per row and the short
    #  ones two entries per thread — another order of additions; switch both off to compare like with like)
    assert lib.amgh_debug_set_tunable(b"gs_lpr", 1) == 0 and lib.amgh_debug_set_tunable(b"gs_ept", 1) == 0
    try:
        Z = AMG.aspreconditioner(ml).ldiv(B)
        singles = [AMG.aspreconditioner(ml).ldiv(B[:, c].copy()) for c in range(bs)]
        # blocks of 2 / 4 / 8 / 16 columns restrict and prolong through an INTERLEAVED copy of the gathered vector (one
        # sector serves every column); column by column (rhs_il = 0) must give the same bits
        assert lib.amgh_debug_set_tunable(b"rhs_il", 0) == 0
        Z_cols = AMG.aspreconditioner(ml).ldiv(B)
        assert np.array_equal(Z, Z_cols)
    finally:
        lib.amgh_debug_set_tunable(b"rhs_il", 1)
        lib.amgh_debug_set_tunable(b"gs_lpr", 0)
        lib.amgh_debug_set_tunable(b"gs_ept", 0)
    oh = O.OracleHierarchy(ml)
    for c in range(bs):
        assert np.array_equal(Z[:, c], singles[c]), (c, rel(Z[:, c], singles[c]))
        if c in (0, bs - 1):
            assert rel(Z[:, c], oh.precond(B[:, c])) <= TOL


def test_multiple_right_hand_sides_all_sweep_kernels():
    """bs = 4 on a hierarchy whose levels go through every Gauss-Seidel execution path (slot kernel on the wide
    dependency levels, single-workgroup chains, block-inverse sweeps) and through SOR / Jacobi: one launch
    covers all columns, each column must still equal the single-RHS oracle run."""
    A = AMG.poisson((40, 40, 40))
    n = A.m
    B = np.stack([uniform(n, 51), A @ np.ones(n), uniform(n, 53) - 0.5, np.cos(np.arange(n))], axis=1)
    cases = [
        (dict(), "V"),
        (dict(presmoother=AMG.SOR(1.2, iter=2), postsmoother=AMG.SOR(0.8, sweep=AMG.BackwardSweep())), "W"),
        (dict(presmoother=AMG.Jacobi(2 / 3, iter=2), postsmoother=AMG.GaussSeidel(AMG.ForwardSweep())), "F"),
    ]
    for kw, cyc in cases:
        ml = AMG.ruge_stuben(A, **kw)
        oh = O.OracleHierarchy(ml)
        cycle = {"V": AMG.V(), "W": AMG.W(), "F": AMG.F()}[cyc]
        X = AMG._solve(ml, B, cycle, maxiter=3, calculate_residual=False)
        for c in range(B.shape[1]):
            xo, _, _ = oh.solve(B[:, c], cycle=cycle.code, maxiter=3, calculate_residual=False)
            assert rel(X[:, c], xo) <= TOL, (kw, cyc, c)
        # the same hierarchy still serves single-RHS calls (its scratch was grown, not replaced by a smaller one)
        x1 = AMG._solve(ml, B[:, 2].copy(), cycle, maxiter=3, calculate_residual=False)
        assert rel(x1, X[:, 2]) <= 1e-14
        # wider blocks take the 8- and 2-columns-per-workgroup sweep kernels; same columns, same results
        for reps in (2, 1.5):
            Bw = np.concatenate([B] * 2, axis=1)[:, : int(4 * reps)]
            Xw = AMG._solve(ml, Bw, cycle, maxiter=3, calculate_residual=False)
            assert rel(Xw[:, :4], X) <= 1e-14 and rel(Xw[:, 4:], X[:, : Bw.shape[1] - 4]) <= 1e-14


def test_cycles_are_bitwise_reproducible_run_to_run():
    """No data races in the sweeps (merged groups, long-row slots, block-inverse pipeline, chains): the same input
    gives bit-identical output every time, single and multi-RHS."""
    A = AMG.poisson((48, 48, 40))
    ml = AMG.ruge_stuben(A)
    n = A.m
    p = AMG.aspreconditioner(ml)
    r = uniform(n, 77) - 0.3
    z0 = p.ldiv(r)
    for _ in range(6):
        assert np.array_equal(p.ldiv(r), z0)
    R = np.stack([r, uniform(n, 78), np.sin(np.arange(n))], axis=1)
    Z0 = p.ldiv(R)
    for _ in range(3):
        assert np.array_equal(p.ldiv(R), Z0)
    assert rel(Z0[:, 0], z0) <= 1e-14


def test_eltype_promotion_contract():  # runtests.jl:244-259
    a = AMG.poisson(100).to_scipy()
    b = uniform(100, 1)
    for T, V in ((np.float64, np.float64), (np.float32, np.float32), (np.float64, np.float32), (np.float32, np.float64)):
        ml = AMG.smoothed_aggregation(a.astype(T))
        x = AMG._solve(ml, b.astype(V))
        assert x.dtype == np.promote_types(T, V)
        # the default reltol is sqrt(eps(eltype(b))) (multilevel.jl:162): a Float32 right-hand side stops at 3.5e-4;
        # a Float32 result also carries its rounding (|x| ~ 2e2, cond ~ 4e3) into the residual
        assert np.linalg.norm(a @ x.astype(np.float64) - b) <= (1e-3 if V == np.float32 else 1e-7) * np.linalg.norm(b)


def test_level_ordered_handover_between_levels_is_bitwise_the_natural_one():
    """gs_coarse_lo: Rp's rows / Pp's columns renumbered to the next level's dependency-level order, the restricted
    residual written into that level's level-ordered right-hand side, the correction read from its level-ordered x.
    Every sum keeps its order: V / W / F cycles and blocks of right-hand sides come out bit for bit."""
    from amg_amd.device import DeviceHierarchy
    lib = AMG.hip_lib()
    A = AMG.poisson((40, 36, 32))
    ml = AMG.ruge_stuben(A)
    n = A.m
    R = np.asfortranarray(np.stack([uniform(n, 5), uniform(n, 6) - 0.5, np.sin(np.arange(n))], axis=1))
    out = {}
    try:
        for flag in (0, 1):
            lib.amgh_debug_set_tunable(b"gs_coarse_lo", flag)
            for bs in (1, 3):
                dev = DeviceHierarchy(ml, 0, bs)
                for cyc in (0, 1, 2):
                    out[(flag, bs, cyc)] = dev.precond_apply(R if bs == 3 else np.ascontiguousarray(R[:, 0]), cyc)
                del dev
    finally:
        lib.amgh_debug_set_tunable(b"gs_coarse_lo", 1)
    for bs in (1, 3):
        for cyc in (0, 1, 2):
            assert np.array_equal(out[(0, bs, cyc)], out[(1, bs, cyc)]), (bs, cyc)
    oh = O.OracleHierarchy(ml)
    assert rel(out[(1, 1, 0)], oh.precond(np.ascontiguousarray(R[:, 0]))) <= TOL
