"""The Float32 instance of the solve phase (libamghip_f32.so: the library's source compiled with amgh_real = float)
against the Float32 instance of the CPU oracle (liboracle_f32.so: the restatement compiled with real_t = float).

The reference's hierarchy is generic in eltype(A); its tests run Float64 and Float32 (test/runtests.jl:244-259).
Both sides here get the SAME Float32 matrices (the Float64 hierarchy rounded once), so what is compared is the Float32
arithmetic of the cycle.  Tolerance: the two differ in the ORDER of Float32 additions only where the Float64 paths
differ at 1e-13 .. 1e-11 (merged dependency levels, lanes-per-row sums); scaled by eps(Float32) / eps(Float64) that is
5e-5 relative in the 2-norm — written below as F32_TOL."""
import ctypes as C

import os

import numpy as np
import pytest

import amg_amd as AMG
from amg_amd.device import DeviceCSR, DeviceHierarchy, smooth_standalone
from conftest import load_csc, uniform
from oracle import oracle as O

pytestmark = pytest.mark.gpu

F32_TOL = 5e-5
F32 = np.float32


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def as_f32_matrix(A):
    """The Float32 matrix a Julia user would pass: T.(A)."""
    S = A.to_scipy().astype(F32)
    return AMG.SparseMatrixCSC.from_scipy(S)


def test_the_float32_instance_is_a_different_library_with_the_same_entry_points():
    l64, l32 = AMG.hip_lib(), AMG.hip_lib("float32")
    assert l64 is not l32 and l64.real_dtype == "float64" and l32.real_dtype == "float32"
    for name in ("amgh_create", "amgh_push_level", "amgh_push_level_begin", "amgh_push_level_end", "amgh_set_coarse",
                 "amgh_finalize", "amgh_solve", "amgh_solve_d", "amgh_precond_apply_d", "amgh_pcg", "amgh_cycle_d",
                 "amgh_level_spmv", "amgh_level_smooth", "amgh_csr_create", "amgh_csr_gs_d", "amgh_dot_d"):
        assert hasattr(l32, name), name
    assert hasattr(l32, "amgh_dist_create_rccl") and hasattr(l32, "amgh_dist_create_ipc")   # the row-sharded path too
    assert not hasattr(l32, "amgh_setup_spgemm")          # the GPU half of the setup is Float64 only


def test_standalone_operators_in_float32():
    A = AMG.poisson((20, 18, 16))
    n = A.m
    rp, ci, va = A.csr_arrays()
    op = DeviceCSR(n, n, rp, ci, va, dtype=F32)
    x, b = (uniform(n, 3) - 0.5).astype(F32), uniform(n, 4).astype(F32)
    y = op.spmv(x)
    assert y.dtype == F32 and rel(y, O.spmv(A, x, dtype=F32)) <= 1e-6
    r = op.residual(x, b)
    assert rel(r, b - O.spmv(A, x, dtype=F32)) <= 1e-6
    # dot / norm through the Float32 reduction
    lib = AMG.hip_lib("float32")
    xd, sc = AMG.DeviceBuffer(n, 0, x, dtype=F32), AMG.DeviceBuffer(2048, 0, dtype=F32)
    out = C.c_float(0)
    assert lib.amgh_dot_d(0, n, xd.ptr, xd.ptr, sc.ptr, C.byref(out), None) == 0
    assert abs(out.value - float(np.dot(x.astype(np.float64), x.astype(np.float64)))) <= 1e-5 * float(np.dot(x, x))


@pytest.mark.parametrize("shape", [(300,), (48, 40), (20, 18, 16), (64, 64, 48)])
def test_smoothers_in_float32_vs_float32_oracle(shape):
    A = AMG.poisson(shape)
    n = A.m
    x0, b = (uniform(n, 11) - 0.5).astype(F32), uniform(n, 12).astype(F32)
    for s in (AMG.Jacobi(2 / 3), AMG.Jacobi(0.5, iter=3), AMG.GaussSeidel(AMG.ForwardSweep()),
              AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(), AMG.GaussSeidel(AMG.SymmetricSweep(), 3),
              AMG.SOR(0.5, iter=2), AMG.SOR(1.2, AMG.ForwardSweep())):
        x = x0.copy()
        smooth_standalone(s, A, x, b, dtype=F32)
        ref = O.smooth(s, A, x0, b, hermitian=True, dtype=F32)
        assert x.dtype == F32 and rel(x, ref) <= F32_TOL, repr(s)
        assert rel(x, x0) > 1e-3                 # it did sweep


def _hierarchies():
    yield "rs poisson 3-D", AMG.ruge_stuben(as_f32_matrix(AMG.poisson((24, 24, 24))))
    yield "rs poisson 2-D", AMG.ruge_stuben(as_f32_matrix(AMG.poisson((70, 50))))
    yield "sa poisson 1-D", AMG.smoothed_aggregation(as_f32_matrix(AMG.poisson(1000)))
    yield "sa poisson 3-D jacobi", AMG.smoothed_aggregation(as_f32_matrix(AMG.poisson((20, 20, 20))),
                                                              presmoother=AMG.Jacobi(2 / 3), postsmoother=AMG.Jacobi(2 / 3))
    yield "rs randlap sor", AMG.ruge_stuben(as_f32_matrix(load_csc("randlap")), presmoother=AMG.SOR(1.1),
                                            postsmoother=AMG.SOR(1.1))


@pytest.mark.parametrize("cycle", [0, 1, 2])
def test_cycles_in_float32_vs_float32_oracle(cycle):
    cyc = (AMG.V(), AMG.W(), AMG.F())[cycle]
    for name, ml in _hierarchies():
        A0 = ml.levels[0].A
        assert A0.eltype == F32, name
        r = uniform(A0.m, 5).astype(F32)
        z = AMG.aspreconditioner(ml, cyc).ldiv(r)
        assert z.dtype == F32, name
        dev = ml.device(dtype=F32)
        assert dev.lib is AMG.hip_lib("float32") and dev.dtype == F32      # it ran on the Float32 instance
        ref = O.OracleHierarchy(ml, dtype=F32).precond(r, cycle)
        assert ref.dtype == F32
        assert rel(z, ref) <= F32_TOL, (name, rel(z, ref))
        # and it is Float32 arithmetic, not a rounded Float64 cycle: the Float64 result differs at the 1e-7 level
        z64 = O.OracleHierarchy(ml).precond(r.astype(np.float64), cycle)
        assert 1e-9 < rel(z, z64) < 1e-3, (name, rel(z, z64))


def test_wavefront_of_blocks_in_float32():
    """The Float32 instance sweeps a single-column fine level as a wavefront of blocks too (packed rows of four values per
    16-byte chunk, the quotient as reciprocal + one fmaf correction): forced on a small 3-D operator, against the Float32
    oracle and against the level schedules of the same instance."""
    from amg_amd.device import DeviceHierarchy
    lib = AMG.hip_lib("float32")
    A = as_f32_matrix(AMG.poisson((18, 16, 14)))
    for pre, post in ((AMG.GaussSeidel(), AMG.GaussSeidel()), (AMG.SOR(1.1), AMG.GaussSeidel(AMG.BackwardSweep(), iter=2))):
        ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=post)
        r = uniform(A.m, 9).astype(F32)
        ref = O.OracleHierarchy(ml, dtype=F32).precond(r)
        out = {}
        for mode in (2, 0):
            assert lib.amgh_debug_set_tunable(b"gs_bw", mode) == 0 and lib.amgh_debug_set_tunable(b"gs_bw_rows", 64) == 0
            try:
                dev = DeviceHierarchy(ml, 0, 1, dtype=F32)
                if mode == 2:
                    assert dev.gs_sweep_stats(0, False)["slot_entries"] == 6 * A.m      # it is the block layout
                out[mode] = dev.precond_apply(r)
            finally:
                lib.amgh_debug_set_tunable(b"gs_bw", 1)
                lib.amgh_debug_set_tunable(b"gs_bw_rows", 512)
        assert out[2].dtype == F32 and rel(out[2], ref) <= F32_TOL and rel(out[2], out[0]) <= F32_TOL


def test_solve_in_float32_follows_the_float32_oracle():
    for name, ml in _hierarchies():
        A0 = ml.levels[0].A
        n = A0.m
        b = (A0.to_scipy() @ np.ones(n)).astype(F32)
        x, hist = AMG._solve(ml, b, log=True, maxiter=40)
        assert x.dtype == F32 and hist.dtype == F32, name
        oh = O.OracleHierarchy(ml, dtype=F32)
        xo, ho, ito = oh.solve(b, maxiter=40)            # default reltol = sqrt(eps(Float32)) on both sides
        assert abs(len(hist) - len(ho)) <= 1, (name, len(hist), len(ho))
        k = min(len(hist), len(ho))
        # early residuals agree tightly; near the Float32 floor they are rounding noise on both sides
        assert np.allclose(hist[:min(k, 4)], ho[:min(k, 4)], rtol=1e-3), name
        assert hist[k - 1] <= 10 * max(ho[k - 1], 3e-4 * ho[0]), name
        assert rel(x, xo) <= 1e-3, (name, rel(x, xo))
        if "poisson" in name:        # (randlap is a singular graph Laplacian: A * ones = 0)
            assert rel(x, np.ones(n)) <= 5e-3, name


def test_multiple_right_hand_sides_in_float32():
    ml = AMG.ruge_stuben(as_f32_matrix(AMG.poisson((20, 20, 16))))
    n = ml.levels[0].A.m
    B = np.stack([uniform(n, 1), uniform(n, 2) - 0.5, np.sin(np.arange(n))], axis=1).astype(F32)
    Z = AMG.aspreconditioner(ml).ldiv(B)
    assert Z.dtype == F32 and Z.shape == B.shape
    oh = O.OracleHierarchy(ml, dtype=F32)
    for j in range(B.shape[1]):
        assert rel(Z[:, j], oh.precond(np.ascontiguousarray(B[:, j]))) <= F32_TOL


def test_eltype_rules_pick_the_instance():   # runtests.jl:244-259, multilevel.jl:154
    a = AMG.poisson(100).to_scipy()
    b = uniform(100, 1)
    for T, V in ((np.float64, np.float64), (F32, F32), (np.float64, F32), (F32, np.float64)):
        ml = AMG.smoothed_aggregation(a.astype(T))
        x = AMG._solve(ml, b.astype(V))
        assert x.dtype == np.promote_types(T, V)
        both32 = T == F32 and V == F32
        assert ((0, 1, "f32") in ml._dev) == both32 and ((0, 1) in ml._dev) == (not both32)


def test_float32_handle_keeps_exact_order_sweeps_where_float64_inverts_blocks():
    """Block-inverse sweeps are accurate to cond * eps: admitted up to cond 1e4 in Float64, never in Float32."""
    ml = AMG.ruge_stuben(as_f32_matrix(AMG.poisson((24, 24, 24))))
    d64, d32 = ml.device(), ml.device(dtype=F32)
    L = len(ml.levels)
    steps64 = [d64.gs_sweep_steps(l) for l in range(L)]
    steps32 = [d32.gs_sweep_steps(l) for l in range(L)]
    levels = [d32.gs_dependency_levels(l) for l in range(L)]
    assert levels == [d64.gs_dependency_levels(l) for l in range(L)]
    assert all(s32 >= s64 for s32, s64 in zip(steps32, steps64))


def test_row_sharded_cycle_in_float32():
    """The sharded path of the Float32 instance (amgh_dist_* with amgh_real = float): Jacobi-smoothed hierarchy on 2
    and 4 virtual ranks against the Float32 oracle (exact across shards, up to Float32 rounding), Gauss-Seidel against
    the Float64 sharded run."""
    from amg_amd import sharded as SH
    A = AMG.poisson((32, 24, 20))
    A32 = as_f32_matrix(A)
    b = uniform(A.m, 5).astype(F32)
    jac = AMG.Jacobi(2.0 / 3.0, iter=2)
    ml = AMG.ruge_stuben(A32, presmoother=jac, postsmoother=jac)
    for nranks in (2, 4):
        def work(rank, group):
            sh = SH.ShardedHierarchy.from_multilevel(ml, rank, nranks, 0, ("local", group), 500, dtype=F32, gs_mode="hybrid")
            x, hist = sh.solve(b[sh.r0:sh.r1], maxiter=4, calculate_residual=True, reltol=1e-30)
            z = sh.precond_apply(b[sh.r0:sh.r1])
            return x, hist, z, sh.lc
        res = SH.run_local_ranks(nranks, work, dtype=F32)
        x = np.concatenate([r[0] for r in res])
        z = np.concatenate([r[2] for r in res])
        assert x.dtype == F32 and res[0][1].dtype == F32 and res[0][3] >= 2
        oh = O.OracleHierarchy(ml, dtype=F32)
        xo, ho, _ = oh.solve(b, maxiter=4, reltol=1e-30)
        assert rel(x, xo) <= F32_TOL and np.allclose(res[0][1], ho, rtol=1e-3)
        assert rel(z, oh.precond(b)) <= F32_TOL
    # default smoother: the frozen-halo hybrid in Float32 stays within Float32 rounding of its Float64 run
    ml_gs = AMG.ruge_stuben(A32)
    out = {}
    for dt in (F32, np.float64):
        def work(rank, group, dt=dt):
            sh = SH.ShardedHierarchy.from_multilevel(ml_gs, rank, 2, 0, ("local", group), 500, dtype=dt, gs_mode="hybrid")
            return sh.precond_apply(b[sh.r0:sh.r1].astype(dt))
        out[np.dtype(dt).name] = np.concatenate(SH.run_local_ranks(2, work, dtype=dt))
    assert out["float32"].dtype == F32 and rel(out["float32"], out["float64"]) <= F32_TOL


def test_float32_gauss_seidel_pipelined_across_the_ranks():
    """The Float32 instance of the exact sweep pipelined across the ranks (8-byte mailboxes {value, tag} polled in the neighbour's
    array): 3 virtual ranks, block layouts forced on the small shards — the pipeline really runs (gs_pipelined), equals the ranks
    sweeping in turn bit for bit, and is the Float32 oracle's cycle within Float32 rounding."""
    from amg_amd import sharded as SH
    lib = AMG.hip_lib(F32)
    A32 = as_f32_matrix(AMG.poisson((32, 28, 36)))
    b = (uniform(A32.m, 8) - 0.4).astype(F32)
    ml = AMG.ruge_stuben(A32)
    for name, v in ((b"gs_bw", 2), (b"gs_bw_rows", 64)):
        assert lib.amgh_debug_set_tunable(name, v) == 0
    serialized = False
    try:
        out = {}
        for mode in ("exact", "exact-turns"):
            def work(rank, group, mode=mode):
                sh = SH.ShardedHierarchy.from_multilevel(ml, rank, 3, 0, ("local", group), 4000, dtype=F32, gs_mode=mode)
                x, _ = sh.solve(b[sh.r0:sh.r1], maxiter=2, calculate_residual=False)
                return x, sh.gs_pipelined(), sh.pipe_serialized()
            res = SH.run_local_ranks(3, work, dtype=F32)
            out[mode] = np.concatenate([r[0] for r in res])
            serialized = serialized or any(r[2] for r in res)    # (streams of two virtual ranks on one hardware queue: found at finalize, swept in turns)
            if not serialized:
                assert all(r[1] and r[1][0] for r in res), [r[1] for r in res]
    finally:
        for name, v in ((b"gs_bw", 1), (b"gs_bw_rows", 512)):
            lib.amgh_debug_set_tunable(name, v)
    assert out["exact"].dtype == F32 and np.array_equal(out["exact"], out["exact-turns"])
    xo, _, _ = O.OracleHierarchy(ml, dtype=F32).solve(b, maxiter=2, calculate_residual=False)
    assert rel(out["exact"], xo) <= F32_TOL
    if serialized:
        # (the library re-creates the ranks' sweep streams in rank order and probes again, three times, before it settles for turns:
        # with the 8 hardware queues tests/conftest.py asks for, 2-4 virtual ranks must end up side by side)
        msg = "the virtual ranks' streams shared a hardware queue in this process: swept in turns (oracle parity held)"
        if os.environ.get("GPU_MAX_HW_QUEUES") == "8":
            pytest.fail(msg + " although GPU_MAX_HW_QUEUES=8 was requested and the streams were re-created: the pipelined sweep did not run")
        pytest.skip(msg)


@pytest.mark.parametrize("bs", [1, 3, 4])
def test_float32_dataflow_sweeps_single_columns_and_blocks_of_right_hand_sides(bs):
    """The Float32 instance of the dataflow sweep (gs_flow.hpp with R = float: 8-byte mailboxes {value, epoch}, 4 values per
    16-byte chunk), forced onto a small hierarchy: levels really run it (mode 3), blocks of right-hand sides (one walker
    wave per column) equal their single columns bit for bit, and the cycle is the Float32 oracle's within Float32 rounding
    (multilevel.jl:28-59, smoother.jl:61-90)."""
    lib = AMG.hip_lib(F32)
    A = as_f32_matrix(AMG.poisson((24, 20, 16)))
    ml = AMG.ruge_stuben(A)
    n = ml.levels[0].A.m
    B = np.stack([uniform(n, 10 + c) - 0.3 * c for c in range(bs)], axis=1).astype(F32)
    for name, v in ((b"gs_bw", 2), (b"gs_bw_rows", 128), (b"gs_lpr", 1), (b"gs_ept", 1)):
        assert lib.amgh_debug_set_tunable(name, v) == 0
    try:
        dev1, devb = DeviceHierarchy(ml, 0, 1, dtype=F32), DeviceHierarchy(ml, 0, bs, dtype=F32)
        assert lib.amgh_debug_bw_mode(devb.h, 0) == 3 and lib.amgh_debug_bw_mode(dev1.h, 0) == 3
        Z = devb.precond_apply(B if bs > 1 else B[:, 0].copy())
        Z = Z.reshape(n, bs)
        singles = [dev1.precond_apply(B[:, c].copy()) for c in range(bs)]
        assert lib.amgh_debug_bw_poll_giveups(devb.h, 0) == 0
    finally:
        for name, v in ((b"gs_bw", 1), (b"gs_bw_rows", 512), (b"gs_lpr", 0), (b"gs_ept", 0)):
            lib.amgh_debug_set_tunable(name, v)
    oh = O.OracleHierarchy(ml, dtype=F32)
    for c in range(bs):
        assert np.array_equal(Z[:, c], singles[c]), c
    assert Z.dtype == F32 and rel(Z[:, bs - 1], oh.precond(np.ascontiguousarray(B[:, bs - 1]))) <= F32_TOL
