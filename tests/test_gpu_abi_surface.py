"""Entry points of include/amghip.h that no other GPU test calls directly: the reductions (amgh_dot_d, the norm
inside _solve), the TimerOutputs-style profile (amgh_profile_enable / _read), the HIP-event timer
(amgh_timer_begin / _end), hipGraph replay of whole cycles (amgh_set_use_graph), the sweep statistics
(amgh_gs_sweep_stats), and W / F cycles at a size where every sweep path (slots, long-row slots, chains, block
inverses) is in use."""
import ctypes as C
import math

import numpy as np
import pytest

import amg_amd as AMG
from conftest import uniform
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def rel(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y)) / max(np.linalg.norm(y), 1e-300)


@pytest.fixture(scope="module")
def h128():
    A = AMG.poisson((128, 128, 128))
    ml = AMG.ruge_stuben(A)
    return A, ml, ml.device()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 262144 + 17, 4_000_003])
def test_dot_and_norm_against_a_compensated_host_sum(n):
    """amgh_dot_d (two-pass wavefront reduction, fixed grid) vs math.fsum (exactly rounded): the pairwise-ish device
    sum must stay within 4 ulp * sqrt(n) of sum |x_i y_i| — and give the same bits on every call."""
    lib = AMG.hip_lib()
    x = uniform(n, 11) - 0.5
    y = uniform(n, 12) * 3.0 - 1.0
    xd, yd = AMG.DeviceBuffer(n, 0, x), AMG.DeviceBuffer(n, 0, y)
    scratch = AMG.DeviceBuffer(1100, 0)
    out = C.c_double(0)
    vals = []
    for _ in range(3):
        assert lib.amgh_dot_d(0, n, xd.ptr, yd.ptr, scratch.ptr, C.byref(out), None) == 0
        vals.append(out.value)
    assert vals[0] == vals[1] == vals[2]
    ref = math.fsum((x * y).tolist())
    bound = 4.0 * np.finfo(np.float64).eps * math.sqrt(n) * float(np.sum(np.abs(x * y)))
    assert abs(vals[0] - ref) <= bound
    assert lib.amgh_dot_d(0, n, xd.ptr, xd.ptr, scratch.ptr, C.byref(out), None) == 0
    ref2 = math.fsum((x * x).tolist())
    assert abs(out.value - ref2) <= 4.0 * np.finfo(np.float64).eps * math.sqrt(n) * ref2


def test_solve_history_is_the_norm_of_the_true_residual(h128):
    """norm(b) and norm(b - A x) of _solve! (multilevel.jl:170,190) against fsum-based norms of host vectors."""
    A, ml, dev = h128
    b = uniform(A.m, 3)
    x, hist = AMG._solve(ml, b, maxiter=2, reltol=1e-30, log=True)
    nb = math.sqrt(math.fsum((b * b).tolist()))
    assert abs(hist[0] - nb) <= 1e-13 * nb
    r = b - A.to_scipy() @ x
    nr = math.sqrt(math.fsum((r * r).tolist()))
    assert abs(hist[-1] - nr) <= 1e-9 * nr      # (the host SpMV rounds differently: 1e-9 on a residual 1e-1 below b)


def test_profile_labels_cover_the_cycle(h128):
    """The six TimerOutputs labels (multilevel.jl:180,216-236) per level: every level reports both smoothers, residual,
    restriction and prolongation; the coarse solve sits on the last level only; the labelled times add up to the
    HIP-event time of the whole cycle (within 5 %: what is not labelled is one memset per level)."""
    A, ml, dev = h128
    lib = dev.lib
    n = A.m
    bd, zd = AMG.DeviceBuffer(n, 0, uniform(n, 0)), AMG.DeviceBuffer(n, 0)
    for _ in range(2):
        assert lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0) == 0
    lib.amgh_dev_sync(0)
    dev.profile(True)
    dev.profile_read(reset=True)
    cycles = 3
    dev.timer_begin()
    for _ in range(cycles):
        assert lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0) == 0
    total_ms = dev.timer_end()
    prof = dev.profile_read()
    dev.profile(False)
    L = len(ml.levels)
    assert list(prof) == ["Presmoother", "Residual eval", "Restriction", "Coarse solve", "Prolongation", "Postsmoother"]
    for lab in ("Presmoother", "Residual eval", "Restriction", "Prolongation", "Postsmoother"):
        assert np.all(prof[lab][:L] > 0) and prof[lab][L] == 0
    assert np.all(prof["Coarse solve"][:L] == 0) and prof["Coarse solve"][L] > 0
    labelled = sum(float(v.sum()) for v in prof.values())
    assert abs(labelled - total_ms) <= 0.05 * total_ms, (labelled, total_ms)
    assert prof["Presmoother"][0] + prof["Postsmoother"][0] > 0.1 * labelled   # the fine-level sweeps dominate


def test_graph_replay_is_bitwise_the_eager_cycle(h128):
    A, ml, dev = h128
    lib = dev.lib
    n = A.m
    bd, zd = AMG.DeviceBuffer(n, 0, uniform(n, 4)), AMG.DeviceBuffer(n, 0)
    assert lib.amgh_set_use_graph(dev.h, 0) == 0
    assert lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0) == 0
    lib.amgh_dev_sync(0)
    eager = zd.download()
    assert lib.amgh_set_use_graph(dev.h, 1) == 0
    got = []
    for _ in range(4):      # eager warm-up, capture + first replay, replays
        assert lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0) == 0
        lib.amgh_dev_sync(0)
        got.append(zd.download())
    assert lib.amgh_set_use_graph(dev.h, 0) == 0
    for z in got:
        assert np.array_equal(z, eager)
    # a whole solve through cached graphs (cycle + residual per iteration)
    b = uniform(n, 5)
    x0, h0 = AMG._solve(ml, b, reltol=1e-6, log=True)
    lib.amgh_set_use_graph(dev.h, 1)
    x1, h1 = AMG._solve(ml, b, reltol=1e-6, log=True)
    lib.amgh_set_use_graph(dev.h, 0)
    assert np.array_equal(x0, x1) and np.array_equal(h0, h1)


def test_sweep_stats_describe_the_schedule(h128):
    A, ml, dev0 = h128
    from amg_amd.device import DeviceHierarchy
    lib = AMG.hip_lib()
    lib.amgh_debug_set_tunable(b"gs_bw", 0)          # the merged dependency-level groups (what levels below ~1.5 M rows get)
    try:
        dev = DeviceHierarchy(ml, 0, 1)
    finally:
        lib.amgh_debug_set_tunable(b"gs_bw", 1)
    st = dev.gs_sweep_stats(0, False)
    assert st["rows"] == A.m and 1 <= st["launches"] <= dev.gs_sweep_steps(0, False)   # narrow groups chain into one launch
    assert st["entries"] > 0 and st["slot_entries"] >= 0
    if st["levels_per_group"] > 1:      # merged groups: composite rows + the other triangle as a pre-pass
        assert st["tri_entries"] > 0 and st["entries"] >= (A.nnz - A.m) // 2
    assert dev.gs_dependency_levels(0) == 3 * 128 - 2
    # the same level as a wavefront of blocks (what single-column hierarchies get from 1.5 M 7-point rows on — so the
    # fixture's own fine level —, from 3 M rows otherwise): the operator's own
    # off-diagonal entries — no composite rows, no pre-pass —, 3 * 16 - 2 depths of the quotient graph for 16^3 blocks of
    # 8^3 rows, executed as ONE launch — a dataflow (rows published as they are computed: gs_bw_flow, the default), or the
    # blocks chained by flags (gs_bw_flow = 0) — or, gs_bw_chain = 0 as well, as a launch per depth.  The `full` footprint
    # keeps all three layouts (the default keeps the dataflow layout only)
    lib.amgh_debug_set_tunable(b"gs_bw", 2); lib.amgh_debug_set_tunable(b"gs_lean", 0)
    try:
        devb = DeviceHierarchy(ml, 0, 1)
    finally:
        lib.amgh_debug_set_tunable(b"gs_bw", 1); lib.amgh_debug_set_tunable(b"gs_lean", -1)
    sb = devb.gs_sweep_stats(0, False)
    assert sb["tri_entries"] == 0 and sb["entries"] == A.nnz - A.m and sb["slot_entries"] == 6 * A.m
    assert sb["launches"] == 1 and devb.gs_sweep_steps(0, False) == 46 and devb.gs_dependency_levels(0) == 3 * 128 - 2
    b = uniform(A.m, 77)
    zb = devb.precond_apply(b)
    assert rel(zb, dev.precond_apply(b)) <= 1e-12
    lib.amgh_debug_set_tunable(b"gs_bw_flow", 0)
    try:
        assert devb.gs_sweep_stats(0, False)["launches"] == 1
        assert np.array_equal(devb.precond_apply(b), zb)        # chained by flags: the same sweep bit for bit
        lib.amgh_debug_set_tunable(b"gs_bw_chain", 0)
        assert devb.gs_sweep_stats(0, False)["launches"] == 46
        assert np.array_equal(devb.precond_apply(b), zb)        # a launch per depth: the same sweep bit for bit
        assert dev0.gs_sweep_stats(0, False)["launches"] == 1   # (the default footprint holds the dataflow layout only)
        assert np.array_equal(dev0.precond_apply(b), dev0.precond_apply(b))
    finally:
        lib.amgh_debug_set_tunable(b"gs_bw_chain", 1); lib.amgh_debug_set_tunable(b"gs_bw_flow", 1)
    assert lib.amgh_debug_bw_poll_giveups(devb.h, 0) == 0 and lib.amgh_debug_bw_poll_giveups(dev.h, 0) == -1
    # (the default at this size: the fine level as blocks, the 1.0 M-row second level on merged groups — forced above)
    assert dev0.gs_sweep_stats(0, False) == sb and rel(dev0.precond_apply(b), zb) <= 1e-12
    # blocks of 10^3 rows: 98 KB of LDS per workgroup (beyond the 64 KB a kernel gets without asking), another partition,
    # the same sweep bit for bit
    lib.amgh_debug_set_tunable(b"gs_bw", 2); lib.amgh_debug_set_tunable(b"gs_bw_rows", 1000)
    try:
        devc = DeviceHierarchy(ml, 0, 1)
    finally:
        lib.amgh_debug_set_tunable(b"gs_bw", 1); lib.amgh_debug_set_tunable(b"gs_bw_rows", 512)
    assert devc.gs_sweep_steps(0, False) == 3 * 13 - 2 and np.array_equal(devc.precond_apply(b), zb)
    assert lib.amgh_debug_bw_poll_giveups(devc.h, 0) == 0


@pytest.mark.parametrize("cyc", [1, 2])
def test_w_and_f_cycles_at_128_cubed(h128, cyc):
    A, ml, dev = h128
    b = uniform(A.m, 6)
    oh = O.OracleHierarchy(ml)
    cycle = {1: AMG.W(), 2: AMG.F()}[cyc]
    x, hist = AMG._solve(ml, b, cycle, maxiter=3, reltol=1e-30, log=True)
    xo, ho, _ = oh.solve(b, cycle=cyc, maxiter=3, reltol=1e-30)
    assert len(hist) == len(ho) == 4 and np.allclose(hist, ho, rtol=1e-8)
    assert rel(x, xo) <= 1e-10


def test_tiny_operators_swept_out_of_lds_and_small_hierarchies_replayed_from_graphs():
    """An operator that fits LDS entirely (rows, nonzeros, level descriptors, x, b) is swept out of LDS: by ONE wave
    walking its record without any barrier (gs_wave_kernel, gs_tiny = 1: rows of at most 24 off-diagonal entries, a
    symmetric sweep in one launch) or by gs_chain_tiny_kernel (gs_tiny = 2) — both with the products and in-order row
    sums of gs_chain_kernel (gs_tiny = 0): bitwise the same sweep.  A small hierarchy replayed from a hipGraph (opt-in)
    is bitwise the eager cycle, and PCG on the reference's lin_elastic_2d configuration (nns_test.jl:213-226) keeps
    its 13 iterations."""
    from amg_amd.device import DeviceHierarchy
    from conftest import load_csc, load_npz
    lib = AMG.hip_lib()
    d = load_npz("lin_elastic_2d")
    A = load_csc("lin_elastic_2d")
    x0, b = uniform(A.m, 3), uniform(A.m, 4)
    # one stand-alone sweep each way, GS and SOR, both kernels
    out = {}
    for tiny in (1, 2, 0):
        assert lib.amgh_debug_set_tunable(b"gs_tiny", tiny) == 0
        try:
            for cfg in (AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(iter=2),
                        AMG.SOR(1.3, AMG.SymmetricSweep()), AMG.SOR(0.8, AMG.ForwardSweep(), iter=3)):
                xs = x0.copy()
                AMG.device.smooth_standalone(cfg, A, xs, b)       # in place
                out[(tiny, repr((type(cfg).__name__, cfg.sweep_code, cfg.iter, cfg.omega)))] = xs
        finally:
            lib.amgh_debug_set_tunable(b"gs_tiny", 1)
    for (tiny, key), v in out.items():
        if tiny != 0:
            assert np.array_equal(v, out[(0, key)]), (tiny, key)
    # a record beyond the default 64 KB of dynamic LDS (900 rows of a 5-point operator + b + x)
    A2 = AMG.poisson((30, 30))
    x2, b2 = uniform(A2.m, 5), uniform(A2.m, 6)
    big = {}
    for tiny in (1, 0):
        lib.amgh_debug_set_tunable(b"gs_tiny", tiny)
        try:
            xs = x2.copy()
            AMG.device.smooth_standalone(AMG.GaussSeidel(iter=2), A2, xs, b2)
            big[tiny] = xs
        finally:
            lib.amgh_debug_set_tunable(b"gs_tiny", 1)
    assert np.array_equal(big[1], big[0]) and rel(big[1], O.smooth(AMG.GaussSeidel(iter=2), A2, x2, b2, hermitian=True)) <= 1e-13
    # the hierarchy: default (graph replay decided by amgh_finalize) vs eager, tiny kernel vs regular chain
    ml = AMG.smoothed_aggregation(A, B=d["B"])
    auto = DeviceHierarchy(ml, 0, 1)
    assert lib.amgh_set_use_graph(auto.h, 1) == 0      # (opt-in: measured no faster than eager launches)
    eager = DeviceHierarchy(ml, 0, 1)
    assert lib.amgh_set_use_graph(eager.h, 0) == 0
    z_e = eager.precond_apply(d["b"])
    for _ in range(4):      # eager warm-up, capture + first replay, replays
        assert np.array_equal(auto.precond_apply(d["b"]), z_e)
    for tiny in (0, 2):
        lib.amgh_debug_set_tunable(b"gs_tiny", tiny)
        try:
            assert np.array_equal(DeviceHierarchy(ml, 0, 1).precond_apply(d["b"]), z_e)
        finally:
            lib.amgh_debug_set_tunable(b"gs_tiny", 1)
    # a block of right-hand sides: one workgroup (one walking wave) per column, bitwise the single columns
    Bk = np.stack([d["b"], uniform(A.m, 8), -d["b"]], axis=1)
    Zk = DeviceHierarchy(ml, 0, 3).precond_apply(Bk)
    for q in range(3):
        assert np.array_equal(Zk[:, q], eager.precond_apply(np.ascontiguousarray(Bk[:, q])))
    assert rel(z_e, O.OracleHierarchy(ml).precond(d["b"])) <= 1e-10
    xp, log = AMG.cg(A, d["b"], Pl=AMG.aspreconditioner(ml), reltol=1e-10, log=True)
    xpo, _, itp = O.OracleHierarchy(ml).pcg(d["b"], reltol=1e-10)
    assert log["iters"] == itp == 13 and rel(xp, xpo) <= 1e-9


def test_wavefront_of_blocks_schedule_vs_oracle_and_level_schedules():
    """Single-right-hand-side hierarchies may sweep stencil-like fine levels as a WAVEFRONT OF BLOCKS (gs_blocks.hpp:
    acyclic block partition from monotone potentials, one wave walking each block, the scalar loop's arithmetic).
    Forced here on small operators (gs_bw = 2, blocks of ~64 rows): 3-D / 2-D / 1-D Poisson, an irregular Galerkin
    operator, symmetric and directional Gauss-Seidel and SOR, V / W / F cycles — against the oracle, and against the
    same hierarchy on the level schedules (gs_bw = 0).  A block of right-hand sides keeps the level schedules."""
    from amg_amd.device import DeviceHierarchy
    lib = AMG.hip_lib()
    cases = [AMG.poisson((20, 18, 16)), AMG.poisson((48, 40)), AMG.poisson(700)]
    cases.append(AMG.ruge_stuben(AMG.poisson((24, 24, 24))).levels[1].A)          # 19-point-like rows: too long -> not eligible, level schedules
    smoothers = [(AMG.GaussSeidel(), AMG.GaussSeidel()), (AMG.GaussSeidel(AMG.ForwardSweep(), iter=2), AMG.GaussSeidel(AMG.BackwardSweep())),
                 (AMG.SOR(1.2), AMG.SOR(0.8, AMG.ForwardSweep()))]
    used = 0
    for A in cases:
        b = uniform(A.m, 17) - 0.3
        for pre, post in smoothers:
            ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=post)
            oh = O.OracleHierarchy(ml)
            out = {}
            for mode in (2, 0):
                assert lib.amgh_debug_set_tunable(b"gs_bw", mode) == 0
                assert lib.amgh_debug_set_tunable(b"gs_bw_rows", 64) == 0
                try:
                    dev = DeviceHierarchy(ml, 0, 1)
                    if mode == 2:
                        st0 = dev.gs_sweep_stats(0, False)
                        used += bool(dev.gs_sweep_steps(0, False) < dev.gs_dependency_levels(0) and st0["tri_entries"] == 0 and st0["levels_per_group"] >= 1 and st0["slot_entries"] == A.m * 6)
                    for cyc in (0, 1, 2):
                        out[(mode, cyc)] = dev.precond_apply(b, cyc)
                finally:
                    lib.amgh_debug_set_tunable(b"gs_bw", 1)
                    lib.amgh_debug_set_tunable(b"gs_bw_rows", 512)
            for cyc in (0, 1, 2):
                ref = oh.precond(b, cycle=cyc)
                assert rel(out[(2, cyc)], ref) <= 1e-10, (A.m, repr(pre), cyc)
                assert rel(out[(2, cyc)], out[(0, cyc)]) <= 1e-12, (A.m, repr(pre), cyc)
    assert used >= 6    # the three Poisson operators really ran as wavefronts of blocks
    # the device walk is the host execution of the same records — and both the scalar loop — bit for bit
    A = cases[0]
    rp, ci, va = A.csr_arrays()
    x0, bb = uniform(A.m, 21) - 0.5, uniform(A.m, 22)
    for pre in (AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(iter=2)):
        ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=pre)
        lib.amgh_debug_set_tunable(b"gs_bw", 2); lib.amgh_debug_set_tunable(b"gs_bw_rows", 64); lib.amgh_debug_set_tunable(b"gs_lean", 0)
        try:
            dev = DeviceHierarchy(ml, 0, 1)
            x_dev = dev.smooth(0, False, x0, bb)
            # ONE launch per sweep — as a dataflow (rows published as they are computed, the default), or the blocks chained by
            # flags (gs_bw_flow = 0) — and one launch per depth of the quotient graph (gs_bw_chain = 0 as well): the same bits
            assert dev.gs_sweep_stats(0, False)["launches"] == 1 and lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0
            for rep in range(3):      # (epochs of the mailboxes: repeated sweeps on the same schedule)
                assert np.array_equal(dev.smooth(0, False, x0, bb), x_dev)
            assert lib.amgh_debug_set_tunable(b"gs_bw_flow", 0) == 0
            assert dev.gs_sweep_stats(0, False)["launches"] == 1
            for rep in range(3):      # (epochs of the flags)
                assert np.array_equal(dev.smooth(0, False, x0, bb), x_dev), repr(pre)
            assert lib.amgh_debug_set_tunable(b"gs_bw_chain", 0) == 0
            assert dev.gs_sweep_stats(0, False)["launches"] == dev.gs_sweep_steps(0, False) > 1
            assert np.array_equal(dev.smooth(0, False, x0, bb), x_dev), repr(pre)
            assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0
        finally:
            lib.amgh_debug_set_tunable(b"gs_bw", 1); lib.amgh_debug_set_tunable(b"gs_bw_rows", 512); lib.amgh_debug_set_tunable(b"gs_bw_chain", 1)
            lib.amgh_debug_set_tunable(b"gs_bw_flow", 1); lib.amgh_debug_set_tunable(b"gs_lean", -1)
        assert dev.gs_sweep_stats(0, False)["slot_entries"] == 6 * A.m
        assert np.array_equal(x_dev, O.smooth(pre, A, x0, bb, hermitian=True)), repr(pre)
        xh = x0.copy()
        for it in range(pre.iter):
            for back in ((0,), (1,), (0, 1))[pre.sweep_code]:
                st = np.zeros(4, dtype=np.int64)
                rc = lib.amgh_debug_bw_sweep_host(A.m, np.ascontiguousarray(rp, np.int32).ctypes.data, np.ascontiguousarray(ci, np.int32).ctypes.data,
                                                  np.ascontiguousarray(va).ctypes.data, 64, back, 1.0, xh.ctypes.data, bb.ctypes.data, st.ctypes.data)
                assert rc == 0
        assert np.array_equal(x_dev, xh), repr(pre)
    # nrhs > 1: never the single-column layout
    A = cases[0]
    ml = AMG.ruge_stuben(A)
    lib.amgh_debug_set_tunable(b"gs_bw", 2)
    try:
        dev3 = DeviceHierarchy(ml, 0, 3)
        B = np.stack([uniform(A.m, 1), uniform(A.m, 2), uniform(A.m, 3)], axis=1)
        Z = dev3.precond_apply(B)
    finally:
        lib.amgh_debug_set_tunable(b"gs_bw", 1)
    oh = O.OracleHierarchy(ml)
    for q in range(3):
        assert rel(Z[:, q], oh.precond(np.ascontiguousarray(B[:, q]))) <= 1e-10


def test_chained_blocks_on_random_patterns_are_the_host_plan_bit_for_bit():
    """The block layout on operators that are NOT grids: random sparsity, structurally non-symmetric (a block then reads
    old values of blocks that never read it back — the far / near split of its external positions and its predecessor /
    successor lists are all that orders the chained kernel; the dataflow sweep, whose only order is the data it waits for,
    declines such a pattern) and symmetric with zero diagonals (rows the sweep skips — and still publishes).
    Forward, backward and symmetric sweeps on the device — dataflow, chained, one launch per depth — against the
    host execution of the same plan (amgh_debug_bw_sweep_host: bitwise the scalar loop, tests/test_bw_host.py)."""
    from amg_amd.device import DeviceHierarchy
    from test_bw_host import _short_rows
    lib = AMG.hip_lib()
    for A, herm in ((_short_rows(6000, 6, False), False), (_short_rows(5000, 5, True, zero_diag=(0, 17, 4999)), True)):
        rp, ci, va = AMG.device.smoother_matrix_csr(A, None)      # what a HermitianSymmetry hierarchy sweeps: the CSC arrays read as rows
        rp, ci, va = np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(ci, np.int32), np.ascontiguousarray(va)
        x0, bb = uniform(A.m, 41) - 0.5, uniform(A.m, 42)
        for pre in (AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(iter=2)):
            ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=pre)
            lib.amgh_debug_set_tunable(b"gs_bw", 2); lib.amgh_debug_set_tunable(b"gs_bw_rows", 64); lib.amgh_debug_set_tunable(b"gs_lean", 0)
            try:
                dev = DeviceHierarchy(ml, 0, 1)
                st = dev.gs_sweep_stats(0, False)
                assert st["launches"] == 1 and st["tri_entries"] == 0 and dev.gs_sweep_steps(0, False) > 1     # really the block layout
                x_dev = dev.smooth(0, False, x0, bb)          # (the symmetric pattern: as a dataflow; the non-symmetric one: chained by flags)
                lib.amgh_debug_set_tunable(b"gs_bw_flow", 0)
                x_chained = dev.smooth(0, False, x0, bb)
                lib.amgh_debug_set_tunable(b"gs_bw_chain", 0)
                x_launched = dev.smooth(0, False, x0, bb)
            finally:
                lib.amgh_debug_set_tunable(b"gs_bw", 1); lib.amgh_debug_set_tunable(b"gs_bw_rows", 512); lib.amgh_debug_set_tunable(b"gs_bw_chain", 1)
                lib.amgh_debug_set_tunable(b"gs_bw_flow", 1); lib.amgh_debug_set_tunable(b"gs_lean", -1)
            assert np.array_equal(x_chained, x_launched)
            xh = x0.copy()
            for it in range(pre.iter):
                for back in ((0,), (1,), (0, 1))[pre.sweep_code]:
                    rc = lib.amgh_debug_bw_sweep_host(A.m, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, 64, back, 1.0, xh.ctypes.data, bb.ctypes.data, None)
                    assert rc == 0
            assert np.array_equal(x_dev, xh) and np.array_equal(x_launched, xh), (herm, repr(pre))
            assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0


def test_jacobi_on_a_zero_vector_skips_the_matrix_pass_bitwise():
    """Every pre-smoother below the fine level of a cycle (and the fine one of ldiv!) starts from x = 0: the damped Jacobi
    sweep is then x = (1 - w) 0 + w ((b - 0) / d), evaluated by a vector kernel instead of a pass over the matrix —
    the same expression with the same operands: bitwise the full sweep, V / W / F, blocks of right-hand sides, w > 1."""
    from amg_amd.device import DeviceHierarchy
    lib = AMG.hip_lib()
    A = AMG.poisson((48, 40))
    b = uniform(A.m, 12) - 0.4
    B = np.stack([b, uniform(A.m, 13), -b], axis=1)
    for omega, iters in ((2.0 / 3.0, 1), (1.25, 2)):
        jac = AMG.Jacobi(omega, iter=iters)
        ml = AMG.smoothed_aggregation(A, presmoother=jac, postsmoother=jac)
        out = {}
        for flag in (1, 0):
            assert lib.amgh_debug_set_tunable(b"jacobi_zero", flag) == 0
            try:
                dev1, dev3 = DeviceHierarchy(ml, 0, 1), DeviceHierarchy(ml, 0, 3)
                for cyc in (0, 1, 2):
                    out[(flag, cyc, 1)] = dev1.precond_apply(b, cyc)
                    out[(flag, cyc, 3)] = dev3.precond_apply(B, cyc)
            finally:
                lib.amgh_debug_set_tunable(b"jacobi_zero", 1)
        for (flag, cyc, bs), v in out.items():
            if flag == 1:
                assert np.array_equal(v, out[(0, cyc, bs)]), (omega, cyc, bs)
        assert rel(out[(1, 0, 1)], O.OracleHierarchy(ml).precond(b)) <= 1e-10


def test_footprint_policies_are_bitwise_the_same_hierarchy():
    """AMGH_LEAN / tunable gs_lean — full (0: every copy kept), trim (2, the default: no un-merged slot copy, no CSR
    copy of slotted composite rows once the SELL-like build has read it, no natural-order P / R / coarse A where the
    cycle runs level-ordered, backward pre-pass triangle on first use), lean (1: trim + compaction at layout time, no
    SELL-like copies): same kernels on the same numbers, fewer bytes."""
    from amg_amd.device import DeviceHierarchy
    lib = AMG.hip_lib()
    A = AMG.poisson((64, 64, 64))
    ml = AMG.ruge_stuben(A)
    b = uniform(A.m, 8)
    try:
        assert lib.amgh_debug_set_tunable(b"gs_lean", 0) == 0
        full = DeviceHierarchy(ml, 0, 1)
        assert lib.amgh_debug_set_tunable(b"gs_lean", -1) == 0     # the default policy
        trim = DeviceHierarchy(ml, 0, 1)
        # (full and trim carry a SELL-like copy of long-row groups whose row sums add in another order; the lean one has
        #  none: compare like with like)
        assert lib.amgh_debug_set_tunable(b"gs_sell", 0) == 0
        assert lib.amgh_debug_set_tunable(b"gs_lean", 0) == 0
        full_nosell = DeviceHierarchy(ml, 0, 1)
        assert lib.amgh_debug_set_tunable(b"gs_lean", 1) == 0
        lean = DeviceHierarchy(ml, 0, 1)
    finally:
        lib.amgh_debug_set_tunable(b"gs_lean", -1)
        lib.amgh_debug_set_tunable(b"gs_sell", 1)
    for a, c in ((full, trim), (full_nosell, lean)):
        if a is full_nosell:
            lib.amgh_debug_set_tunable(b"gs_sell", 0)
        try:
            assert np.array_equal(a.precond_apply(b), c.precond_apply(b))
            xa, ha, _ = a.solve(b, np.zeros_like(b), 0, 30, 0.0, 1e-8, True, True)
            xc, hc, _ = c.solve(b, np.zeros_like(b), 0, 30, 0.0, 1e-8, True, True)
            assert np.array_equal(xa, xc) and np.array_equal(ha, hc)
        finally:
            lib.amgh_debug_set_tunable(b"gs_sell", 1)
    df, dt, dl = full.device_bytes_detail(), trim.device_bytes_detail(), lean.device_bytes_detail()
    for d in (dt, dl):
        assert d["unmerged_slots"] < df["unmerged_slots"] and d["natural_APR"] < df["natural_APR"]
        assert d["prepass_triangles"] < 0.6 * df["prepass_triangles"]      # symmetric sweeps never ask for the backward one
    assert dl["merged_csr"] < 0.2 * df["merged_csr"] and dt["merged_csr"] < 0.5 * df["merged_csr"]
    assert trim.device_bytes() < 0.75 * full.device_bytes() and lean.device_bytes() <= trim.device_bytes()
    for dev, d in ((full, df), (trim, dt)):
        assert abs(sum(d.values()) - dev.device_bytes()) <= 0.02 * dev.device_bytes()   # the categories add up
    # the stand-alone hooks of a released natural-order operator go through the level-ordered copy: the same products
    for lvl in (0, 1):
        n, nc = ml.levels[lvl].A.m, ml.levels[lvl].P.n
        xn, xc = uniform(n, 20 + lvl), uniform(nc, 30 + lvl)
        for dev in (trim, lean):
            assert np.array_equal(dev.spmv(lvl, 0, xn), full.spmv(lvl, 0, xn))      # A
            assert np.array_equal(dev.spmv(lvl, 1, xc), full.spmv(lvl, 1, xc))      # P
            assert np.array_equal(dev.spmv(lvl, 2, xn), full.spmv(lvl, 2, xn))      # R
    # a smoother that STARTS backward builds the backward pre-pass triangle on first use
    x0 = uniform(A.m, 9)
    ml_b = AMG.ruge_stuben(A, presmoother=AMG.GaussSeidel(AMG.BackwardSweep()), postsmoother=AMG.GaussSeidel(AMG.BackwardSweep()))
    zb = DeviceHierarchy(ml_b, 0, 1).precond_apply(x0)
    lib.amgh_debug_set_tunable(b"gs_lean", 0)
    try:
        zb_full = DeviceHierarchy(ml_b, 0, 1).precond_apply(x0)
    finally:
        lib.amgh_debug_set_tunable(b"gs_lean", -1)
    assert np.array_equal(zb, zb_full)
    from oracle import oracle as O
    assert rel(zb, O.OracleHierarchy(ml_b).precond(x0)) <= 1e-10


@pytest.mark.parametrize("case", ["poisson3d", "rs_coarse", "irregular"])
def test_device_built_merged_groups_equal_the_host_construction_bit_for_bit(case, monkeypatch):
    """The composite rows of the merged dependency-level groups are built on the device (gs_merge_dev.hpp); the host
    construction (AMGH_HOST_MERGE=1) is its reference: same grouping decisions aside, a sweep through either must
    give the same bits, because every coefficient is accumulated in the same order."""
    from amg_amd.device import DeviceCSR
    if case == "poisson3d":
        A = AMG.poisson((40, 40, 40))
    elif case == "rs_coarse":
        A = AMG.ruge_stuben(AMG.poisson((48, 48, 48)), max_levels=3).levels[1].A     # a 19-point-like Galerkin operator
    else:
        import scipy.sparse as sp
        rng = np.random.default_rng(5)
        n = 6000
        B = sp.random(n, n, density=4.0 / n, random_state=7, format="csr")
        B = B + B.T + sp.diags(np.arange(n) % 5 + 8.0)
        B = B + sp.diags([-0.3 * np.ones(n - 1), -0.3 * np.ones(n - 1)], [-1, 1])
        A = AMG.SparseMatrixCSC.from_scipy(B.tocsc())
    rp, ci, va = A.colptr, A.rowval, A.nzval      # column i swept as row i (smoother.jl:81-86)
    n = A.m
    x0, b = uniform(n, 3) - 0.5, uniform(n, 4)
    gs = AMG.GaussSeidel(AMG.SymmetricSweep(), 2)
    lib = AMG.hip_lib()
    out = {}
    # groups of exactly 2 levels on both sides (the host path chooses from a sampled estimate, the device path from
    # exact counts: with more candidates they may pick different group sizes), and no SELL-like copy (device path only)
    assert lib.amgh_debug_set_tunable(b"gs_merge", 2) == 0 and lib.amgh_debug_set_tunable(b"gs_sell", 0) == 0
    try:
        for mode in ("device", "host"):
            if mode == "host":
                monkeypatch.setenv("AMGH_HOST_MERGE", "1")
            else:
                monkeypatch.delenv("AMGH_HOST_MERGE", raising=False)
            op = DeviceCSR(n, n, rp, ci, va)
            assert lib.amgh_csr_prepare(op.h, 0, 1) == 0
            out[mode] = op.smooth(gs, x0.copy(), b)
    finally:
        lib.amgh_debug_set_tunable(b"gs_merge", 16)
        lib.amgh_debug_set_tunable(b"gs_sell", 1)
    assert np.array_equal(out["device"], out["host"])
    # and both are the lexicographic sweep
    xo = O.smooth(gs, A, x0, b)
    assert rel(out["device"], xo) <= 1e-11


@pytest.mark.parametrize("alg", ["superlu", "dense_lu"])
def test_linear_solve_wrapper_coarse_solver(alg):
    """coarse_solver = LinearSolveWrapper(alg) (coarse_solver.jl:24-58): a third-party factorisation set up once and
    called back by the device cycle through amgh_set_coarse_host for every coarse solve — V and W cycles, one
    right-hand side and a block (the wrapper loops the columns, coarse_solver.jl:36-41)."""
    A = AMG.poisson((24, 24, 24))
    algo = AMG.SuperLUFactorization() if alg == "superlu" else AMG.DenseLUFactorization()
    ml = AMG.ruge_stuben(A, max_levels=3, coarse_solver=AMG.LinearSolveWrapper(algo))   # a coarsest level of ~600 rows
    assert ml.final_A.m > 100 and not ml.coarse_solver.uses_dense()
    ref = AMG.ruge_stuben(A, max_levels=3)                                              # default QRSolver: dense operator on device
    b = uniform(A.m, 12)
    oh = O.OracleHierarchy(ml)
    for cyc, cycle in ((0, AMG.V()), (1, AMG.W())):
        x, hist = AMG._solve(ml, b, cycle, reltol=1e-9, log=True)
        xo, ho, _ = oh.solve(b, cycle=cyc, reltol=1e-9)
        assert len(hist) == len(ho) and rel(x, xo) <= 1e-10
        xr, hr = AMG._solve(ref, b, cycle, reltol=1e-9, log=True)
        assert len(hr) == len(hist) and rel(x, xr) <= 1e-9
    B = np.column_stack([b, uniform(A.m, 13), uniform(A.m, 14) - 0.5])
    X = AMG._solve(ml, B, reltol=1e-9)
    for c in range(3):
        assert rel(X[:, c], AMG._solve(ml, B[:, c].copy(), reltol=1e-9)) <= 1e-12
    z = AMG.aspreconditioner(ml).ldiv(b)
    assert rel(z, oh.precond(b)) <= 1e-10


def test_precs_builders_and_commonsolve_split():
    """LinearSolve `precs` builders (precs.jl:7-38): `(builder)(A, p) -> (Pl, I)` with Pl usable as `cg(A, b; Pl)`;
    CommonSolve `init` + `solve!` (multilevel.jl:252-264) = `solve`."""
    A = AMG.poisson((40, 40))
    b = uniform(A.m, 21)
    for builder, setup in ((AMG.RugeStubenPreconBuilder(), AMG.ruge_stuben),
                           (AMG.SmoothedAggregationPreconBuilder(), AMG.smoothed_aggregation),
                           (AMG.RugeStubenPreconBuilder(blocksize=1, max_levels=3), lambda M: AMG.ruge_stuben(M, max_levels=3))):
        Pl, Pr = builder(A, None)
        assert isinstance(Pl, AMG.Preconditioner) and isinstance(Pr, AMG.Identity)
        assert np.array_equal(Pr.ldiv(b), b)
        x, info = AMG.cg(A, b, Pl=Pl, reltol=1e-10, log=True)
        oh = O.OracleHierarchy(setup(A))
        xo, ho, it = oh.pcg(b, reltol=1e-10)
        assert info["iters"] == it and rel(x, xo) <= 1e-9
    # a scipy matrix goes in like an AbstractSparseMatrixCSC
    Pl2, _ = AMG.RugeStubenPreconBuilder()(A.to_scipy(), None)
    assert rel(Pl2.ldiv(b), AMG.aspreconditioner(AMG.ruge_stuben(A)).ldiv(b)) == 0.0
    # init / solve!
    solt = AMG.init(AMG.RugeStubenAMG(), A, b, max_levels=4)
    assert isinstance(solt, AMG.AMGSolver) and len(solt.ml) == 4
    x1 = AMG.solve_(solt, reltol=1e-9)
    x2 = AMG.solve(A, b, AMG.RugeStubenAMG(), max_levels=4, reltol=1e-9)
    assert np.array_equal(x1, x2)
