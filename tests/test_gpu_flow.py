"""The wavefront of blocks as a DATAFLOW (csrc/hip/gs_flow.hpp): rows published into mailboxes the moment they are
computed, blocks starting on finished faces, the packed rows streamed straight into registers.  Same plan, same
arithmetic as the chained and launched sweeps of gs_blocks.hpp — and so the scalar lexicographic loop of
/root/reference/src/smoother.jl:61-90 (gs!) and :193-221 (sor_step!) BIT FOR BIT: against the oracle's loops, against
the host execution of the plan (amgh_debug_bw_sweep_host), against the two other executions on the device.  Plus what a
protocol built on polling owes its callers: a wait that gives up is an error code, never numbers (a forced protocol
error), and a soak under memory load, eager and graph-replayed."""
import ctypes
import threading

import numpy as np
import pytest

import amg_amd as AMG
from amg_amd.device import DeviceHierarchy
from conftest import uniform
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def rel(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y)) / max(np.linalg.norm(y), 1e-300)


class tunables:
    """amgh_debug_set_tunable for the duration of a with block (values restored to the given defaults)."""
    DEFAULTS = {"gs_bw": 1, "gs_bw_rows": 512, "gs_bw_chain": 1, "gs_bw_flow": 1, "gs_lean": -1, "gs_bw_spin": 0, "gs_bw_skip_pub": -1,
                "gs_lpr": 0, "gs_ept": 0, "gs_bw_nc": -1, "gs_bw_nrhs": 1, "gs_flow_xzero": 1, "gs_bw_relay": 3, "gs_bw_dict": 1,
                "gs_bw_inorder": 1}   # (the suite's default, tests/conftest.py; the library's is 0)

    def __init__(self, lib, **kw):
        self.lib, self.kw = lib, kw

    def __enter__(self):
        for k, v in self.kw.items():
            assert self.lib.amgh_debug_set_tunable(k.encode(), int(v)) == 0
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            self.lib.amgh_debug_set_tunable(k.encode(), self.DEFAULTS[k])


def _host_plan_sweeps(lib, A, pre, x0, bb, rows):
    rp, ci, va = AMG.device.smoother_matrix_csr(A, None)
    rp, ci, va = np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(ci, np.int32), np.ascontiguousarray(va)
    xh = x0.copy()
    omega = getattr(pre, "omega", 1.0)
    for _ in range(pre.iter):
        for back in ((0,), (1,), (0, 1))[pre.sweep_code]:
            rc = lib.amgh_debug_bw_sweep_host(A.m, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, rows, back, float(omega), xh.ctypes.data, bb.ctypes.data, None)
            assert rc == 0
    return xh


def _irregular_long_rows(n, seed, cap=18, tries=14):
    """random structurally symmetric operator whose rows hold up to `cap` off-diagonal entries (the 18-entry records): an edge is
    kept only while both of its rows have room; mostly short-range edges + a few long ones (blocks with many external columns)"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    deg = np.zeros(n, dtype=np.int64)
    rows, cols, vals = [], [], []
    seen = set()
    for i in range(n):
        for t in range(tries):
            j = int(i + rng.integers(-40, 41)) if t < tries - 2 else int(rng.integers(0, n))
            if j < 0 or j >= n or j == i or (min(i, j), max(i, j)) in seen or deg[i] >= cap or deg[j] >= cap:
                continue
            seen.add((min(i, j), max(i, j)))
            v = -float(rng.random()) - 0.1
            rows += [i, j]; cols += [j, i]; vals += [v, v * (0.5 + float(rng.random()))]   # (values not symmetric: only the pattern has to be)
            deg[i] += 1; deg[j] += 1
    M = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    d = np.asarray(abs(M).sum(axis=1)).ravel() + 1.0
    K = (M + sp.diags(d)).tocsc()
    assert int(np.diff(K.tocsr().indptr).max()) - 1 > 12          # really rows of the 18-entry kernels
    return AMG.SparseMatrixCSC.from_scipy(K)


def test_relayed_sweep_on_irregular_long_rows_is_the_scalar_loop_bit_for_bit():
    """Rows of up to 18 entries on an irregular (random, structurally symmetric, unsymmetric values) pattern: the relayed dataflow
    sweep of the 18-entry records = the single walker = the chained and launched kernels = the host plan = the oracle's scalar
    loop, bitwise — forward, backward, symmetric; SOR at 1e-13.  (The S matrix path: the smoother sweeps the transpose's rows.)"""
    lib = AMG.hip_lib()
    A = _irregular_long_rows(6000, 3)
    x0, bb = uniform(A.m, 41) - 0.5, uniform(A.m, 42)
    for pre in (AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(iter=2), AMG.SOR(1.1)):
        ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=pre)
        with tunables(lib, gs_bw=2, gs_bw_rows=64, gs_lean=0):
            dev = DeviceHierarchy(ml, 0, 1)
            assert lib.amgh_debug_bw_mode(dev.h, 0) == 3 and dev.gs_sweep_stats(0, False)["slot_entries"] // A.m == 18, repr(pre)
            x_relay = dev.smooth(0, False, x0, bb)
            with tunables(lib, gs_bw_relay=0):
                assert np.array_equal(dev.smooth(0, False, x0, bb), x_relay), repr(pre)
            # (random values: every row its own dictionary entry — nothing to gain, no dictionary layout for this operator)
            assert lib.amgh_debug_bw_dict(dev.h, 0) == 0
            with tunables(lib, gs_bw_flow=0):
                assert np.array_equal(dev.smooth(0, False, x0, bb), x_relay), repr(pre)
            assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0
        xo = O.smooth(pre, A, x0, bb, hermitian=True)
        if isinstance(pre, AMG.SOR):
            assert rel(x_relay, xo) <= 1e-13, repr(pre)
        else:
            assert np.array_equal(x_relay, xo), repr(pre)


def test_dataflow_sweep_is_the_scalar_loop_bit_for_bit():
    """Gauss-Seidel forward / backward / symmetric / repeated, on grids and on an irregular symmetric pattern with zero
    diagonals (rows the sweep skips, smoother.jl:87 — and still publishes): dataflow = chained = launched = the host plan =
    the oracle's scalar loop, bitwise.  SOR against the oracle at 1e-13 (omega / d is a division on both sides, the
    products' order is the same) and bitwise against the host plan."""
    from test_bw_host import _short_rows
    lib = AMG.hip_lib()
    cases = [(AMG.poisson((20, 18, 16)), 64), (AMG.poisson((48, 40)), 64), (AMG.poisson((24, 24, 24)), 216),
             (_short_rows(5000, 5, True, zero_diag=(0, 17, 4999)), 64)]
    sm = [AMG.GaussSeidel(AMG.ForwardSweep()), AMG.GaussSeidel(AMG.BackwardSweep()), AMG.GaussSeidel(), AMG.GaussSeidel(iter=3),
          AMG.SOR(1.3), AMG.SOR(0.7, AMG.BackwardSweep())]
    ran = 0
    for A, rows in cases:
        x0, bb = uniform(A.m, 31) - 0.5, uniform(A.m, 32)
        for pre in sm:
            ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=pre)
            with tunables(lib, gs_bw=2, gs_bw_rows=rows, gs_lean=0):
                dev = DeviceHierarchy(ml, 0, 1)
                assert lib.amgh_debug_bw_mode(dev.h, 0) == 3, (A.m, repr(pre))      # really the dataflow sweep
                x_flow = dev.smooth(0, False, x0, bb)
                for rep in range(2):                                                    # epochs of the mailboxes
                    assert np.array_equal(dev.smooth(0, False, x0, bb), x_flow)
                with tunables(lib, gs_bw_relay=0):                                      # one walker per block (gs_bw_flow_kernel) instead of the relay
                    assert np.array_equal(dev.smooth(0, False, x0, bb), x_flow), (A.m, repr(pre))
                if A.m != 5000:                                                         # grids: the relay read the dictionary layout
                    assert lib.amgh_debug_bw_dict(dev.h, 0) == 1, (A.m, repr(pre))
                with tunables(lib, gs_bw_dict=0):                                       # ... and here the plain records
                    assert lib.amgh_debug_bw_dict(dev.h, 0) == 0
                    assert np.array_equal(dev.smooth(0, False, x0, bb), x_flow), (A.m, repr(pre))
                assert np.array_equal(dev.smooth(0, False, x0, bb), x_flow)             # ... and back: the two share mailboxes and epochs
                with tunables(lib, gs_bw_flow=0):
                    assert lib.amgh_debug_bw_mode(dev.h, 0) == 2
                    x_chain = dev.smooth(0, False, x0, bb)
                    with tunables(lib, gs_bw_chain=0):
                        assert lib.amgh_debug_bw_mode(dev.h, 0) == 1
                        x_launch = dev.smooth(0, False, x0, bb)
                assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0
            assert np.array_equal(x_flow, x_chain) and np.array_equal(x_flow, x_launch), (A.m, repr(pre))
            assert np.array_equal(x_flow, _host_plan_sweeps(lib, A, pre, x0, bb, rows)), (A.m, repr(pre))
            xo = O.smooth(pre, A, x0, bb, hermitian=True)
            if isinstance(pre, AMG.SOR):
                assert rel(x_flow, xo) <= 1e-13, (A.m, repr(pre))
            else:
                assert np.array_equal(x_flow, xo), (A.m, repr(pre))
            ran += 1
    assert ran == len(cases) * len(sm)


def test_dataflow_default_footprint_keeps_only_its_own_layout_and_cycles_match_the_oracle():
    """What a user gets without touching a tunable: levels swept as wavefronts of blocks hold the dataflow layout only
    (no row-major records, no flags: the run-time switches have nothing to switch to), V / W / F cycles and the solve
    are the oracle's at 1e-10 (multilevel.jl:200-239), PCG counts its iterations."""
    lib = AMG.hip_lib()
    A = AMG.poisson((40, 36, 32))
    ml = AMG.ruge_stuben(A)
    oh = O.OracleHierarchy(ml)
    b = uniform(A.m, 5) - 0.4
    with tunables(lib, gs_bw=2, gs_bw_rows=128):
        dev = DeviceHierarchy(ml, 0, 1)
    assert lib.amgh_debug_bw_mode(dev.h, 0) == 3
    with tunables(lib, gs_bw_flow=0, gs_bw_chain=0):
        assert lib.amgh_debug_bw_mode(dev.h, 0) == 3 and dev.gs_sweep_stats(0, False)["launches"] == 1
    for cyc in (0, 1, 2):
        assert rel(dev.precond_apply(b, cyc), oh.precond(b, cycle=cyc)) <= 1e-10
    x, hist, its = dev.solve(b, np.zeros(A.m), 0, 50, 0.0, 1e-9, True, True)
    xo, ho, ito = oh.solve(b, reltol=1e-9, maxiter=50)
    assert its == ito and rel(x, xo) <= 1e-10 and rel(hist, ho) <= 1e-8
    assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0


@pytest.mark.parametrize("bs", [1, 3])
def test_sweeps_that_start_from_zero_read_no_x_and_give_the_same_bits(bs):
    """Every pre-smoother of a cycle below the fine level, and the fine one of ldiv!, starts from x = 0
    (multilevel.jl:214-221): the dataflow sweep then fills its blocks' LDS with zeros instead of reading x (and nothing
    zeroes x in memory first) — the same arithmetic on the same values: bit for bit the cycle that reads x, for V / W / F
    cycles (W and F revisit levels with x != 0: those sweeps read x as ever), SOR included."""
    lib = AMG.hip_lib()
    A = AMG.poisson((30, 28, 26))
    n = A.m
    B = np.stack([uniform(n, 40 + c) - 0.2 for c in range(bs)], axis=1)
    for kw in (dict(), dict(presmoother=AMG.SOR(1.3), postsmoother=AMG.SOR(0.8, AMG.BackwardSweep()))):
        ml = AMG.ruge_stuben(A, **kw)
        with tunables(lib, gs_bw=2, gs_bw_rows=128):
            dev = DeviceHierarchy(ml, 0, bs)
            assert lib.amgh_debug_bw_mode(dev.h, 0) == 3
            rhs = B if bs > 1 else B[:, 0].copy()
            for cyc in (0, 1, 2):
                z_fast = dev.precond_apply(rhs, cyc)
                with tunables(lib, gs_flow_xzero=0):
                    assert np.array_equal(dev.precond_apply(rhs, cyc), z_fast), (cyc, kw)
        assert rel(z_fast.reshape(n, bs)[:, 0], O.OracleHierarchy(ml).precond(B[:, 0].copy(), cycle=2)) <= 1e-10


def test_dataflow_declines_structurally_nonsymmetric_patterns():
    """A block reads its far-side x when it starts; what keeps a later block from overwriting those rows first is, in the
    dataflow sweep, only the data that block waits for — so the pattern must carry every dependency both ways.  A
    structurally non-symmetric operator keeps the chained sweep (whose block-level flags order symmetrised edges)."""
    from test_bw_host import _short_rows
    lib = AMG.hip_lib()
    A = _short_rows(6000, 6, False)
    pre = AMG.GaussSeidel()
    ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=pre)
    x0, bb = uniform(A.m, 41) - 0.5, uniform(A.m, 42)
    with tunables(lib, gs_bw=2, gs_bw_rows=64):
        dev = DeviceHierarchy(ml, 0, 1)
        assert lib.amgh_debug_bw_mode(dev.h, 0) == 2
        x_dev = dev.smooth(0, False, x0, bb)
    assert np.array_equal(x_dev, _host_plan_sweeps(lib, A, pre, x0, bb, 64))


def test_a_forced_protocol_error_is_an_error_code_not_numbers():
    """One block of the sweep publishes nothing (test hook gs_bw_skip_pub): its readers' polls give up after gs_bw_spin
    tries, the sweep runs to its end on stale values — and every entry point that hands results out returns AMGH_ESTATE
    instead of them.  The word is lowered by the report: the next call on the same handle is clean and right."""
    lib = AMG.hip_lib()
    A = AMG.poisson((32, 32, 32))
    ml = AMG.ruge_stuben(A)
    b = uniform(A.m, 9)
    with tunables(lib, gs_bw=2, gs_bw_rows=64):
        dev = DeviceHierarchy(ml, 0, 1)
    assert lib.amgh_debug_bw_mode(dev.h, 0) == 3
    z_ref = dev.precond_apply(b)
    nblocks = dev.gs_sweep_stats(0, False)["rows"] // 64
    with tunables(lib, gs_bw_skip_pub=nblocks // 2, gs_bw_spin=3000):
        with pytest.raises(AMG.AMGError, match="rc=-3"):
            dev.precond_apply(b)
        with pytest.raises(AMG.AMGError, match="rc=-3"):
            dev.solve(b, np.zeros(A.m), 0, 2, 0.0, 1e-8, True, True)
        with pytest.raises(AMG.AMGError, match="rc=-3"):
            dev.smooth(0, False, np.zeros(A.m), b)
        with pytest.raises(AMG.AMGError, match="rc=-3"):
            dev.pcg(b, maxiter=2)
    assert np.array_equal(dev.precond_apply(b), z_ref)
    assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0


@pytest.mark.parametrize("case,bs", [("poisson", 8), ("poisson", 3), ("poisson", 5), ("galerkin", 4), ("galerkin", 7), ("galerkin", 2),
                                     ("poisson-sor", 4), ("galerkin-sor", 3), ("poisson", 33)])
def test_blocks_of_right_hand_sides_on_the_dataflow_layout_equal_the_single_columns_bit_for_bit(case, bs):
    """bs > 1 (`MultiLevelWorkspace{TX,bs}`, /root/reference/src/multilevel.jl:28-59; the reference loops the columns inside
    gs!, smoother.jl:77): ONE launch sweeps all columns, a workgroup taking one block for a group of up to 4 (7-point rows) /
    3 (longer rows) columns — a walker wave, an LDS x and a set of mailboxes per column, one fetcher wave for the group,
    every walker streaming the same records; odd block sizes leave walkers of the last group idle.  Per column the
    arithmetic is the single-column kernel's: the block's cycle equals the cycles of its columns bit for bit, with every
    cap on the columns per workgroup, and the oracle's at 1e-10."""
    lib = AMG.hip_lib()
    if case.startswith("poisson"):
        A = AMG.poisson((32, 28, 24))
    else:
        A = AMG.ruge_stuben(AMG.poisson((40, 40, 40)), max_levels=3).levels[1].A
    # (SOR, sor_step! smoother.jl:193-221: forward-only before, backward-only after the coarse correction)
    ml = (AMG.ruge_stuben(A, presmoother=AMG.SOR(1.2, AMG.ForwardSweep()), postsmoother=AMG.SOR(0.9, AMG.BackwardSweep(), iter=2))
          if case.endswith("-sor") else AMG.ruge_stuben(A))
    n = A.m
    B = np.stack([uniform(n, 60 + c) - 0.2 * c for c in range(bs)], axis=1)
    # (single columns sum the long rows of merged slot launches with several lanes per row — another order of additions:
    #  one thread per row on both sides, as in test_gpu_parity's block tests)
    with tunables(lib, gs_bw=2, gs_bw_rows=128, gs_lpr=1, gs_ept=1):
        dev1 = DeviceHierarchy(ml, 0, 1)
        devb = DeviceHierarchy(ml, 0, bs)
        assert lib.amgh_debug_bw_mode(devb.h, 0) == 3 and lib.amgh_debug_bw_mode(dev1.h, 0) == 3
        Z = devb.precond_apply(B)
        for nc in (1, 3, 0):
            with tunables(lib, gs_bw_nc=nc):
                assert np.array_equal(devb.precond_apply(B), Z), nc
        # (the multi-column kernels read the dictionary layout — column records + the blocks' value rows in LDS: the only record
        # layout a schedule of the default footprint holds; here a hierarchy built without it, on the plain records)
        assert lib.amgh_debug_bw_dict(devb.h, 0) == 1
        with tunables(lib, gs_bw_dict=0):
            assert lib.amgh_debug_bw_dict(devb.h, 0) == 1      # (nothing to switch to)
            devp = DeviceHierarchy(ml, 0, bs)
            assert lib.amgh_debug_bw_dict(devp.h, 0) == 0
            for nc in (2, 0):
                with tunables(lib, gs_bw_nc=nc):
                    assert np.array_equal(devp.precond_apply(B), Z), ("plain records", nc)
            del devp
        X, _, its = devb.solve(B, np.zeros_like(B), 0, 3, 0.0, 0.0, False, False)
        singles = [dev1.precond_apply(B[:, c].copy()) for c in range(bs)]
        x0 = dev1.solve(B[:, 0].copy(), np.zeros(n), 0, 3, 0.0, 0.0, False, False)[0]
        assert lib.amgh_debug_bw_poll_giveups(devb.h, 0) == 0
        with tunables(lib, gs_bw_nrhs=0):                  # the switch: blocks of right-hand sides on the level schedules
            assert lib.amgh_debug_bw_mode(DeviceHierarchy(ml, 0, bs).h, 0) == 0
    oh = O.OracleHierarchy(ml)
    for c in range(bs):
        assert np.array_equal(Z[:, c], singles[c]), (c, rel(Z[:, c], singles[c]))
    assert np.array_equal(X[:, 0], x0)
    assert rel(Z[:, bs - 1], oh.precond(B[:, bs - 1])) <= 1e-10


@pytest.fixture(scope="module")
def soak_fixture():
    A = AMG.poisson((256, 256, 256))
    ml = AMG.ruge_stuben(A, setup="gpu", device=0)
    dev = DeviceHierarchy(ml, 0, 1)
    return A, ml, dev


def test_soak_2000_sweeps_under_memory_load_eager_and_graph_replayed(soak_fixture):
    """250 V-cycles at 256^3 = 2000 dataflow sweeps of the two finest levels (16.7 M and 8.4 M rows, ~50 000 blocks,
    ~10 M mailboxes), the same right-hand side every time: every result equals the first BIT FOR BIT — eager, then
    replayed from a hipGraph — while a second stream hammers HBM with SpMVs of a 117 M-entry operator from another host
    thread (hand-offs under uneven load: stale lines, torn granules or an overtaken tag would show as a different bit
    or a raised error word).  Give-ups 0."""
    A, ml, dev = soak_fixture
    lib = AMG.hip_lib()
    assert lib.amgh_debug_bw_mode(dev.h, 0) == 3 and lib.amgh_debug_bw_mode(dev.h, 1) == 3
    n = A.m
    b = uniform(n, 3) - 0.5
    # the load: a stand-alone copy of the fine operator multiplied over and over on its own stream
    rp, ci, va = A.csr_arrays()
    rp, ci, va = np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(ci, np.int32), np.ascontiguousarray(va)
    op = ctypes.c_void_p()
    assert lib.amgh_csr_create(ctypes.byref(op), 0, n, n, rp.ctypes.data, ci.ctypes.data, va.ctypes.data) == 0
    xd, yd, sd = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.amgh_dev_alloc(0, 8 * n, ctypes.byref(xd)) == 0 and lib.amgh_dev_alloc(0, 8 * n, ctypes.byref(yd)) == 0
    assert lib.amgh_dev_alloc(0, 8 * 4096, ctypes.byref(sd)) == 0
    assert lib.amgh_dev_upload(0, xd, b.ctypes.data, 8 * n) == 0
    h2 = ctypes.c_void_p()
    assert lib.amgh_create(ctypes.byref(h2), 0, 1) == 0     # (a second handle: its non-blocking stream carries the load)
    st2 = lib.amgh_stream(h2)
    stop = threading.Event()
    launched, raised = [0], []

    def hammer():
        while not stop.is_set():
            for _ in range(8):
                if lib.amgh_csr_spmv_d(op, xd, yd, st2) != 0:
                    return
                launched[0] += 1
            out = ctypes.c_double(0)        # (a dot product: the call waits for ITS stream only — a device-wide wait from
            rc = lib.amgh_dot_d(0, 1024, yd, yd, sd, ctypes.byref(out), st2)   # another thread would break a graph capture)
            if rc != 0:
                raised.append(rc)

    z1 = dev.precond_apply(b)
    assert rel(z1, O.OracleHierarchy(ml).precond(b)) <= 1e-10
    th = threading.Thread(target=hammer)
    th.start()
    try:
        for graph in (0, 1):
            assert lib.amgh_set_use_graph(dev.h, graph) == 0
            for k in range(125):
                z = dev.precond_apply(b)
                assert np.array_equal(z, z1), (graph, k)
    finally:
        stop.set()
        th.join()
        lib.amgh_set_use_graph(dev.h, 0)
        lib.amgh_csr_destroy(op)
        lib.amgh_dev_free(0, xd); lib.amgh_dev_free(0, yd); lib.amgh_dev_free(0, sd)
        lib.amgh_destroy(h2)
    assert launched[0] > 200 and not raised     # the load really ran beside the cycles
    assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0 and lib.amgh_dev_sync(0) == 0



def test_soak_blocks_of_eight_right_hand_sides_at_full_size(soak_fixture):
    """The 256^3 hierarchy with workspace block size 8: 80 V-cycles (eager, then replayed from a hipGraph) = 640 dataflow
    launches of 4 column groups x 32 768 / 16 559 blocks, the same block of right-hand sides every time: every result equals
    the first bit for bit, its columns equal the single-column cycles bit for bit, give-ups 0."""
    A, ml, dev1 = soak_fixture
    lib = AMG.hip_lib()
    n, bs = A.m, 8
    B = np.stack([uniform(n, 20 + c) - 0.1 * c for c in range(bs)], axis=1)
    dev = DeviceHierarchy(ml, 0, bs)
    assert lib.amgh_debug_bw_mode(dev.h, 0) == 3 and lib.amgh_debug_bw_mode(dev.h, 1) == 3
    Z1 = dev.precond_apply(B)
    # (levels 2-3: single columns sum long composite rows with several lanes per row — compare like with like)
    with tunables(lib, gs_lpr=1, gs_ept=1):
        Zl = dev.precond_apply(B)
        for c in (0, 5, 7):
            assert np.array_equal(Zl[:, c], dev1.precond_apply(B[:, c].copy())), c
    try:
        for graph in (0, 1):
            assert lib.amgh_set_use_graph(dev.h, graph) == 0
            for k in range(40):
                assert np.array_equal(dev.precond_apply(B), Z1), (graph, k)
    finally:
        lib.amgh_set_use_graph(dev.h, 0)
    assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0 and lib.amgh_dev_sync(0) == 0


def test_dictionary_layout_rows_that_repeat_and_rows_that_do_not():
    """The dictionary layout of the dataflow records (gs_flow.hpp FlowDict: a block's distinct value rows in LDS, the record
    carries the columns and a dictionary index): a 19-point operator (Galerkin product of the 7-point grid: rows of up to 18
    entries), a grid with a few perturbed rows per block — the relayed sweep on it = on the plain records = the oracle's
    scalar loop, bitwise; an operator whose rows all differ (a dictionary as large as the records) keeps the plain records."""
    import scipy.sparse as sp
    lib = AMG.hip_lib()
    S = AMG.poisson((20, 20, 20)).to_scipy().tocsr()
    # the 19-point operator of a 16^3 grid: the 6 face and the 12 edge neighbours (rows of up to 18 entries)
    E, N = sp.identity(16, format="csr"), sp.diags([np.ones(15), np.ones(15)], [-1, 1], format="csr")
    k3 = lambda a, b, c: sp.kron(sp.kron(a, b), c, format="csr")
    K19 = 25.0 * sp.identity(16 ** 3) - 2.0 * (k3(N, E, E) + k3(E, N, E) + k3(E, E, N)) - (k3(N, N, E) + k3(N, E, N) + k3(E, N, N))
    A19 = AMG.SparseMatrixCSC.from_scipy(K19.tocsc())
    assert int(np.diff(K19.tocsr().indptr).max()) == 19
    P = S.copy().tolil()
    rng = np.random.default_rng(5)
    for i in rng.integers(0, S.shape[0], 300):
        P[i, i] = 6.0 + rng.random()
    Ap = AMG.SparseMatrixCSC.from_scipy(P.tocsc())
    R = S.copy()
    R.data = R.data * (1.0 + 0.01 * rng.random(R.data.size))
    R = (R + R.T) * 0.5                       # every row its own values: no dictionary (but the pattern stays symmetric)
    Ar = AMG.SparseMatrixCSC.from_scipy(sp.csc_matrix(R))
    # the 9-point operator of a 48 x 40 grid (rows of up to 8 entries: the 12-entry records)
    E2a, E2b = sp.identity(48, format="csr"), sp.identity(40, format="csr")
    N2a = sp.diags([np.ones(47), np.ones(47)], [-1, 1], format="csr")
    N2b = sp.diags([np.ones(39), np.ones(39)], [-1, 1], format="csr")
    K9 = 10.0 * sp.identity(48 * 40) - 1.5 * (sp.kron(N2a, E2b) + sp.kron(E2a, N2b)) - 0.5 * sp.kron(N2a, N2b)
    A9 = AMG.SparseMatrixCSC.from_scipy(sp.csc_matrix(K9))
    assert int(np.diff(sp.csr_matrix(K9).indptr).max()) == 9
    pre = AMG.GaussSeidel()
    for A, want, rows in ((A19, 1, 128), (Ap, 1, 216), (A9, 1, 64), (Ar, 0, 512)):
        x0, bb = uniform(A.m, 51) - 0.5, uniform(A.m, 52)
        ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=pre)
        with tunables(lib, gs_bw=2, gs_bw_rows=rows, gs_lean=0):
            dev = DeviceHierarchy(ml, 0, 1)
            assert lib.amgh_debug_bw_mode(dev.h, 0) == 3
            assert lib.amgh_debug_bw_dict(dev.h, 0) == want, (A.m, A.nnz)
            xd = dev.smooth(0, False, x0, bb)
            with tunables(lib, gs_bw_dict=0):
                assert np.array_equal(dev.smooth(0, False, x0, bb), xd)
            assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0
        assert np.array_equal(xd, O.smooth(pre, A, x0, bb, hermitian=True)), (A.m, A.nnz)
