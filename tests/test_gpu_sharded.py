"""The row-sharded cycle behind the C ABI (`amgh_dist_*`, include/amghip.h): N ranks on the LOCAL transport
(threads of this process sharing the one GPU a gpurun box has — halo entries move by device-to-device copies
between the ranks' streams).  Jacobi / residual / R / P are exactly the single-GPU arithmetic -> compared with the
oracle at 1e-10 per cycle.  Gauss-Seidel across shards is the processor-block hybrid (exact inside a shard, halo
frozen per directional sweep): compared PER CYCLE with a host emulation of the same frozen-halo sweeps (the
Python mirror of the sharded driver with the oracle's loops as local arithmetic).  The RCCL transport is driven
with one rank (communicator, all-reduce, self-contained exchange plan); more ranks need more GPUs."""
import os

import numpy as np
import pytest

import amg_amd as AMG
import dist_mirror as D  # noqa: E402
from amg_amd import sharded as SH
from conftest import uniform
from dist_backends import OracleOps, emulate_sharded_cycles, run_virtual_ranks
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def rel(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y)) / max(np.linalg.norm(y), 1e-300)


def sharded_run(ml, nranks, shard_min_rows, fn, gs_mode="hybrid"):
    """fn(sh) on every rank; returns the per-rank results.  gs_mode: "hybrid" (every shard sweeps at once, halo frozen per
    directional sweep — what the emulations below restate) or "exact" (the library's default: the ranks sweep in turn)."""
    def work(rank, group):
        sh = SH.ShardedHierarchy.from_multilevel(ml, rank, nranks, 0, ("local", group), shard_min_rows, gs_mode=gs_mode)
        return fn(sh)
    return SH.run_local_ranks(nranks, work)


def sharded_solve(ml, b, nranks, shard_min_rows, gs_mode="hybrid", **kw):
    res = sharded_run(ml, nranks, shard_min_rows, lambda sh: (*sh.solve(b[sh.r0:sh.r1], **kw), sh.lc, sh.stats()), gs_mode)
    return np.concatenate([r[0] for r in res]), res[0][1], res[0][2], [r[3] for r in res]


def emulated_cycles(ml, b, nranks, shard_min_rows, cycles, cyc=0):
    """Host emulation of the sharded cycle: iterates after 1..cycles cycles from x0 = 0."""
    def work(comm):
        dml = D.DistMultiLevel(ml, comm, OracleOps(), shard_min_rows=shard_min_rows)
        r0, r1 = dml.local_range(0)
        dml.set_rhs(b[r0:r1])
        dml.ops.zero(dml.x[0], r1 - r0)
        out = []
        for _ in range(cycles):
            dml.cycle(0, cyc)
            out.append(dml.ops.download(dml.x[0], r1 - r0))
        return out
    res = run_virtual_ranks(nranks, work)
    return [np.concatenate([r[k] for r in res]) for k in range(cycles)]


def sharded_cycles(ml, b, nranks, shard_min_rows, cycles, cyc=0, gs_mode="hybrid"):
    def fn(sh):
        out = []
        for k in range(1, cycles + 1):   # exactly k cycles from x0 = 0 (calculate_residual = False)
            x, _ = sh.solve(b[sh.r0:sh.r1], cycle=cyc, maxiter=k, calculate_residual=False)
            out.append(x)
        return out
    res = sharded_run(ml, nranks, shard_min_rows, fn, gs_mode)
    return [np.concatenate([r[k] for r in res]) for k in range(cycles)]


@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_exact_gauss_seidel_across_shards_is_the_oracle_cycle_for_cycle(nranks):
    """The library's default on sharded levels: lexicographic Gauss-Seidel / SOR over the WHOLE level (smoother.jl:61-90,
    :193-221) — the ranks sweep in turn, every turn's boundary values travel before the next — so every cycle's iterate is
    the single-process oracle's at the north-star tolerance: symmetric, directional and repeated sweeps, SOR, V / W / F."""
    A = AMG.poisson((40, 36, 48))
    b = uniform(A.m, 6) - 0.3
    cases = [(AMG.GaussSeidel(), AMG.GaussSeidel()),
             (AMG.GaussSeidel(AMG.ForwardSweep(), iter=2), AMG.GaussSeidel(AMG.BackwardSweep())),
             (AMG.SOR(1.2), AMG.SOR(0.9, AMG.ForwardSweep()))]
    for pre, post in cases:
        ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=post)
        oh = O.OracleHierarchy(ml)
        for cyc in ((0, 1, 2) if pre is cases[0][0] else (0,)):
            got = sharded_cycles(ml, b, nranks, 4000, 2, cyc=cyc, gs_mode="exact")
            for k in range(2):
                xo, _, _ = oh.solve(b, cycle=cyc, maxiter=k + 1, calculate_residual=False)
                assert rel(got[k], xo) <= 1e-10, (nranks, repr(pre), cyc, k)
    ml = AMG.ruge_stuben(A)
    x, hist, lc, stats = sharded_solve(ml, b, nranks, 4000, gs_mode="exact", reltol=1e-9, maxiter=60)
    xo, ho, _ = O.OracleHierarchy(ml).solve(b, reltol=1e-9, maxiter=60)
    assert lc >= 2 and len(hist) == len(ho) and np.allclose(hist, ho, rtol=1e-8) and rel(x, xo) <= 1e-10


@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_gauss_seidel_pipelined_across_the_ranks_is_the_oracle_and_the_turns_bit_for_bit(nranks):
    """Exact order as ONE sweep (amgh_dist_set_gs_mode 1 where every rank holds the dataflow layout of its shard — forced here on
    small shards): all ranks launch at once, a block polls the rows it reads of the neighbouring rank in that rank's mailboxes.
    Every cycle's iterate is the oracle's (1e-10) and, row for row the same arithmetic, BITWISE the iterate of the ranks
    sweeping in turn; GS forward / backward / symmetric, SOR, V and W cycles; the ranks share this box's one GPU, so every
    launch is the persistent form with all ranks' workgroups resident."""
    from test_gpu_flow import tunables
    lib = AMG.hip_lib()
    A = AMG.poisson((40, 36, 48))
    b = uniform(A.m, 6) - 0.3
    cases = [(AMG.GaussSeidel(), AMG.GaussSeidel(), (0, 1)),
             (AMG.GaussSeidel(AMG.ForwardSweep(), iter=2), AMG.GaussSeidel(AMG.BackwardSweep()), (0,)),
             (AMG.SOR(1.2), AMG.SOR(0.9, AMG.ForwardSweep()), (0,))]
    serialized = False
    with tunables(lib, gs_bw=2, gs_bw_rows=64):
        for pre, post, cycs in cases:
            ml = AMG.ruge_stuben(A, presmoother=pre, postsmoother=post)
            oh = O.OracleHierarchy(ml)
            piped = sharded_run(ml, nranks, 4000, lambda sh: (sh.gs_pipelined(), sh.pipe_serialized()), "exact")
            # (virtual ranks are threads of ONE process: where two of their streams share a hardware queue the library finds out at
            # finalize — a bounded probe — and sweeps in turns; everything below still holds, the pipeline itself then is what the
            # IPC-process tests of test_gpu_ipc.py run)
            serialized = serialized or any(p[1] for p in piped)
            if not serialized:
                assert all(len(p[0]) >= 2 and p[0][0] and p[0][1] for p in piped), piped   # the two large levels, on every rank (a third, of ~2 900 rows, is too small for blocks: in turns)
            for cyc in cycs:
                got = sharded_cycles(ml, b, nranks, 4000, 2, cyc=cyc, gs_mode="exact")
                turns = sharded_cycles(ml, b, nranks, 4000, 2, cyc=cyc, gs_mode="exact-turns")
                for k in range(2):
                    xo, _, _ = oh.solve(b, cycle=cyc, maxiter=k + 1, calculate_residual=False)
                    assert rel(got[k], xo) <= 1e-10, (nranks, repr(pre), cyc, k)
                    assert np.array_equal(got[k], turns[k]), (nranks, repr(pre), cyc, k)
        assert lib.amgh_dev_sync(0) == 0
    if serialized:
        # (the library re-creates the ranks' sweep streams in rank order and probes again, three times, before it settles for turns:
        # with the 8 hardware queues tests/conftest.py asks for, 2-4 virtual ranks must end up side by side)
        msg = "the virtual ranks' streams shared a hardware queue in this process: swept in turns (oracle parity held)"
        if os.environ.get("GPU_MAX_HW_QUEUES") == "8":
            pytest.fail(msg + " although GPU_MAX_HW_QUEUES=8 was requested and the streams were re-created: the pipelined sweep did not run")
        pytest.skip(msg)


def test_mailbox_protocol_probe_guards_the_pipelined_sweep():
    """amgh_dist_finalize runs the mailbox protocol itself between neighbouring ranks (the sweeps' own write-through store /
    system-scope poll on memory mapped as the mailbox arrays are) before any level is allowed to depend on it: it passes between
    ranks on this device (levels pipeline), and a rank that publishes nothing (test hook) makes its neighbours run into the
    bound — every level then sweeps with the ranks in turn, on every rank, and the cycle is still the oracle's
    (multilevel.jl:214-239)."""
    from test_gpu_flow import tunables
    lib = AMG.hip_lib()
    A = AMG.poisson((40, 36, 48))
    b = uniform(A.m, 9) - 0.3
    ml = AMG.ruge_stuben(A)
    xo, _, _ = O.OracleHierarchy(ml).solve(b, maxiter=2, calculate_residual=False)

    def probe(sh):
        x, _ = sh.solve(b[sh.r0:sh.r1], maxiter=2, calculate_residual=False)
        return x, sh.gs_pipelined(), sh.pipe_protocol_failed(), sh.pipe_serialized()
    with tunables(lib, gs_bw=2, gs_bw_rows=64):
        good = sharded_run(ml, 3, 4000, probe, "exact")
        assert not any(r[2] for r in good)
        if not any(r[3] for r in good):
            assert all(r[1] and r[1][0] for r in good), [r[1] for r in good]
        assert rel(np.concatenate([r[0] for r in good]), xo) <= 1e-10
        os.environ["AMGH_MAIL_PROBE_MUTE"] = "1"
        os.environ["AMGH_MAIL_PROBE_MS"] = "100"
        try:
            bad = sharded_run(ml, 3, 4000, probe, "exact")
        finally:
            os.environ.pop("AMGH_MAIL_PROBE_MUTE", None)
            os.environ.pop("AMGH_MAIL_PROBE_MS", None)
        if not any(r[3] for r in bad):
            assert all(r[2] for r in bad), [r[2] for r in bad]                      # found on every rank (collective)
        assert not any(any(r[1]) for r in bad), [r[1] for r in bad]                # no level pipelines
        assert np.array_equal(np.concatenate([r[0] for r in bad]), np.concatenate([r[0] for r in good]))
        assert lib.amgh_dev_sync(0) == 0


@pytest.mark.parametrize("nranks", [2, 4])
def test_sharded_jacobi_equals_the_oracle_cycle_for_cycle(nranks):
    A = AMG.poisson((32, 24, 20))
    b = uniform(A.m, 5)
    jac = AMG.Jacobi(2.0 / 3.0, iter=2)
    ml = AMG.ruge_stuben(A, presmoother=jac, postsmoother=jac)
    oh = O.OracleHierarchy(ml)
    for cyc in (0, 1, 2):
        x, hist, lc, stats = sharded_solve(ml, b, nranks, 500, cycle=cyc, reltol=1e-8, maxiter=60)
        assert lc >= 2
        xo, ho, _ = oh.solve(b, cycle=cyc, reltol=1e-8, maxiter=60)
        assert len(hist) == len(ho)
        assert np.allclose(hist, ho, rtol=1e-9)
        assert rel(x, xo) <= 1e-10
        assert all(s["halo_exchanges"] > 0 and s["halo_bytes_sent"] > 0 for s in stats)


def test_halo_plan_of_a_z_slab_partition():
    """7-point stencil, first axis fastest, 2 ranks: each rank needs exactly the neighbour's boundary plane, and
    the rows that read it are the first / last plane of the shard (everything else is interior)."""
    A = AMG.poisson((8, 8, 12))
    ml = AMG.ruge_stuben(A, max_levels=2)
    infos = sharded_run(ml, 2, 100, lambda sh: (sh.plan_info(0), sh.r0, sh.r1))
    (p0, a0, a1), (p1, b0, b1) = infos
    assert (a0, a1, b0, b1) == (0, 384, 384, 768)
    assert np.array_equal(p0["halo"], np.arange(384, 448)) and np.array_equal(p1["halo"], np.arange(320, 384))
    assert np.array_equal(p0["send_idx"], np.arange(320, 384)) and np.array_equal(p1["send_idx"], np.arange(0, 64))
    assert p0["send_cnt"].tolist() == [0, 64] and p0["recv_cnt"].tolist() == [0, 64]
    assert p0["interior"] == (0, 320) and p1["interior"] == (64, 384)


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_hybrid_gauss_seidel_matches_the_frozen_halo_emulation_per_cycle(nranks):
    A = AMG.poisson((40, 40, 48))
    b = uniform(A.m, 6)
    ml = AMG.ruge_stuben(A)
    got = sharded_cycles(ml, b, nranks, 4000, 3)
    want = emulated_cycles(ml, b, nranks, 4000, 3)
    lc = SH.num_sharded_levels([l.A.m for l in ml.levels] + [ml.final_A.m], nranks, 4000)
    want2 = emulate_sharded_cycles(ml, b, nranks, lc, 3)   # the global-matrix emulation the full-size test uses
    for k in range(3):
        assert rel(got[k], want[k]) <= 1e-10, (nranks, k)
        assert rel(want2[k], want[k]) <= 1e-12, (nranks, k)
    # and the hybrid converges to the same solution as the exact lexicographic sweep, in about as many cycles
    x, hist, lc, _ = sharded_solve(ml, b, nranks, 4000, reltol=1e-10, maxiter=60)
    xo, ho, _ = O.OracleHierarchy(ml).solve(b, reltol=1e-10, maxiter=60)
    assert lc >= 2 and hist[-1] <= 1e-10 * hist[0]
    assert rel(x, xo) <= 1e-8 and abs(len(hist) - len(ho)) <= 2


def test_sor_and_w_cycle_on_shards_match_the_emulation():
    A = AMG.poisson((32, 32, 32))
    b = uniform(A.m, 9)
    ml = AMG.ruge_stuben(A, presmoother=AMG.SOR(1.2, AMG.ForwardSweep()), postsmoother=AMG.SOR(1.2, AMG.BackwardSweep()))
    got = sharded_cycles(ml, b, 4, 2000, 2, cyc=1)
    want = emulated_cycles(ml, b, 4, 2000, 2, cyc=1)
    assert rel(got[0], want[0]) <= 1e-10 and rel(got[1], want[1]) <= 1e-10


def test_everything_collapsed_and_single_rank():
    A = AMG.poisson((20, 20, 20))
    b = uniform(A.m, 7)
    ml = AMG.ruge_stuben(A)
    xo, ho = AMG._solve(ml, b, reltol=1e-8, log=True)
    # problem below the shard threshold: rank 0 runs the plain single-GPU cycle, the others hold nothing
    x, hist, lc, _ = sharded_solve(ml, b, 2, 10 ** 9, reltol=1e-8)
    assert lc == 0 and len(hist) == len(ho) and rel(x, xo) <= 1e-12
    # one rank, two sharded levels: no exchange at all, the exact sweep
    x, hist, lc, stats = sharded_solve(ml, b, 1, 500, reltol=1e-8)
    assert lc >= 2 and len(hist) == len(ho) and rel(x, xo) <= 1e-12 and stats[0]["halo_exchanges"] == 0


def test_rccl_transport_with_one_rank():
    """ncclGetUniqueId / ncclCommInitRank / ncclAllReduce through the dlopen'ed librccl (more ranks need more GPUs)."""
    lib = AMG.hip_lib()
    if not lib.amgh_dist_rccl_available():
        pytest.skip("librccl not found")
    A = AMG.poisson((20, 20, 20))
    b = uniform(A.m, 7)
    ml = AMG.ruge_stuben(A)
    sh = SH.ShardedHierarchy.from_multilevel(ml, 0, 1, 0, ("rccl", SH.rccl_unique_id()), 500)
    x, hist = sh.solve(b, reltol=1e-8)
    xo, ho = AMG._solve(ml, b, reltol=1e-8, log=True)
    assert len(hist) == len(ho) and rel(x, xo) <= 1e-12
    assert sh.allreduce([1.5, 2.0])[1] == 2.0 and sh.allreduce([3.0], "max")[0] == 3.0
    sh.barrier()
