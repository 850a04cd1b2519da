"""BASELINE.json configurations at FULL size on the GPU: C2 (2-D 1024^2, smoothed aggregation +
Jacobi) and C3 (3-D 256^3, Ruge-Stuben + symmetric Gauss-Seidel).  Checked against the CPU oracle
(which still finishes in seconds per cycle) and through size-independent properties."""
import numpy as np
import pytest

import amg_amd as AMG
from conftest import uniform
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def rel(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y)) / max(np.linalg.norm(y), 1e-300)


def test_c2_poisson_1024x1024_sa_jacobi_full_size():
    A = AMG.poisson((1024, 1024))
    n = A.m
    assert n == 1048576 and A.nnz == 5 * n - 4 * 1024
    jac = AMG.Jacobi(2.0 / 3.0)
    ml = AMG.smoothed_aggregation(A, presmoother=jac, postsmoother=jac)
    oh = O.OracleHierarchy(ml)
    b = uniform(n, 0)
    x, hist = AMG._solve(ml, b, reltol=1e-8, maxiter=300, log=True)
    xo, ho, _ = oh.solve(b, reltol=1e-8, maxiter=300)
    assert len(hist) == len(ho) and hist[-1] <= 1e-8 * hist[0]
    assert rel(x, xo) <= 1e-10
    p = AMG.aspreconditioner(ml)
    r1, r2 = uniform(n, 1) - 0.5, uniform(n, 2) - 0.5
    z1, z2 = p.ldiv(r1), p.ldiv(r2)
    assert rel(p.ldiv(2.0 * r1 - 3.0 * r2), 2.0 * z1 - 3.0 * z2) <= 1e-10        # linear
    assert abs(z1 @ r2 - r1 @ z2) <= 1e-9 * abs(z1 @ r2)                          # symmetric (Jacobi pre = post)
    assert rel(z1, oh.precond(r1)) <= 1e-10


def test_2d_gauss_seidel_hierarchy_large_levels_as_wavefronts_of_blocks():
    """2-D grids have two offset classes: their wavefront of blocks is a line; operators of >= 200 000 rows take it where the
    cost model of the dataflow execution agrees (tunable gs_bw_two_min_rows; round 4: 6 M rows — 4096^2: the 16.8 M-row
    level 8.5 -> 6.3 ms per smoother, profiles/r04_2d_blocks.log; round 5: profiles/r05_block_layout_threshold.log).
    A 3072 x 2048 Poisson hierarchy with the defaults (ruge_stuben, symmetric Gauss-Seidel, smoother.jl:61-90): the fine
    level and the 3.1 M-row level below it run the dataflow sweep, small levels keep the merged groups, the cycle is the
    oracle's at 1e-10 and the fine-level smoother alone is the scalar loop bit for bit."""
    lib = AMG.hip_lib()
    A = AMG.poisson((3072, 2048))
    n = A.m
    ml = AMG.ruge_stuben(A, setup="gpu", device=0)
    dev = ml.device(0, 1)
    assert lib.amgh_debug_bw_mode(dev.h, 0) == 3 and lib.amgh_debug_bw_mode(dev.h, 1) == 3
    assert lib.amgh_debug_bw_mode(dev.h, len(ml.levels) - 1) == 0
    assert dev.gs_sweep_stats(0, False)["launches"] == 1
    b = uniform(n, 5) - 0.3
    oh = O.OracleHierarchy(ml)
    assert rel(dev.precond_apply(b), oh.precond(b)) <= 1e-10
    x0 = uniform(n, 6) - 0.5
    assert np.array_equal(dev.smooth(0, False, x0, b), O.smooth(ml.levels[0].presmoother, A, x0, b, hermitian=True))
    assert lib.amgh_debug_bw_poll_giveups(dev.h, 0) == 0


@pytest.fixture(scope="module")
def c3():
    """poisson((256,256,256)) + ruge_stuben defaults, built once for the C3 and C4 tests."""
    A = AMG.poisson((256, 256, 256))
    ml = AMG.ruge_stuben(A)
    return A, ml


def test_c3_gpu_built_hierarchy_is_the_host_built_one_at_full_size(c3):
    """What bench.py times (`ruge_stuben(A, setup="gpu", device=0)`: strength, interpolation, transposes and R*A*P on
    the GPU, the HBM hierarchy built level by level beside the splitting) is THE reference hierarchy, not merely a
    convergent one: every level's A, P and R equal the host library's (pinned by the reference's setup goldens) bit
    for bit at 256^3 — structure and values.  At this size the SpGEMM meets columns that outgrow its LDS hash table
    (the host-library fallback), which the small cases of tests/test_gpu_setup.py never reach."""
    A, ml = c3
    g = AMG.ruge_stuben(A, setup="gpu", device=0)
    assert len(g) == len(ml)

    def same(X, Y):
        return (X.shape == Y.shape and np.array_equal(X.colptr, Y.colptr) and np.array_equal(X.rowval, Y.rowval)
                and np.array_equal(X.nzval, Y.nzval))
    for l, (a, b) in enumerate(zip(ml.levels, g.levels)):
        assert same(a.A, b.A), f"A differs on level {l}"
        assert same(a.P, b.P), f"P differs on level {l}"
        assert same(a.R, b.R), f"R differs on level {l}"
    assert same(ml.final_A, g.final_A)
    # and the handle built on the way runs the same cycle as the one built afterwards from the host hierarchy
    b = uniform(A.m, 0)
    z_pipe = AMG.aspreconditioner(g).ldiv(b)
    z_host = AMG.aspreconditioner(ml).ldiv(b)
    assert np.array_equal(z_pipe, z_host)
    del g


def test_c3_poisson_256cubed_rs_gauss_seidel_full_size(c3):
    A, ml = c3
    n = A.m
    assert n == 16777216 and A.nnz == 117047296
    assert [l.A.m for l in ml.levels][:3] == [16777216, 8388608, 1398103]
    oh = O.OracleHierarchy(ml)
    p = AMG.aspreconditioner(ml)
    b = uniform(n, 0)
    # one V-cycle (ldiv!) against the oracle at full size
    z = p.ldiv(b)
    assert rel(z, oh.precond(b)) <= 1e-10
    assert np.array_equal(p.ldiv(b), z)       # bitwise reproducible run to run (no races in ~2 000 dependent launches)
    # the V-cycle with symmetric Gauss-Seidel pre/post is a symmetric linear operator
    r2 = uniform(n, 2) - 0.5
    z2 = p.ldiv(r2)
    assert rel(p.ldiv(b - 2.0 * r2), z - 2.0 * z2) <= 1e-10
    assert abs(z @ r2 - b @ z2) <= 1e-9 * abs(z @ r2)
    # stationary iteration: converges to reltol, residuals fall monotonically, x solves A x = b
    x, hist = AMG._solve(ml, b, reltol=1e-8, log=True)
    # (direct interpolation on the 7-point stencil contracts by ~0.78 per V-cycle at this size: ~75 cycles)
    assert hist[-1] <= 1e-8 * hist[0] and np.all(np.diff(hist) < 0) and len(hist) - 1 < 100
    r = b - A.to_scipy() @ x
    assert np.isclose(np.linalg.norm(r), hist[-1], rtol=1e-6)
    # fine-level operator: interior rows annihilate constants, <Ax,y> = <x,Ay>
    dev = ml.device()
    ones = dev.spmv(0, 0, np.ones(n)).reshape((256, 256, 256))
    assert np.all(ones[1:-1, 1:-1, 1:-1] == 0)
    Ab, Ar2 = dev.spmv(0, 0, b), dev.spmv(0, 0, r2)
    assert abs(Ab @ r2 - b @ Ar2) <= 1e-12 * abs(Ab @ r2)


# ---- C4: 256^3 row-sharded (virtual ranks on the one GPU of the box; the RCCL transport needs more GPUs) --------------
def _sharded(ml, nranks, fn, gs_mode="hybrid"):
    from amg_amd import sharded as SH

    def work(rank, group):
        sh = SH.ShardedHierarchy.from_multilevel(ml, rank, nranks, 0, ("local", group), gs_mode=gs_mode)
        try:
            return fn(sh)
        finally:
            sh.close()
    return SH.run_local_ranks(nranks, work)


def _with_smoothers(ml, pre, post):
    """The same hierarchy matrices with other smoothers (no second setup)."""
    levels = [AMG.Level(l.A, l.P, l.R, pre, post) for l in ml.levels]
    return AMG.MultiLevel(levels, ml.final_A, ml.coarse_solver, pre, post, ml.symmetry, method=ml.method)


@pytest.mark.parametrize("nranks", [2, 8])
def test_c4_256cubed_row_sharded_jacobi_equals_the_oracle_per_cycle(c3, nranks):
    """Jacobi / residual / R / P across shards are exactly the single-GPU arithmetic: every cycle's iterate and
    residual norm equal the oracle's at the north-star tolerance."""
    A, ml = c3
    n = A.m
    jac = AMG.Jacobi(2.0 / 3.0)
    mlj = _with_smoothers(ml, jac, jac)
    b = uniform(n, 0)
    cycles = 3

    def fn(sh):
        assert sh.lc == 4                       # 16.7M / 8.4M / 1.4M / 229k rows sharded, 38k and below on rank 0
        xs = [sh.solve(b[sh.r0:sh.r1], maxiter=k, calculate_residual=False)[0] for k in range(1, cycles + 1)]
        _, hist = sh.solve(b[sh.r0:sh.r1], maxiter=cycles, reltol=1e-30)
        return xs, hist, sh.stats()
    res = _sharded(mlj, nranks, fn)
    oh = O.OracleHierarchy(mlj)
    _, ho, _ = oh.solve(b, maxiter=cycles, reltol=1e-30)
    assert np.allclose(res[0][1], ho, rtol=1e-9)
    for k in range(cycles):
        xo, _, _ = oh.solve(b, maxiter=k + 1, calculate_residual=False)
        x = np.concatenate([r[0][k] for r in res])
        assert rel(x, xo) <= 1e-10, (nranks, k)
    # z-slab partition: a rank's fine-level halo is one 256^2 plane per neighbour
    from amg_amd import sharded as SH
    assert all(r[2]["halo_exchanges"] > 0 for r in res)


def test_c4_256cubed_row_sharded_gauss_seidel_matches_the_frozen_halo_emulation_per_cycle(c3):
    """The default smoother across 4 shards: exact lexicographic order inside a shard, halo frozen per directional
    sweep.  Every cycle's iterate equals a host emulation of exactly that sweep (oracle loops on the global
    matrices with the rows outside the shard emptied), so a change in the freeze order cannot pass; and the
    deviation from the single-GPU (exact) cycle stays a small perturbation."""
    from dist_backends import emulate_sharded_cycles
    A, ml = c3
    n = A.m
    b = uniform(n, 0)
    nranks, cycles = 4, 2

    def fn(sh):
        return [sh.solve(b[sh.r0:sh.r1], maxiter=k, calculate_residual=False)[0] for k in range(1, cycles + 1)]
    res = _sharded(ml, nranks, fn)
    want = emulate_sharded_cycles(ml, b, nranks, 4, cycles)
    oh = O.OracleHierarchy(ml)
    for k in range(cycles):
        x = np.concatenate([r[k] for r in res])
        assert rel(x, want[k]) <= 1e-10, k
        xo, _, _ = oh.solve(b, maxiter=k + 1, calculate_residual=False)
        assert rel(x, xo) <= 5e-2            # hybrid vs exact sweep: same cycle up to the 3 interfaces per level


def test_c4_256cubed_row_sharded_exact_gauss_seidel_is_the_oracle_cycle(c3):
    """Config C4 with the library's default across shards — lexicographic Gauss-Seidel over the whole level, the ranks
    sweeping in turn: one V-cycle of the default hierarchy on 2 and 4 shards IS the single-process oracle's cycle at the
    north-star tolerance (/root/reference/src/smoother.jl:61-90, multilevel.jl:214-239)."""
    A, ml = c3
    b = uniform(A.m, 0)
    want = O.OracleHierarchy(ml).precond(b)
    for nranks in (2, 4):
        res = _sharded(ml, nranks, lambda sh: sh.solve(b[sh.r0:sh.r1], maxiter=1, calculate_residual=False)[0], gs_mode="exact")
        assert rel(np.concatenate(res), want) <= 1e-10, nranks


def test_c4_two_processes_over_the_ipc_transport_full_size(c3):
    """Config C4 with REAL processes at full size: 256^3 row-sharded over 2 processes (one GPU, IPC transport: hipIpc
    peer-mapped send buffers + stream-written flags), one V-cycle of the default (Gauss-Seidel) hierarchy against the
    host emulation of the frozen-halo sweeps at 1e-10."""
    import subprocess
    import sys
    import tempfile
    import os
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from amg_amd import sharded as SH
    from sharded_emulation import emulate_sharded_cycles
    A, ml = c3
    b = uniform(A.m, 0)
    name = "/amgh_c4_%d_%s" % (os.getpid(), os.urandom(4).hex())
    env = dict(os.environ, AMGH_IPC_TIMEOUT_S="300", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as outdir:
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_gpu_worker.py"), str(r), "2", name, outdir, "c4"],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT) for r in range(2)]
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append((p.returncode, out.decode()))
        for r, (rc, out) in enumerate(outs):
            assert rc == 0 and f"IPC_GPU_RANK_{r}_OK" in out, (r, out[-3000:])
        parts = [dict(np.load(os.path.join(outdir, f"rank{r}.npz"))) for r in range(2)]
    lc = SH.num_sharded_levels([l.A.m for l in ml.levels] + [ml.final_A.m], 2)
    assert all(int(p["lc"]) == lc for p in parts) and lc >= 3
    got = np.concatenate([p["cycles"][0] for p in parts])
    want = emulate_sharded_cycles(ml, b, 2, lc, 1)[0]
    assert rel(got, want) <= 1e-10
