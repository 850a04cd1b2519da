"""The row-sharded cycle with REAL processes: N worker processes (tests/ipc_gpu_worker.py), each one rank of
`amgh_dist_*` over the IPC transport (hipIpc peer-mapped send buffers, stream-written / stream-awaited flags in a
POSIX shared-memory segment), all on the one GPU a gpurun box has.  This is the multi-process exchange path of
BASELINE.json config C4 executed for real — plan exchange between processes, ordering of overlapped exchanges,
reuse of the double-buffered send copies, collapse onto rank 0 — checked against the oracle (Jacobi: exact) and the
frozen-halo emulation (Gauss-Seidel / SOR) cycle by cycle."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import amg_amd as AMG
from amg_amd import sharded as SH
from conftest import ROOT
from dist_backends import emulate_sharded_cycles
from ipc_cases import assemble, build_case
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def rel(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y)) / max(np.linalg.norm(y), 1e-300)


def run_ipc(case, nranks, timeout=420, expect_rc=None, env_extra=None, transport="ipc"):
    name = "rccl:" if transport == "rccl" else "/amgh_g_%d_%s" % (os.getpid(), os.urandom(4).hex())
    env = dict(os.environ, AMGH_IPC_TIMEOUT_S="90", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    with tempfile.TemporaryDirectory() as outdir:
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_gpu_worker.py"), str(r), str(nranks),
                                   name, outdir, case], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT)
                 for r in range(nranks)]
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append((p.returncode, out.decode()))
        if expect_rc is not None:
            return outs
        for r, (rc, out) in enumerate(outs):
            assert rc == 0 and f"IPC_GPU_RANK_{r}_OK" in out, (case, nranks, r, out[-3000:])
        return [dict(np.load(os.path.join(outdir, f"rank{r}.npz"))) for r in range(nranks)]


@pytest.mark.parametrize("nranks", [2, 4])
def test_ipc_processes_jacobi_equals_the_oracle(nranks):
    ml, b, smr, _ = build_case("jacobi")
    parts = run_ipc("jacobi", nranks)
    assert all(int(p["lc"]) >= 2 for p in parts) and all(int(p["halo_exchanges"]) > 0 for p in parts)
    oh = O.OracleHierarchy(ml)
    for key, cyc in (("solve_v", 0), ("solve_w", 1), ("solve_f", 2)):
        xo, ho, _ = oh.solve(b, cycle=cyc, reltol=1e-8, maxiter=60)
        hist = parts[0][key + "_hist"]
        assert all(np.array_equal(p[key + "_hist"], hist) for p in parts)   # the same bits on every rank
        assert len(hist) == len(ho) and np.allclose(hist, ho, rtol=1e-9)
        assert rel(assemble(parts, key + "_x"), xo) <= 1e-10
    assert rel(assemble(parts, "ldiv"), oh.precond(b)) <= 1e-10
    assert rel(assemble(parts, "spmv"), O.spmv(ml.levels[0].A, b)) <= 1e-13
    # 30 more preconditioner applications on the same right-hand side: the same result as the first one
    assert rel(assemble(parts, "repeat"), assemble(parts, "ldiv")) == 0.0


def test_ipc_processes_overlap_interior_rows_with_the_exchange():
    ml, b, smr, _ = build_case("jacobi_overlap")
    want = emulate_sharded_cycles(ml, b, 2, SH.num_sharded_levels([l.A.m for l in ml.levels] + [ml.final_A.m], 2, smr), 2)
    for overlap in ("1", "0"):
        parts = run_ipc("jacobi_overlap", 2, env_extra={"AMGH_DIST_OVERLAP": overlap})
        got = assemble(parts, "cycles")
        for k in range(2):
            assert rel(got[k], want[k]) <= 1e-10, (overlap, k)
        assert rel(assemble(parts, "spmv"), O.spmv(ml.levels[0].A, b)) <= 1e-13


@pytest.mark.parametrize("nranks", [2, 3])
def test_ipc_processes_hybrid_gauss_seidel_matches_the_frozen_halo_emulation(nranks):
    ml, b, smr, _ = build_case("gs")
    lc = SH.num_sharded_levels([l.A.m for l in ml.levels] + [ml.final_A.m], nranks, smr)
    parts = run_ipc("gs", nranks)
    assert all(int(p["lc"]) == lc for p in parts) and lc >= 2
    want = emulate_sharded_cycles(ml, b, nranks, lc, 3)
    got = assemble(parts, "cycles")
    for k in range(3):
        assert rel(got[k], want[k]) <= 1e-10, (nranks, k)
    xo, ho, _ = O.OracleHierarchy(ml).solve(b, reltol=1e-10, maxiter=60)
    hist = parts[0]["solve_v_hist"]
    assert hist[-1] <= 1e-10 * hist[0] and abs(len(hist) - len(ho)) <= 2
    assert rel(assemble(parts, "solve_v_x"), xo) <= 1e-8


@pytest.mark.parametrize("nranks", [2, 3])
def test_ipc_processes_exact_gauss_seidel_is_the_oracle_cycle(nranks):
    """REAL processes, the library's default across shards (lexicographic order over the whole level: the ranks sweep in
    turn, their boundary values exchanged between turns): every cycle and the solve equal the single-process oracle's."""
    ml, b, smr, _ = build_case("gs")
    parts = run_ipc("gs", nranks, env_extra={"AMG_DIST_GS_MODE": "exact"})
    oh = O.OracleHierarchy(ml)
    got = assemble(parts, "cycles")
    for k in range(3):
        xo, _, _ = oh.solve(b, maxiter=k + 1, calculate_residual=False)
        assert rel(got[k], xo) <= 1e-10, (nranks, k)
    xo, ho, _ = oh.solve(b, reltol=1e-10, maxiter=60)
    hist = parts[0]["solve_v_hist"]
    assert len(hist) == len(ho) and np.allclose(hist, ho, rtol=1e-8)
    assert rel(assemble(parts, "solve_v_x"), xo) <= 1e-10


@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_ipc_processes_pipelined_gauss_seidel_is_the_oracle_cycle(nranks):
    """REAL processes, exact order as ONE sweep pipelined across the ranks: every process maps its neighbours' mailbox arrays
    with hipIpc handles, its blocks poll the neighbour's cells through that mapping while all the ranks' (persistent) launches
    share the one GPU.  Every cycle and the solve equal the single-process oracle's."""
    ml, b, smr, _ = build_case("gs")
    parts = run_ipc("gs", nranks, env_extra={"AMG_DIST_GS_MODE": "exact", "AMG_TUNABLES": "gs_bw=2,gs_bw_rows=64"})
    assert all(p["pipelined"][0] == 1 for p in parts), [p["pipelined"] for p in parts]
    oh = O.OracleHierarchy(ml)
    got = assemble(parts, "cycles")
    for k in range(3):
        xo, _, _ = oh.solve(b, maxiter=k + 1, calculate_residual=False)
        assert rel(got[k], xo) <= 1e-10, (nranks, k)
    xo, ho, _ = oh.solve(b, reltol=1e-10, maxiter=60)
    hist = parts[0]["solve_v_hist"]
    assert len(hist) == len(ho) and np.allclose(hist, ho, rtol=1e-8)
    assert rel(assemble(parts, "solve_v_x"), xo) <= 1e-10


def test_ipc_processes_sor_w_cycle():
    ml, b, smr, _ = build_case("sor_w")
    lc = SH.num_sharded_levels([l.A.m for l in ml.levels] + [ml.final_A.m], 4, smr)
    parts = run_ipc("sor_w", 4)
    want = emulate_sharded_cycles(ml, b, 4, lc, 2, cyc=1)
    got = assemble(parts, "cycles")
    assert rel(got[0], want[0]) <= 1e-10 and rel(got[1], want[1]) <= 1e-10


def test_ipc_dead_rank_releases_the_streams_of_the_others():
    """A rank exits in the middle of a solve: the survivors' streams wait on flags nobody will write; the pid watch
    releases every flag and the call returns AMGH_ESTATE instead of hanging."""
    outs = run_ipc("die", 3, timeout=240, expect_rc=True)
    assert outs[2][0] == 7
    for r in (0, 1):
        rc, out = outs[r]
        assert rc == 0 and f"IPC_GPU_RANK_{r}_SAW_DEAD_PEER" in out, (r, out[-3000:])


def _gpu_count():
    return int(AMG.hip_lib().amgh_device_count())


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("transport", ["rccl", "ipc"])
def test_two_processes_on_two_gpus(transport):
    """On a multi-GPU node: one process per GPU over RCCL (ncclSend / ncclRecv called by the library) and over the IPC
    transport across devices (peer-mapped buffers over xGMI) — the same checks as on one GPU."""
    ml, b, smr, _ = build_case("gs")
    lc = SH.num_sharded_levels([l.A.m for l in ml.levels] + [ml.final_A.m], 2, smr)
    parts = run_ipc("gs", 2, transport=transport, env_extra={"AMG_IPC_DEVICE_OF_RANK": "0,1"})
    want = emulate_sharded_cycles(ml, b, 2, lc, 3)
    got = assemble(parts, "cycles")
    for k in range(3):
        assert rel(got[k], want[k]) <= 1e-10, (transport, k)
    ml, b, smr, _ = build_case("jacobi_overlap")
    parts = run_ipc("jacobi_overlap", 2, transport=transport, env_extra={"AMG_IPC_DEVICE_OF_RANK": "0,1"})
    want = emulate_sharded_cycles(ml, b, 2, SH.num_sharded_levels([l.A.m for l in ml.levels] + [ml.final_A.m], 2, smr), 2)
    got = assemble(parts, "cycles")
    for k in range(2):
        assert rel(got[k], want[k]) <= 1e-10, (transport, k)
