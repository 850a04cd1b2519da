"""N > 1 path on CPU: world_size-2 gloo run of the row-sharded driver (partitioning, halo
all-gather, collapse to rank 0) with the test-side CPU backend, checked against the oracle."""
import os
import subprocess
import sys

import numpy as np

import amg_amd as AMG
import dist_mirror as D  # noqa: E402
from conftest import ROOT


def test_row_ranges_and_vector_plan():
    assert D.row_ranges(10, 3) == [(0, 3), (3, 6), (6, 10)]
    A = AMG.poisson((4, 4, 6))
    rp, ci, va = A.csr_arrays()
    ranges = D.row_ranges(A.m, 2)
    needs = D._needs(rp, ci, ranges, ranges)
    # z-slab partition of a 7-point stencil: each rank needs exactly the neighbour's boundary plane
    assert np.array_equal(needs[0], np.arange(48, 64)) and np.array_equal(needs[1], np.arange(32, 48))
    p0 = D.VectorPlan(ranges, 0, needs)
    p1 = D.VectorPlan(ranges, 1, needs)
    assert p0.max_send == p1.max_send == 16 and p0.total_send == 32
    assert np.array_equal(p0.send_idx, np.arange(32, 48)) and np.array_equal(p1.send_idx, np.arange(0, 16))
    assert np.array_equal(p0.unpack_idx, 16 + np.arange(16)) and np.array_equal(p1.unpack_idx, np.arange(16))
    loc = p1.localize(np.array([48, 95, 32, 47]))
    assert loc.tolist() == [0, 47, 48, 63]


def test_gloo_world_size_2():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "DIST_WORKER_OK" in out, out[-3000:]


def test_gloo_drives_the_library_sharded_cycle_on_the_host():
    """`amgh_dist_*` ITSELF — not a mirror of it — through whole V-cycles without a GPU: world_size 2 and 3 launched over gloo,
    the library executing its sharded cycle in host memory (device = -1 with a host tail): halo plans, exchange ordering,
    exact Gauss-Seidel in turns, the hybrid, Jacobi, collapse onto rank 0, the all-reduced residual norms — against the
    oracle at 1e-10 (tests/dist_host_worker.py)."""
    for nranks, port in ((2, "29541"), (3, "29543")):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", AMGH_IPC_TIMEOUT_S="120")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "dist_host_worker.py")]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, cwd=ROOT)
        out = r.stdout.decode()
        assert r.returncode == 0 and "DIST_HOST_WORKER_OK" in out, (nranks, out[-4000:])


def test_node_levels_gloo():
    """bench_dist.py's once-per-node hand-over of the host hierarchy (export on rank 0, memory-mapped row slices on the
    others), world_size 2 over gloo."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29537", os.path.join(ROOT, "tests", "node_levels_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "NODE_LEVELS_OK" in out, out[-3000:]


def _spawn_ipc_workers(nranks, extra=()):
    name = "/amgh_t_%d_%s" % (os.getpid(), os.urandom(4).hex())
    env = dict(os.environ, OMP_NUM_THREADS="2", AMGH_IPC_TIMEOUT_S="60")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_plan_worker.py"), str(r), str(nranks),
                               name, *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT)
             for r in range(nranks)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out.decode()))
    return outs


def test_ipc_plans_two_and_three_processes():
    """`amgh_dist_*` itself (not a mirror of it) through the collective setup with real processes: halo needs exchanged
    over the IPC transport's shared-memory rendezvous, plans, interior ranges, collapse, host all-reduce / barrier —
    no GPU (device = -1 builds the plans only)."""
    for nranks in (2, 3):
        outs = _spawn_ipc_workers(nranks)
        for r, (rc, out) in enumerate(outs):
            assert rc == 0 and f"IPC_PLAN_RANK_{r}_OK" in out, (nranks, r, out[-3000:])


def test_ipc_dead_peer_does_not_hang_the_others():
    """A rank that dies between two collectives: the survivors return AMGH_ESTATE (pid watch) instead of waiting."""
    outs = _spawn_ipc_workers(3, extra=("die",))
    assert outs[2][0] == 7
    for r in (0, 1):
        rc, out = outs[r]
        assert rc == 0 and f"IPC_PLAN_RANK_{r}_SAW_DEAD_PEER" in out, (r, out[-3000:])
