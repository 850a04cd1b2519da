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


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def test_bench_gpus_2_starts_by_itself_and_finishes():
    """`python bench.py --gpus 2`, started the way the driver starts `--gpus 1` (no launcher, no RANK / WORLD_SIZE): bench.py
    spawns its two ranks, they rendezvous over gloo, run the library's own sharded cycle (here in host memory: --host-exec, the
    launcher's self-test on a box without GPUs), rank 0 checks the assembled result against the oracle, and the parent process
    prints rank 0's ONE JSON line and returns 0 — no hang (multilevel.jl:214-239 is the cycle)."""
    import json
    import time
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "24", "--steps", "2", "--warmup", "1",
                        "--host-exec", "--cpu-budget", "0.5"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                   # stdout is the line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["steps"] == 2 and d["launcher"].startswith("self")
    assert d["parity"]["ok"] and d["parity"]["rel_err"] <= 1e-10
    assert d["config"]["sharded_levels"] == 2 and d["config"]["host_execution"]
    assert d["config"]["halo_exchanges_per_cycle"] > 0
    assert time.perf_counter() - t0 < 300


def test_bench_gpus_3_host_exec_three_ranks():
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--size", "20", "--steps", "1", "--warmup", "0",
                        "--host-exec", "--no-cpu-baseline"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads(r.stdout.decode().strip())
    assert d["n_gpus"] == 3 and d["parity"]["ok"]


def test_bench_rendezvous_is_bounded_and_names_the_launcher():
    """A rank whose peers never arrive (RANK / WORLD_SIZE set by a caller who launched nobody else) fails within the rendezvous
    timeout with a message that says how to launch — it does not wait for ever."""
    import socket
    import time
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "16", "--host-exec"],
                       env=_clean_env(RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AMGH_RENDEZVOUS_TIMEOUT_S="5"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert time.perf_counter() - t0 < 120
    err = r.stderr.decode()
    assert "found no peers" in err and "python bench.py --gpus 2" in err, err[-2000:]
    assert r.stdout.decode().strip() == ""


def test_bench_self_launch_fails_fast_when_a_rank_dies():
    """Without --host-exec on a box without GPUs every rank fails at its first look for a device: the parent returns non-zero
    promptly (a rank that dies takes the others with it after the grace period) and prints no line."""
    import time
    if AMG.gpu_available():
        import pytest
        pytest.skip("a GPU is visible: the ranks would run")
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "16", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=_clean_env(AMGH_BENCH_GRACE_S="5"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and r.stdout.decode().strip() == ""
    assert time.perf_counter() - t0 < 120
