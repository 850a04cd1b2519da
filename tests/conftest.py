import os
import sys

import numpy as np
import pytest

# Virtual ranks (tests of the row-sharded path with N ranks as threads of ONE process on ONE device) need the ranks' sweep
# streams to run side by side: the HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 by
# default), round-robin at creation, and two ranks on one queue run their kernels one after the other.  The library finds
# that out at amgh_dist_finalize and sweeps in turns (amgh_dist_pipe_serialized; the tests then skip their pipelined part) —
# with 8 queues the suite's streams do not collide and the pipelined sweeps really run.  Read when the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# torch bundles its own ROCm runtime (libamdhip64.so.7 / libhsa-runtime64): it has to be the first HIP
# runtime loaded into a process that will use torch.cuda, otherwise torch's device init fails.  The
# sharded-driver tests use torch tensors, so load it before libamghip.so pulls in /opt/rocm's copy.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """Tests never compile in the timed path: make sure the in-tree .so files exist."""
    import __graft_entry__ as g
    g.build(only_missing=True)
    # The library's default row sum in the relayed Gauss-Seidel walk is the DEPENDENCY-AWARE one (csrc/hip/gs_relay.hpp, LATE:
    # the far half of a row summed above the hand-over — the same iterate, one reassociation per row).  Most of this suite pins
    # more than the iterate: BITS — relayed = single walker = chained = launched kernels = host plan = the oracle's scalar loop,
    # blocks of right-hand sides = their single columns, the pipelined sharded sweep = the turn loop.  Those comparisons need the
    # stored-order sum on every path, so the suite runs with gs_bw_inorder = 1; tests/test_gpu_late.py (and bench.py, smoke())
    # run the shipping default against the oracle.
    import amg_amd as AMG
    for dt in ("float64", "float32"):
        try:
            AMG.hip_lib(dt).amgh_debug_set_tunable(b"gs_bw_inorder", 1)
            # (likewise the single-wave walk of small operators: one lane per row = the scalar loop's bits; the shipping
            # default — four lanes per row, gs_waveq_kernel — runs in tests/test_gpu_waveq.py, bench.py and smoke())
            AMG.hip_lib(dt).amgh_debug_set_tunable(b"gs_wave_quad", 0)
            # (and the collapsed coarse tail — one dense operator for the small levels, the same linear map in another rounding:
            # the shipping default runs in tests/test_gpu_tail.py, bench.py and smoke())
            AMG.hip_lib(dt).amgh_debug_set_tunable(b"tail_dense_rows", 0)
        except Exception:  # noqa: BLE001  (no library: the tests that need it fail on their own)
            pass


def load_csc(name):
    import amg_amd as AMG
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return AMG.SparseMatrixCSC.from_arrays(int(d["m"]), int(d["n"]), d["colptr"], d["rowval"], d["nzval"])


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def uniform(n, seed=0):
    """U[0,1) from a self-contained splitmix64 stream (state_k = seed + k*0x9E3779B97F4A7C15)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
