"""Worker of tests/test_gpu_ipc.py: ONE rank (= one process) of the row-sharded cycle through libamghip's
`amgh_dist_*` C ABI over the IPC transport (peer-mapped send buffers + stream-written flags in shared memory).
Every rank of a test shares GPU 0 of the box (hipIpc allows it; RCCL does not), so the never-before-executed
N > 1 multi-process exchange path runs on the single-GPU test box.  Results go to OUTDIR/rank<r>.npz; the parent
test compares them with the oracle / the frozen-halo emulation.

    python ipc_gpu_worker.py RANK NRANKS /shm_name OUTDIR CASE
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import amg_amd as AMG  # noqa: E402
from amg_amd import sharded as SH  # noqa: E402
from ipc_cases import build_case  # noqa: E402


def main():
    rank, nranks, name, outdir, case = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    ml, b, shard_min_rows, plan = build_case(case)
    if name.startswith("rccl:"):
        # RCCL over real GPUs (one per rank): rank 0 publishes the 128-byte id through a file of the output directory
        import time
        path = os.path.join(outdir, "rccl_id.bin")
        if rank == 0:
            with open(path + ".tmp", "wb") as f:
                f.write(SH.rccl_unique_id())
            os.rename(path + ".tmp", path)
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise SystemExit("no RCCL id from rank 0")
            time.sleep(0.05)
        transport, device = ("rccl", open(path, "rb").read()), rank
    else:
        transport, device = ("ipc", name), int(os.environ.get("AMG_IPC_DEVICE_OF_RANK", "0").split(",")[rank % len(os.environ.get("AMG_IPC_DEVICE_OF_RANK", "0").split(","))])
    for kv in filter(None, os.environ.get("AMG_TUNABLES", "").split(",")):   # e.g. gs_bw=2,gs_bw_rows=64: block layouts on small shards
        k, v = kv.split("=")
        assert AMG.hip_lib().amgh_debug_set_tunable(k.encode(), int(v)) == 0
    sh = SH.ShardedHierarchy.from_multilevel(ml, rank, nranks, device, transport, shard_min_rows, gs_mode=os.environ.get("AMG_DIST_GS_MODE", "hybrid"))
    bl = b[sh.r0:sh.r1]
    out = {"r0": sh.r0, "r1": sh.r1, "lc": sh.lc, "pipelined": np.array(sh.gs_pipelined(), dtype=np.int32)}
    if case == "die":
        # rank nranks-1 vanishes after one cycle; the others are inside a 40-cycle solve whose streams wait for flags
        # only the dead rank would write: they must come back with AMGH_ESTATE, not hang
        if rank == nranks - 1:
            sh.solve(bl, maxiter=1, calculate_residual=False)
            os._exit(7)
        try:
            sh.solve(bl, maxiter=1, calculate_residual=False)
            sh.solve(bl, maxiter=40, reltol=1e-300)
            print(f"IPC_GPU_RANK_{rank}_UNEXPECTED_SUCCESS", flush=True)
        except AMG.AMGError as e:
            assert "invalid state" in str(e), str(e)
            print(f"IPC_GPU_RANK_{rank}_SAW_DEAD_PEER", flush=True)
        os._exit(0)   # the handle is broken: no collective teardown
    for key, kw in plan:
        if key.startswith("cycles"):
            # iterates after exactly k cycles from x0 = 0
            xs = []
            for k in range(1, kw["cycles"] + 1):
                x, _ = sh.solve(bl, cycle=kw.get("cycle", 0), maxiter=k, calculate_residual=False)
                xs.append(x)
            out[key] = np.stack(xs)
        elif key.startswith("solve"):
            x, hist = sh.solve(bl, **kw)
            out[key + "_x"], out[key + "_hist"] = x, hist
        elif key == "ldiv":
            out["ldiv"] = sh.precond_apply(bl, 0)
        elif key == "spmv":
            out["spmv"] = sh.spmv(0, bl)
    st = sh.stats()
    out["halo_exchanges"], out["halo_bytes_sent"] = st["halo_exchanges"], st["halo_bytes_sent"]
    # many exchanges back to back (every plan's two send-buffer copies and its done flags get reused)
    for _ in range(30):
        sh.precond_apply_d(0)
    sh.sync()
    out["repeat"] = sh._down(sh._x)
    v = sh.allreduce([float(rank + 1)], "sum")
    assert v[0] == nranks * (nranks + 1) / 2
    sh.barrier()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    sh.close()
    print(f"IPC_GPU_RANK_{rank}_OK", flush=True)


if __name__ == "__main__":
    main()
