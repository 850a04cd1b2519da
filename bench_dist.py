#!/usr/bin/env python3
"""bench_dist.py — N > 1 leg of bench.py: the 3-D Poisson V-cycle row-sharded over N MI355X, one process per GPU,
through libamghip's `amgh_dist_*` C ABI (RCCL over xGMI called by the library itself: neighbour send/recv of halo
entries, coarse levels collapsed onto rank 0).  STRONG scaling: the 256^3 problem is fixed, each rank owns 1/N of
the rows of every sharded level.

Launched by:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                  --master-port P bench.py --gpus N --steps K --warmup W

torch is used for ONE thing here: the rendezvous the launcher already provides (a gloo group on CPU to hand the
128-byte RCCL id from rank 0 to the others and to find out who built the hierarchy).  It never touches the GPU; the
timed barriers and the max-over-ranks are the library's own (stream sync + RCCL all-reduce).

The host hierarchy is built ONCE per node (rank 0), the level matrices of the sharded levels go to /dev/shm as
.npy files and every other rank maps them and reads only its own rows.
"""
import json
import os
import shutil
import tempfile
import time

import numpy as np


def node_levels(rank, world, bcast, build):
    """The host hierarchy ONCE per node: rank 0 runs `build()` (-> MultiLevel), decides how many levels are sharded,
    and — for world > 1 — exports the sharded levels' matrices as .npy files; every other rank maps them and will read
    only its own rows.  `bcast(obj)` broadcasts a small Python object from rank 0.  Returns (levels, info, tail, dir):
    `tail` (the collapsed levels as a MultiLevel) on rank 0 only."""
    import amg_amd as AMG
    from amg_amd import sharded as SH
    shm = None
    tail = None
    levels = None
    if rank == 0:
        ml = build()
        A = ml.levels[0].A if ml.levels else ml.final_A
        sizes = [l.A.m for l in ml.levels] + [ml.final_A.m]
        lc = SH.num_sharded_levels(sizes, world)
        levels = SH.level_arrays(ml, lc)
        tail = AMG.MultiLevel(ml.levels[lc:], ml.final_A, ml.coarse_solver, ml.presmoother, ml.postsmoother,
                              ml.symmetry, method=ml.method)
        info = dict(n_tail=sizes[lc], nnz=A.nnz, nlev=len(ml), lc=lc, sizes=sizes)
        if world > 1:
            # shared memory if it has the room (a container's /dev/shm can be as small as 64 MB), else the temp directory
            need = sum(a.nbytes for d in levels for key in ("A", "S", "P", "R") if d[key] is not None for a in d[key])
            base = tempfile.gettempdir()
            try:
                if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 1.25 * need:
                    base = "/dev/shm"
            except OSError:
                pass
            shm = tempfile.mkdtemp(prefix="amgh_levels_", dir=base)
            SH.export_levels(levels, shm)
        info["shm"] = shm
    else:
        info = None
    info = bcast(info)
    if rank != 0:
        levels = SH.load_levels(info["shm"])
    return levels, info, tail, shm


def main_distributed(args):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # schedule builds of the shards run on every rank at once: share the host's cores between the local ranks
    os.environ.setdefault("AMGH_BUILD_THREADS", str(max(2, min(32, (os.cpu_count() or 8) // max(1, local_world)))))

    import torch.distributed as dist  # rendezvous only (CPU / gloo)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def bcast(obj):
        if world == 1:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    import __graft_entry__ as g
    if rank == 0:
        g.build(only_missing=True)
    if world > 1:
        dist.barrier()
    import amg_amd as AMG
    from amg_amd import sharded as SH
    from bench import spmv_bytes, uniform

    if os.environ.get("AMG_DIST_ONE_GPU") == "1":   # functional check on a single-GPU box: every rank on device 0
        local_rank = 0
    N = args.size
    n = N ** 3
    # ---- host hierarchy: once per node --------------------------------------------------------------------------
    t0 = time.perf_counter()
    levels, info, tail, shm = node_levels(rank, world, bcast,
                                          lambda: AMG.ruge_stuben(AMG.poisson((N, N, N)), setup=getattr(args, "setup", "gpu")))
    t_setup = time.perf_counter() - t0
    # ---- the sharded handle -------------------------------------------------------------------------------------
    t0 = time.perf_counter()
    uid = bcast(SH.rccl_unique_id() if rank == 0 else None)
    sh = SH.ShardedHierarchy(levels, info["n_tail"], tail, rank, world, local_rank, ("rccl", uid))
    t_shard = time.perf_counter() - t0
    sh.barrier()
    if rank == 0 and shm:
        shutil.rmtree(shm, ignore_errors=True)
    b = uniform(n, 0)
    sh.set_rhs(b[sh.r0:sh.r1])

    for _ in range(args.warmup):
        sh.precond_apply_d(0)
    sh.barrier()
    sh.stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sh.precond_apply_d(0)
    sh.barrier()
    elapsed = float(sh.allreduce([time.perf_counter() - t0], "max")[0])
    st = sh.stats(reset=True)
    ex_per_cycle = st["halo_exchanges"] / max(1, args.steps)
    halo = sh.allreduce([st["halo_bytes_sent"] / max(1, args.steps)], "max")[0]
    halo_sum = sh.allreduce([st["halo_bytes_sent"] / max(1, args.steps)], "sum")[0]

    # fine-level sharded SpMV (neighbour exchange + local rows), timed the same way
    spmv_ms = None
    if sh.lc > 0:
        y = AMG.DeviceBuffer(max(sh.nloc, 1), local_rank)
        lib = sh.lib
        for _ in range(3):      # x = the level's resident vector (the last cycle's result): exchange + local rows, no copy
            lib.amgh_dist_spmv_d(sh.h, 0, None, y.ptr)
        sh.barrier()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            lib.amgh_dist_spmv_d(sh.h, 0, None, y.ptr)
        sh.barrier()
        spmv_ms = 1e3 * float(sh.allreduce([time.perf_counter() - t0], "max")[0]) / reps

    # parity of the timed output (outside the timed region): global residual reduction of ONE cycle must match the
    # single-GPU / oracle figure to the digits the hybrid smoother allows; here only sanity (finite, contracting)
    z_norm2 = sh.allreduce([float(np.sum(sh._down(sh._x) ** 2))], "sum")[0]

    if rank == 0:
        alg = spmv_bytes(info["nnz"], n, n)
        out = {
            "metric": f"V-cycle unknowns/sec + fine-level SpMV GB/s (% HBM peak), 3-D Poisson {N}^3",
            "value": n * args.steps / elapsed, "unit": "unknowns/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "poisson((%d,%d,%d)) 7-point, ruge_stuben defaults, one V-cycle per step (ldiv!), "
                                   "rows of every level >= 200000 rows 1-D row-sharded, neighbour send/recv of halo "
                                   "entries before every operator (RCCL called by libamghip, interior rows overlapped), "
                                   "coarser levels collapsed to rank 0; Gauss-Seidel is exact inside a shard, halo frozen "
                                   "per directional sweep" % (N, N, N),
                       "unknowns": n, "nnz": info["nnz"], "levels": info["nlev"], "sharded_levels": info["lc"],
                       "halo_exchanges_per_cycle": ex_per_cycle, "halo_bytes_sent_per_cycle_max_rank": halo,
                       "halo_bytes_sent_per_cycle_all_ranks": halo_sum,
                       "parallelism": f"row-shard x{world} (RCCL send/recv halos, libamghip amgh_dist_*)"},
            "roofline": None if spmv_ms is None else {
                "bound": "hbm", "kernel": "csr_stream_kernel<SPMV> on n/N local rows + neighbour halo exchange",
                "achieved": alg / (spmv_ms * 1e-3) / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                "frac": alg / (spmv_ms * 1e-3) / 1e9 / (8000.0 * world), "traffic": None,
                "avg_launch_ms": spmv_ms},
            "check": {"z_norm": float(np.sqrt(z_norm2)), "finite": bool(np.isfinite(z_norm2))},
            "setup_s": t_setup, "shard_s": t_shard,
        }
    sh.barrier()
    sh.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        import sys
        sys.stderr.flush()
        print(json.dumps(out), flush=True)   # the ONE JSON line, after everything RCCL may print
