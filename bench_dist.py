#!/usr/bin/env python3
"""bench_dist.py — N > 1 leg of bench.py: the 3-D Poisson V-cycle row-sharded over N MI355X
(one process per GPU, RCCL over xGMI through torch.distributed).  STRONG scaling: the 256^3
problem is fixed, each rank owns 1/N of the fine rows; coarse levels are collapsed onto rank 0.

Launched by:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                  --master-port P bench.py --gpus N --steps K --warmup W
"""
import json
import os
import time

import numpy as np


def main_distributed(args):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    import __graft_entry__ as g
    import amg_amd as AMG
    from bench import spmv_bytes, uniform
    dist_mod = __import__("amg_amd").dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # AMG_DIST_BACKEND=gloo + AMG_DIST_ONE_GPU=1: functional test of the multi-process path on a box with a
    # single GPU (every rank on cuda:0, collectives staged through the host); never used for numbers
    backend = os.environ.get("AMG_DIST_BACKEND", "nccl")
    if os.environ.get("AMG_DIST_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    if rank == 0:
        g.build(only_missing=True)
    dist.barrier()
    # the library's default is min(cores, cgroup CPU quota, 64): split that between the ranks of this node
    L = AMG.setup_lib()
    nthreads = max(1, L.amgs_set_threads(0) // int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    L.amgs_set_threads(nthreads)
    # host threads of the smoother-schedule builds in libamghip (merged-level composite rows)
    os.environ.setdefault("AMGH_BUILD_THREADS", str(nthreads))

    N = args.size
    t0 = time.perf_counter()
    A = AMG.poisson((N, N, N))
    ml = AMG.ruge_stuben(A)      # replicated deterministic setup: no communication needed to shard it
    t_setup = time.perf_counter() - t0
    n = A.m
    comm = dist_mod.TorchComm()
    ops = dist_mod.HipOps(local_rank)
    t0 = time.perf_counter()
    dml = dist_mod.DistMultiLevel(ml, comm, ops)
    t_shard = time.perf_counter() - t0
    r0, r1 = dml.local_range(0)
    b = uniform(n, 0)
    dml.set_rhs(b[r0:r1])

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        dml.precond_apply(0)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dml.precond_apply(0)
    sync()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # fine-level sharded SpMV (halo all-gather + local rows), timed the same way
    d = dml.levels[0] if dml.lc > 0 else None
    spmv_ms = None
    if d is not None:
        for _ in range(3):
            dml.exchange("x", 0, dml.x[0]); ops.spmv(d["A"], dml.x[0], d["res"])
        sync()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            dml.exchange("x", 0, dml.x[0]); ops.spmv(d["A"], dml.x[0], d["res"])
        sync()
        sp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(sp, op=dist.ReduceOp.MAX)
        spmv_ms = 1e3 * float(sp.item()) / reps

    if rank == 0:
        alg = spmv_bytes(A.nnz, n, n)
        out = {
            "metric": f"V-cycle unknowns/sec + fine-level SpMV GB/s (% HBM peak), 3-D Poisson {N}^3",
            "value": n * args.steps / elapsed, "unit": "unknowns/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "poisson((%d,%d,%d)) 7-point, ruge_stuben defaults, one V-cycle per step (ldiv!), "
                                   "fine rows 1-D row-sharded, halo all-gather before every operator, "
                                   "levels below %d rows collapsed to rank 0; Gauss-Seidel is processor-block "
                                   "hybrid across shards" % (N, N, N, 200000),
                       "unknowns": n, "nnz": A.nnz, "levels": len(ml), "sharded_levels": dml.lc,
                       "parallelism": f"row-shard x{world} (RCCL all-gather halos)"},
            "roofline": None if spmv_ms is None else {
                "bound": "hbm", "kernel": "csr_stream_kernel<SPMV> on n/N local rows + halo all-gather",
                "achieved": alg / (spmv_ms * 1e-3) / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                "frac": alg / (spmv_ms * 1e-3) / 1e9 / (8000.0 * world), "traffic": None,
                "avg_launch_ms": spmv_ms},
            "setup_s": t_setup, "shard_s": t_shard,
        }
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
