#!/usr/bin/env python3
"""bench.py — V-cycle unknowns/s + fine-level SpMV GB/s (% of HBM peak), 3-D Poisson 256^3.

One "step" = one V-cycle of the hot path (`ldiv!` semantics of preconditioner.jl:12-19: x = 0,
one `__solve!`, multilevel.jl:214-239) over the whole 16.7 M-unknown system, on the hierarchy
built by `ruge_stuben` defaults (symmetric Gauss-Seidel pre/post), with b already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256]
    python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...   (N > 1)

Prints ONE JSON line (rank 0).  See DESIGN.md §Measurement for the definitions.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy reaches


def uniform(n, seed=0):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def spmv_bytes(M_nnz, rows, cols):
    """Algorithmic bytes of one CSR SpMV: nnz*(8+4) + (rows+1)*4 + 8*cols + 8*rows."""
    return M_nnz * 12 + (rows + 1) * 4 + 8 * cols + 8 * rows


def vcycle_bytes(ml, sweeps_per_level):
    """Algorithmic bytes of one V-cycle."""
    total = 0
    for lev in ml.levels:
        n, nc, nnz = lev.A.m, lev.P.n, lev.A.nnz
        total += sweeps_per_level * (nnz * 12 + (n + 1) * 4 + 24 * n)
        total += spmv_bytes(nnz, n, n) + 8 * n
        total += spmv_bytes(lev.R.nnz, nc, n) + 8 * nc
        total += spmv_bytes(lev.P.nnz, n, nc) + 8 * n
    total += 8 * ml.final_A.m ** 2
    return total


def pmc_traffic(N):
    """HBM bytes per fine-level SpMV launch from the committed rocprofv3 PMC passes (FETCH_SIZE with the
    gfx950 half-count correction + WRITE_SIZE, separate passes: profiles/r01_pmc_spmv_traffic.json).
    PMC counters cannot be collected from inside this process, so the figure is the measured one of the
    same kernel on the same matrix; None for any other size."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_spmv_traffic.json")
    if N != 256 or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["hbm_traffic_bytes_per_launch"]


def cpu_baseline(ml, b, budget_s=20.0):
    """The CPU restatement of the reference's `_solve` cycle (oracle/amg_oracle.c), single thread,
    timed on this host on a bounded sample: as many whole V-cycles as fit in ~budget_s (>= 1)."""
    from oracle import oracle as O
    oh = O.OracleHierarchy(ml)
    n = ml.levels[0].A.m
    t0 = time.perf_counter()
    cycles = 0
    while True:
        oh.precond(b)
        cycles += 1
        el = time.perf_counter() - t0
        if el >= budget_s or el / cycles * (cycles + 1) > 1.5 * budget_s:
            break
    return {"value": n * cycles / el, "unit": "unknowns/s", "cores": 1, "kind": "port",
            "sample": f"{cycles} V-cycle(s) (ldiv! semantics) of the same 3-D Poisson hierarchy, n={n}, "
                      f"{el:.1f} s, 1 thread of {os.cpu_count()} host CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=256, help="grid points per axis (256 = the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--light", action="store_true", help="profiling runs: skip the extra smoother timing")
    ap.add_argument("--force-dist", action="store_true", help="run the row-sharded driver even with one rank")
    args = ap.parse_args()

    if args.gpus > 1 or args.force_dist:
        from bench_dist import main_distributed  # row-sharded path (torch.distributed over RCCL)
        return main_distributed(args)

    import amg_amd as AMG
    import __graft_entry__ as g
    g.build(only_missing=True)
    if not AMG.gpu_available():
        raise SystemExit("bench.py: no HIP device visible")

    N = args.size
    t0 = time.perf_counter()
    A = AMG.poisson((N, N, N))
    ml = AMG.ruge_stuben(A)            # defaults: Classical(0.25), RS(), symmetric Gauss-Seidel pre/post
    t_setup = time.perf_counter() - t0
    n = A.m
    t0 = time.perf_counter()
    dev = ml.device()
    t_upload = time.perf_counter() - t0
    lib = dev.lib

    b = uniform(n, 0)
    bd = AMG.DeviceBuffer(n, 0, b)
    zd = AMG.DeviceBuffer(n, 0)

    def step():
        rc = lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        if rc != 0:
            raise RuntimeError(lib.amgh_strerror(rc).decode())

    for _ in range(args.warmup):
        step()
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    lib.amgh_dev_sync(0)
    elapsed = time.perf_counter() - t0
    ms_per_step = 1e3 * elapsed / args.steps
    value = n * args.steps / elapsed

    # dominant kernel of the metric: the fine-level CSR SpMV, timed with HIP events on its own stream
    spmv_ms = dev.bench_op(0, 0, reps=50, warmup=5)
    resid_ms = dev.bench_op(0, 3, reps=50, warmup=5)
    alg = spmv_bytes(A.nnz, n, n)
    achieved = alg / (spmv_ms * 1e-3) / 1e9
    # one presmoother application (fwd+bwd GS) on the fine level; skipped with --light because every
    # dependency level is a launch and rocprofv3's kernel tracing costs ~10 ms per dispatch
    sweep_ms = None if args.light else dev.bench_op(0, 4, reps=3, warmup=1)

    vb = vcycle_bytes(ml, 4)
    out = {
        "metric": f"V-cycle unknowns/sec + fine-level SpMV GB/s (% HBM peak), 3-D Poisson {N}^3",
        "value": value, "unit": "unknowns/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "poisson((%d,%d,%d)) 7-point, ruge_stuben defaults (theta=0.25, symmetric "
                               "Gauss-Seidel pre+post), one V-cycle per step (ldiv!), b ~ U[0,1) splitmix64 seed 0"
                               % (N, N, N),
                   "unknowns": n, "nnz": A.nnz, "levels": len(ml),
                   "level_sizes": [l.A.m for l in ml.levels] + [ml.final_A.m],
                   "operator_complexity": round(AMG.operator_complexity(ml), 3),
                   "gs_dependency_levels": [dev.gs_dependency_levels(l) for l in range(len(ml.levels))],
                   "gs_sweep_steps_fwd_bwd": [[dev.gs_sweep_steps(l, False), dev.gs_sweep_steps(l, True)]
                                              for l in range(len(ml.levels))],
                   "parallelism": "1 GPU"},
        "roofline": {"bound": "hbm", "kernel": "csr_stream_kernel<SPMV, StreamCfg<1024,1024,8192,4>> (fine-level A, %d rows, %d nnz)" % (n, A.nnz),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(N), "algorithmic_bytes": alg, "avg_launch_ms": spmv_ms,
                     "fused_residual_ms": resid_ms,
                     "fused_residual_GBs": (alg + 8 * n) / (resid_ms * 1e-3) / 1e9},
        "vcycle": {"algorithmic_bytes": vb, "achieved_GBs": vb / (ms_per_step * 1e-3) / 1e9,
                   "fine_symmetric_gs_ms": sweep_ms},
        "setup_s": t_setup, "upload_s": t_upload, "hbm_bytes": dev.device_bytes(),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ml, b, args.cpu_budget)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
