#!/usr/bin/env python3
"""bench.py — V-cycle unknowns/s + fine-level SpMV GB/s (% of HBM peak), 3-D Poisson 256^3.

One "step" = one V-cycle of the hot path (`ldiv!` semantics of preconditioner.jl:12-19: x = 0,
one `__solve!`, multilevel.jl:214-239) over the whole 16.7 M-unknown system, on the hierarchy
built by `ruge_stuben` defaults (symmetric Gauss-Seidel pre/post), with b already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256]
    python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...   (N > 1, a launcher's RANK / WORLD_SIZE)
    python bench.py --gpus N ...        (N > 1, no launcher: bench.py spawns its N ranks itself, `self_launch`)

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


def pmc_traffic(N, enabled=True):
    """HBM bytes of one fine-level SpMV launch, MEASURED IN THIS RUN: two `rocprofv3 --pmc` passes (FETCH_SIZE,
    WRITE_SIZE — they cannot share a pass) over tools/spmv_bench, which launches the shipped kernel on the same
    matrix plus a known-size read / copy that calibrates the gfx950 FETCH_SIZE half-count in the same process
    (tools/pmc_traffic.py; method of MI355X_MICROARCH.md, HBM section).  -> (bytes or None, how)."""
    if not enabled:
        return None, "skipped (--no-pmc)"
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_traffic as P
        r = P.measure(N)
        return r["hbm_traffic_bytes_per_launch"], (
            "measured in this run: %s; FETCH_SIZE %.0f KiB x %.4f + WRITE_SIZE %.0f KiB x %.4f over %d launches"
            % (r["method"], r["FETCH_SIZE_KiB_avg"], r["fetch_correction"], r["WRITE_SIZE_KiB_avg"],
               r["write_correction"], r["launches"]))
    except Exception as e:  # noqa: BLE001  (no rocprofv3 / counters unavailable: the field is null, never a stale constant)
        return None, f"unavailable: {type(e).__name__}: {str(e)[:160]}"


def read_ceiling(N):
    """What this chip, on this box, in this run, READS: tools/spmv_bench's known-size kernels (2 GiB 16-B/lane read, 1 GiB
    copy), timed with HIP events outside any profiler.  The roofline fraction against the datasheet's 8 TB/s moves ~8 %
    from box to box; the fraction of this ceiling is what the kernel makes of the memory system it was given."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "spmv_bench")
    try:
        r = subprocess.run([exe, str(N), "5", "1", "ship"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=180)
        txt = r.stdout.decode(errors="replace")
        rd = re.search(r"read16\s+2GiB:\s+([0-9.]+) ms\s+([0-9.]+) GB/s", txt)
        cp = re.search(r"copy16\s+1GiB->1GiB:\s+([0-9.]+) ms\s+([0-9.]+) GB/s", txt)
        return (float(rd.group(2)) if rd else None), (float(cp.group(2)) if cp else None)
    except Exception:  # noqa: BLE001
        return None, None


def smi_state():
    """Clock / power state of GPU 0 as rocm-smi reports it (best effort; None when the tool is missing)."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showperflevel", "--showtemp", "--json"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=30)
        d = json.loads(r.stdout.decode(errors="replace"))
        card = d.get("card0") or next(iter(d.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "socclk", "power", "performance level", "temperature (sensor junction)",
                                     "temperature (sensor memory)", "temperature (sensor hbm")):
                keep[k] = v
        return keep or None
    except Exception:  # noqa: BLE001
        return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ml, b, budget_s=20.0):
    """The CPU restatement of the reference's `_solve` cycle (oracle/amg_oracle.c, gcc -O3 -march=native), single
    thread — the reference is single-threaded — timed on this host on a bounded sample: as many whole V-cycles as
    fit in ~budget_s (>= 1).  Also returns the oracle's V-cycle output for the parity check of the timed result."""
    from oracle import oracle as O
    oh = O.OracleHierarchy(ml)
    n = ml.levels[0].A.m
    t0 = time.perf_counter()
    cycles = 0
    while True:
        z = oh.precond(b)
        cycles += 1
        el = time.perf_counter() - t0
        if el >= budget_s or el / cycles * (cycles + 1) > 1.5 * budget_s:
            break
    return {"value": n * cycles / el, "unit": "unknowns/s", "cores": 1, "kind": "port",
            "cpu": cpu_model(), "host_cpus": os.cpu_count(),
            "sample": f"{cycles} V-cycle(s) (ldiv! semantics) of the same 3-D Poisson hierarchy, n={n}, "
                      f"{el:.1f} s, 1 thread of {os.cpu_count()} host CPUs ({cpu_model()}); "
                      "CPU restatement of the reference's _solve cycle (Julia is not installed), gcc -O3 -march=native"}, z


def uncompressed_cycle_ms(ml, lib, bd, zd, n, reps=3):
    """What an operator WITHOUT repeating value rows / few distinct values gets (variable coefficients, elasticity): the same
    hierarchy laid out a second time with the dictionary records (tunable gs_bw_dict) and the value-coded columns (stream_code)
    switched off at build — the plain 80 / 208-byte records and 12-byte entries — timed over `reps` cycles.  Both layouts are
    lossless; this is the round-4 data path under this round's kernels."""
    import amg_amd as AMG
    for name in (b"gs_bw_dict", b"stream_code"):
        if lib.amgh_debug_set_tunable(name, 0) != 0:
            raise RuntimeError("tunable %r not accepted" % name)
    try:
        t0 = time.perf_counter()
        dev2 = AMG.DeviceHierarchy(ml, 0, 1)
        t_layout = time.perf_counter() - t0
        for _ in range(2):
            if lib.amgh_precond_apply_d(dev2.h, bd.ptr, zd.ptr, 0) != 0:
                raise RuntimeError("precond_apply failed")
        if lib.amgh_dev_sync(0) != 0:
            raise RuntimeError("amgh_dev_sync failed")
        t0 = time.perf_counter()
        for _ in range(reps):
            lib.amgh_precond_apply_d(dev2.h, bd.ptr, zd.ptr, 0)
        if lib.amgh_dev_sync(0) != 0:
            raise RuntimeError("amgh_dev_sync failed")
        ms = 1e3 * (time.perf_counter() - t0) / reps
        z = zd.download()
        out = {"ms_per_cycle": ms, "unknowns_per_s": n / (ms * 1e-3), "cycles_timed": reps, "layout_s": t_layout, "hbm_bytes": dev2.device_bytes(),
               "dictionary_layout_by_level": [int(lib.amgh_debug_bw_dict(dev2.h, l)) for l in range(len(ml.levels))],
               "value_coded_operators_by_level": [int(lib.amgh_debug_coded_ops(dev2.h, l)) for l in range(len(ml.levels))],
               "what": "the same cycle with gs_bw_dict = 0 and stream_code = 0 at layout: plain records, 12-byte entries (what an operator "
                       "whose value rows do not repeat gets); bitwise the compressed layouts"}
        del dev2
        return out, z
    finally:
        for name in (b"gs_bw_dict", b"stream_code"):
            lib.amgh_debug_set_tunable(name, 1)


def tail_info(dev):
    """The collapsed coarse tail of a device hierarchy (amghip.h: amgh_tail_dense_build): the level from which the V-cycle's
    recursion is applied as one dense operator, its rows, and what building it cost (inside setup_s)."""
    lv, rows, ms = dev.tail_dense_info(0)
    return {"level": lv, "rows": rows, "operator_bytes": 8 * rows * rows if lv >= 0 else 0, "build_ms": ms,
            "note": "levels >= level: smoothers, residual, restriction, recursion, coarse solve, prolongation as ONE dense operator "
                    "(the same linear map, built from the per-level recursion on the columns of the identity; tunable tail_dense_rows)"}


def secondary_configs(check=True):
    """The smaller BASELINE.json configurations on the same build, a few hundred milliseconds each: C1 (poisson(1000), ruge_stuben,
    symmetric Gauss-Seidel), C2 at full size (poisson((1024,1024)), smoothed_aggregation, Jacobi(2/3)), C5 (lin_elastic_2d,
    smoothed aggregation with near-null-space, aspreconditioner in CG).  With `check` (the run has its CPU leg: the oracle is
    the checker there, never the thing timed) each cycle is compared with the oracle's (<= 1e-10) outside its timed loop."""
    import amg_amd as AMG
    lib = AMG.hip_lib()
    out = {}

    def cycles(ml, n, reps):
        dev = ml.device()
        b = uniform(n, 3)
        bd = AMG.DeviceBuffer(n, 0, b)
        zd = AMG.DeviceBuffer(n, 0)
        best = 1e9
        for _ in range(5):   # (tiny cycles: the clocks only ramp up under sustained load — best of a few rounds behind 200 warm-up cycles)
            for _ in range(200 if n < 100000 else 3):
                lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
            lib.amgh_dev_sync(0)
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
            if lib.amgh_dev_sync(0) != 0:
                raise RuntimeError("amgh_dev_sync failed")
            best = min(best, 1e3 * (time.perf_counter() - t0) / reps)
        err = None
        if check:
            from oracle import oracle as O
            z = zd.download()
            zo = O.OracleHierarchy(ml).precond(b)
            err = float(np.linalg.norm(z - zo) / np.linalg.norm(zo))
            if not err <= 1e-10:
                raise RuntimeError(f"cycle differs from the oracle: {err:.3e}")
        return best, err, dev

    try:
        A = AMG.poisson(1000)
        ml = AMG.ruge_stuben(A)
        ms, err, dev = cycles(ml, 1000, 50)
        out["C1"] = {"workload": "poisson(1000), ruge_stuben defaults", "vcycle_ms": ms, "levels": len(ml), "rel_err_vs_oracle": err,
                     "collapsed_tail": {k: v for k, v in tail_info(dev).items() if k != "note"}}
    except Exception as ex:  # noqa: BLE001
        out["C1"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    try:
        A = AMG.poisson((1024, 1024))
        n = A.m
        jac = AMG.Jacobi(2.0 / 3.0)
        t0 = time.perf_counter()
        ml = AMG.smoothed_aggregation(A, presmoother=jac, postsmoother=jac)
        ts = time.perf_counter() - t0
        ms, err, dev = cycles(ml, n, 20)
        sp = dev.bench_op(0, 0, reps=50, warmup=5)
        alg = spmv_bytes(A.nnz, n, n)
        out["C2"] = {"workload": "poisson((1024,1024)), smoothed_aggregation, Jacobi(2/3) pre+post", "unknowns": n, "setup_s": ts,
                     "vcycle_ms": ms, "unknowns_per_s": n / (ms * 1e-3), "levels": len(ml), "rel_err_vs_oracle": err,
                     "fine_spmv_ms": sp, "fine_spmv_GBs": alg / (sp * 1e-3) / 1e9,
                     "collapsed_tail": {k: v for k, v in tail_info(dev).items() if k != "note"},
                     "note": "the fine operator (%.0f MB) fits the 256 MB Infinity Cache: the SpMV rate is not an HBM figure" % (alg / 1e6)}
    except Exception as ex:  # noqa: BLE001
        out["C2"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    try:
        d = np.load(os.path.join(ROOT, "tests", "golden", "lin_elastic_2d.npz"))
        A = AMG.SparseMatrixCSC.from_arrays(int(d["m"]), int(d["n"]), d["colptr"], d["rowval"], d["nzval"])
        ml = AMG.smoothed_aggregation(A, B=d["B"])
        pl = AMG.aspreconditioner(ml)
        best, iters = 1e9, None
        for _ in range(4):
            t0 = time.perf_counter()
            x, log = AMG.cg(A, d["b"], Pl=pl, reltol=1e-10, log=True)
            best = min(best, 1e3 * (time.perf_counter() - t0))
            iters = int(log["iters"])
        ms, err, dev = cycles(ml, A.m, 50)
        res = float(np.linalg.norm(d["b"] - A @ x) / np.linalg.norm(d["b"]))
        out["C5"] = {"workload": "lin_elastic_2d (n = %d), smoothed_aggregation with B, aspreconditioner in cg, reltol 1e-10" % A.m,
                     "pcg_ms": best, "pcg_iterations": iters, "final_rel_residual": res, "vcycle_ms": ms, "rel_err_vs_oracle": err}
    except Exception as ex:  # noqa: BLE001
        out["C5"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    return out


def self_launch(nranks):
    """`python bench.py --gpus N` with N > 1 and no launcher's environment (RANK / WORLD_SIZE absent): spawn the N ranks — this
    same command line, one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run
    would — forward rank 0's ONE JSON line as this process's stdout and return the worst child's exit code.  A rank that dies
    takes the others with it after a grace period (they would wait for it in a collective for ever); children are killed by PID."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                AMGH_BENCH_LAUNCHER="self")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(nranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout carries the line; the other ranks' stdout (nothing, by construction) joins stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), stderr=None))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    grace = float(os.environ.get("AMGH_BENCH_GRACE_S", "30"))
    first_bad = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and first_bad is None:
            first_bad = time.perf_counter()
            print(f"bench.py: rank {procs.index(bad[0])} exited with code {bad[0].returncode}; the other ranks get {grace:.0f} s", file=sys.stderr, flush=True)
        if first_bad is not None and time.perf_counter() - first_bad > grace:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    reader.join(timeout=10)
    rcs = [p.returncode for p in procs]
    text = (out0[0] if out0 else b"").decode(errors="replace")
    line = None
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                json.loads(ln)
                line = ln
            except ValueError:
                pass
    for ln in text.splitlines():                      # anything else rank 0 wrote to stdout is not the line: stderr
        if ln.strip() and ln.strip() != line:
            print(ln, file=sys.stderr)
    worst = max((abs(rc) if rc is not None else 1) for rc in rcs)
    if line is not None and worst == 0:
        print(line, flush=True)
        return 0
    print(f"bench.py: self-launched run failed: exit codes by rank {rcs}" + ("" if line else "; rank 0 printed no JSON line"), file=sys.stderr, flush=True)
    return worst or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=256, help="grid points per axis (256 = the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--light", action="store_true", help="profiling runs: skip the extra smoother timing")
    ap.add_argument("--no-block-rhs", action="store_true", help="skip the secondary measurement on a block of 8 right-hand sides")
    ap.add_argument("--no-uncompressed", action="store_true", help="skip the secondary cycle on the uncompressed layouts (a second layout of the hierarchy: ~6 s, ~24 GB)")
    ap.add_argument("--no-secondary-configs", action="store_true", help="skip the C1 / C2 / C5 cycles")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--force-dist", action="store_true", help="run the row-sharded driver even with one rank")
    ap.add_argument("--transport", default=None, choices=("rccl", "ipc", "ipc-staged"),
                    help="N > 1: halo transport of the primary measurement (default: AMGH_DIST_TRANSPORT or rccl)")
    ap.add_argument("--smoother", default="gs", choices=("gs", "jacobi"),
                    help="N > 1: smoother of the primary measurement (gs = ruge_stuben defaults; jacobi = Jacobi(2/3))")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1: skip the 32^3 two-cycle exchange that tries every transport before the timed problem")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N > 1: skip the secondary measurements (Jacobi-smoothed hierarchy, IPC transport)")
    ap.add_argument("--setup", default="gpu", choices=("gpu", "host"), help="where the data-parallel half of ruge_stuben runs")
    ap.add_argument("--host-exec", action="store_true",
                    help="N > 1 launcher self-test on a box without GPUs: the library's own sharded cycle (amgh_dist_*) executed in host "
                         "memory over the shared-memory transport, checked against the oracle — functional, never a measurement")
    ap.add_argument("--no-overlap", action="store_true",
                    help="with --setup gpu: build the HBM hierarchy after ruge_stuben instead of level by level beside it")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)   # started the way `--gpus 1` is started: no launcher — be the launcher
    if args.gpus > 1 or args.force_dist or args.host_exec:
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
    t_problem = time.perf_counter() - t0      # (generating the operator: the caller's matrix, not hierarchy setup — reported apart, still inside setup_s)
    # defaults: Classical(0.25), RS(), symmetric Gauss-Seidel pre/post; strength / interpolation / R*A*P on the GPU
    # (bitwise the host library's hierarchy, tests/test_gpu_setup.py), the sequential C/F splitting on the host
    # — and, beside that splitting, the previous step of the pipeline: each level's upload + smoother schedules (they
    # need only that level's A), so that setup_s already contains most of what upload_s used to be
    overlap = args.setup == "gpu" and not args.no_overlap
    overlap_error = None
    try:
        ml = AMG.ruge_stuben(A, setup=args.setup, device=0 if overlap else None)
    except AMG.AMGError as e:       # the pipeline is an optimisation of the untimed part: never let it cost the run
        if not overlap:
            raise
        overlap, overlap_error = False, str(e)
        t0 = time.perf_counter() - t_problem
        ml = AMG.ruge_stuben(A, setup=args.setup)
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

    def sync():
        # the synchronising entry point is where a give-up of the dataflow sweeps' bounded polls surfaces (AMGH_ESTATE):
        # a timed region that ran on stale values fails the run here, before anything is reported
        rc = lib.amgh_dev_sync(0)
        if rc != 0:
            raise SystemExit("bench.py: amgh_dev_sync: " + lib.amgh_strerror(rc).decode())

    clocks_before = smi_state()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    clocks_after = smi_state()
    ms_per_step = 1e3 * elapsed / args.steps
    value = n * args.steps / elapsed
    z_timed = zd.download()      # what the LAST timed V-cycle produced (checked against the oracle below)

    # dominant kernel of the metric: the fine-level CSR SpMV, timed with HIP events on its own stream
    spmv_ms = dev.bench_op(0, 0, reps=50, warmup=5)
    resid_ms = dev.bench_op(0, 3, reps=50, warmup=5)
    alg = spmv_bytes(A.nnz, n, n)
    achieved = alg / (spmv_ms * 1e-3) / 1e9
    read_GBs, copy_GBs = (None, None) if args.light else read_ceiling(N)
    # one presmoother application (fwd+bwd GS) on the fine level; skipped with --light because every
    # dependency level is a launch and rocprofv3's kernel tracing costs ~10 ms per dispatch
    sweep_ms = None if args.light else dev.bench_op(0, 4, reps=3, warmup=1)
    # what the sweeps stream per cycle as executed (composite rows of the merged groups, slot padding, pre-pass
    # triangles) next to what the algorithm needs (the level matrices, once per sweep)
    sweeps = []
    launches = stored = composite = tri = 0
    alg_sweeps = 0
    streamed_bytes = 0
    bw_modes = []
    for l, lev in enumerate(ml.levels):
        per = {}
        mode = int(lib.amgh_debug_bw_mode(dev.h, l))       # 0 level schedules, 1 / 2 / 3 wavefront of blocks (launched / chained / dataflow)
        bw_modes.append(mode)
        for bwd in (False, True):
            st = dev.gs_sweep_stats(l, bwd)
            per["bwd" if bwd else "fwd"] = st
            launches += 2 * st["launches"]                      # pre- and post-smoother run both directions
            stored += 2 * max(st["slot_entries"], st["entries"])
            composite += 2 * st["entries"]
            tri += 2 * st["tri_entries"]
            if mode > 0:
                # a block level streams its packed records: per row 16-byte chunks [values, diagonal, reciprocal | uint16
                # columns, publish word] (csrc/hip/gs_blocks.hpp Packed::chunks), not 12 bytes per entry — on the dictionary
                # layout (gs_flow.hpp FlowDict: the value chunks once per block, in LDS) the column chunks only
                maxk = st["slot_entries"] // max(1, st["rows"])
                chunks = ((maxk + 2 + 1) // 2 + (maxk + 7) // 8) | 1
                if int(lib.amgh_debug_bw_dict(dev.h, l)) == 1:
                    chunks = (maxk + 7) // 8
                streamed_bytes += 2 * st["rows"] * (16 * chunks + 24)
            else:
                streamed_bytes += 2 * (12 * (max(st["slot_entries"], st["entries"]) + st["tri_entries"]) + 24 * st["rows"])
        alg_sweeps += 4 * (lev.A.nnz * 12 + (lev.A.m + 1) * 4 + 24 * lev.A.m)
        sweeps.append(per)
    smooth_ms = None
    cycle_breakdown = None
    if not args.light:
        # the smoother's share of the cycle AS IT RUNS (hipEvent labels of the cycle = the reference's TimerOutputs labels);
        # amgh_bench_op would time the stand-alone sweep, which also gathers / scatters x between numberings
        dev.profile(True)
        for _ in range(3):
            step()
        sync()
        prof = dev.profile_read()
        dev.profile(False)
        smooth_ms = float(sum(prof[k].sum() for k in prof if k in ("Presmoother", "Postsmoother"))) / 3.0
        cycle_breakdown = {k: round(float(prof[k].sum()) / 3.0, 4) for k in prof}

    vb = vcycle_bytes(ml, 4)
    traffic, traffic_how = pmc_traffic(N, enabled=not args.no_pmc and not args.light)
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
                     "traffic": traffic, "traffic_source": traffic_how, "algorithmic_bytes": alg, "avg_launch_ms": spmv_ms,
                     # self-normalisation: the pure-read / copy rate of THIS box measured in this run, and the
                     # kernel against it — the datasheet fraction moves with the box, this one should not
                     "read_ceiling_GBs": read_GBs, "copy_ceiling_GBs": copy_GBs,
                     "frac_of_read_ceiling": None if not read_GBs else achieved / read_GBs,
                     "gpu_state_before_timed_region": clocks_before, "gpu_state_after_timed_region": clocks_after,
                     "fused_residual_ms": resid_ms,
                     "fused_residual_GBs": (alg + 8 * n) / (resid_ms * 1e-3) / 1e9},
        # (EFFECTIVE figures: the 12-bytes-per-entry count of BASELINE.md section 4 over the measured time — with the dictionary
        # records and the value-coded columns the kernels MOVE fewer bytes than this count, sweep_roofline.bytes_streamed_per_cycle)
        "vcycle": {"algorithmic_bytes": vb, "achieved_GBs": vb / (ms_per_step * 1e-3) / 1e9,
                   "frac_of_hbm_peak": vb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "frac_of_hbm_peak_is": "effective (algorithmic 12-byte count / time), not bytes moved / time",
                   "fine_symmetric_gs_ms": sweep_ms},
        # second roofline entry: the kernels that dominate the CYCLE (Gauss-Seidel sweeps: gs_slot / gs_bigslot /
        # chain / block launches + pre-pass), latency-bound — one launch per merged group of dependency levels
        "sweep_roofline": {
            "bound": "hbm", "kernel": "gs_bw_relay_kernel (levels %s: wavefront of blocks as a dataflow, one launch per sweep, a block's walk relayed between 3 waves; "
                                      "dictionary layout of the records on levels %s) + gs_slot_kernel / gs_bigslot_kernel "
                                      "(merged dependency-level groups), all levels, pre + post smoother, both directions"
                                      % ([l for l, m in enumerate(bw_modes) if m == 3], [l for l in range(len(bw_modes)) if int(lib.amgh_debug_bw_dict(dev.h, l)) == 1]),
            "block_wavefront_mode_by_level": bw_modes,
            "dictionary_layout_by_level": [int(lib.amgh_debug_bw_dict(dev.h, l)) for l in range(len(bw_modes))],
            # (bit 0 / 1 / 2: the level-ordered cycle streams A / R / P of the level as value-coded columns, 4 bytes per entry)
            "value_coded_operators_by_level": [int(lib.amgh_debug_coded_ops(dev.h, l)) for l in range(len(bw_modes))],
            "launches_per_cycle": launches, "entries_streamed_per_cycle": stored, "composite_entries_per_cycle": composite,
            "prepass_entries_per_cycle": tri,
            # (as laid out: block levels their packed records — 80 / 208 bytes per 7- / 19-point row, 16 / 48 on the dictionary layout — merged levels 12 bytes per
            # padded composite entry + pre-pass triangle; both sides of the ratio carry b, x in and x out, 24 bytes per row.
            # Measured HBM traffic of the fine-level sweep: profiles/r04_pmc_flow.log)
            "bytes_streamed_per_cycle": streamed_bytes, "algorithmic_bytes_per_cycle": alg_sweeps,
            "inflation": streamed_bytes / alg_sweeps,
            "smoother_ms_per_cycle": smooth_ms, "cycle_ms_by_label": cycle_breakdown,
            "avg_launch_us": None if smooth_ms is None else 1e3 * smooth_ms / max(1, launches),
            "achieved": None if smooth_ms is None else alg_sweeps / (smooth_ms * 1e-3) / 1e9,
            "streamed_GBs": None if smooth_ms is None else streamed_bytes / (smooth_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None if smooth_ms is None else alg_sweeps / (smooth_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "per_level": [{"fwd": p["fwd"], "bwd": p["bwd"]} for p in sweeps[:6]]},
        "setup_s": t_setup, "setup_s_parts": {"poisson_generation_s": t_problem, "ruge_stuben_s": t_setup - t_problem,
                                               "note": "setup_s = the seconds before the first cycle, the generation of the operator included; "
                                                       "ruge_stuben_s is what the reference's ruge_stuben(A) call corresponds to"},
        "upload_s": t_upload, "setup_overlapped_with_upload": bool(overlap),
        **({"overlap_error": overlap_error} if overlap_error else {}),
        "hbm_bytes": dev.device_bytes(), "hbm_bytes_by_category": dev.device_bytes_detail(),
        "collapsed_tail": tail_info(dev),
    }
    if not args.light:
        # secondary (never `value`): time to SOLUTION on the same hierarchy — the device-resident preconditioned CG of amgh_pcg_d
        # (cg(A, b; Pl = aspreconditioner(ml)) as the reference's tests run it, cycle_tests.jl:25), right-hand side and solution in
        # HBM, to the default reltol sqrt(eps): iterations, milliseconds, the true relative residual computed on the host.
        try:
            import ctypes as C
            xd = AMG.DeviceBuffer(n, 0, np.zeros(n))
            hist = np.zeros(201)
            its = C.c_int(0)
            rtol = float(np.sqrt(np.finfo(np.float64).eps))
            lib.amgh_pcg_d(dev.h, bd.ptr, xd.ptr, 0, 1, 200, 0.0, rtol, hist.ctypes.data, C.byref(its))   # warm-up (first-use buffers)
            sync()
            xd.upload(np.zeros(n))
            t0 = time.perf_counter()
            rc = lib.amgh_pcg_d(dev.h, bd.ptr, xd.ptr, 0, 1, 200, 0.0, rtol, hist.ctypes.data, C.byref(its))
            sync()
            t_pcg = 1e3 * (time.perf_counter() - t0)
            if rc != 0:
                raise RuntimeError(lib.amgh_strerror(rc).decode())
            xs = xd.download()
            As = A.to_scipy()
            true_res = float(np.linalg.norm(b - As @ xs) / np.linalg.norm(b))
            out["pcg_to_solution"] = {"reltol": rtol, "iterations": int(its.value), "ms": t_pcg, "ms_per_iteration": t_pcg / max(1, its.value),
                                      "true_rel_residual": true_res, "setup_plus_solve_s": t_setup + 1e-3 * t_pcg,
                                      "what": "amgh_pcg_d: V-cycle-preconditioned CG, b and x resident in HBM (IterativeSolvers.cg semantics)"}
            del xd, As, xs
        except Exception as ex:  # noqa: BLE001
            out["pcg_to_solution"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    if not args.light and not args.no_block_rhs:
        # secondary (never `value`): the same cycle on a block of 8 right-hand sides (workspace block size 8,
        # multilevel.jl:28-59) — one launch per sweep carries all columns; first column checked bitwise against the timed
        # single-column result above (same b in column 0).  A failure here never costs the primary line.
        try:
            bs = 8
            t0 = time.perf_counter()
            devb = ml.device(0, bs)
            t_layout = time.perf_counter() - t0
            Bh = np.stack([b] + [uniform(n, 100 + c) for c in range(1, bs)], axis=1)
            Bd = AMG.DeviceBuffer(n * bs, 0, np.asfortranarray(Bh).ravel(order="F"))
            Zd = AMG.DeviceBuffer(n * bs, 0)
            for _ in range(2):
                if lib.amgh_precond_apply_d(devb.h, Bd.ptr, Zd.ptr, 0) != 0:
                    raise RuntimeError("precond_apply on the block failed")
            if lib.amgh_dev_sync(0) != 0:
                raise RuntimeError("amgh_dev_sync after the warm-up cycles on the block failed")
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                if lib.amgh_precond_apply_d(devb.h, Bd.ptr, Zd.ptr, 0) != 0:
                    raise RuntimeError("precond_apply on the block failed")
            if lib.amgh_dev_sync(0) != 0:
                raise RuntimeError("amgh_dev_sync after the timed cycles on the block failed")
            ms8 = 1e3 * (time.perf_counter() - t0) / reps
            z8 = Zd.download()[:n]
            out["block_of_right_hand_sides"] = {
                "bs": bs, "ms_per_cycle": ms8, "unknowns_per_s": n * bs / (ms8 * 1e-3), "speedup_vs_columns_one_by_one": bs * ms_per_step / ms8,
                "layout_s": t_layout, "hbm_bytes": devb.device_bytes(),
                "first_column_rel_diff_vs_single_column_cycle": float(np.linalg.norm(z8 - z_timed) / np.linalg.norm(z_timed)),
                "block_wavefront_mode_by_level": [int(lib.amgh_debug_bw_mode(devb.h, l)) for l in range(len(ml.levels))]}
            del Bd, Zd, devb
            ml._dev.pop((0, bs), None) if hasattr(ml, "_dev") else None
        except Exception as ex:  # noqa: BLE001
            out["block_of_right_hand_sides"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    if not args.light and not args.no_uncompressed:
        # secondary (never `value`): the cycle a general operator gets — no dictionary records, no value-coded columns
        try:
            unc, z_unc = uncompressed_cycle_ms(ml, lib, bd, zd, n)
            unc["rel_diff_vs_compressed_cycle"] = float(np.linalg.norm(z_unc - z_timed) / np.linalg.norm(z_timed))
            out["vcycle_uncompressed_ms"] = unc["ms_per_cycle"]
            out["vcycle_uncompressed"] = unc
        except Exception as ex:  # noqa: BLE001
            out["vcycle_uncompressed_ms"] = None
            out["vcycle_uncompressed"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    if not args.light and not args.no_secondary_configs:
        try:
            out["secondary_configs"] = secondary_configs(check=not args.no_cpu_baseline)
        except Exception as ex:  # noqa: BLE001
            out["secondary_configs"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    if not args.no_cpu_baseline:
        out["cpu_baseline"], z_oracle = cpu_baseline(ml, b, args.cpu_budget)
        # parity of the TIMED result (checked outside the timed region): the last V-cycle's output against the oracle
        err = float(np.linalg.norm(z_timed - z_oracle) / np.linalg.norm(z_oracle))
        out["parity"] = {"rel_err_vs_oracle": err, "tolerance": 1e-10, "what": "||z - z_oracle|| / ||z_oracle|| of the last timed V-cycle"}
        if not err <= 1e-10:
            raise SystemExit(f"bench.py: the timed V-cycle differs from the oracle: rel.err {err:.3e} > 1e-10")
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
