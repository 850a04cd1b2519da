#!/usr/bin/env python3
"""bench_dist.py — N > 1 leg of bench.py: the 3-D Poisson V-cycle row-sharded over N MI355X, one process per GPU,
through libamghip's `amgh_dist_*` C ABI (neighbour exchange of halo entries before every operator, coarse levels
collapsed onto rank 0).  STRONG scaling: the 256^3 problem is fixed, each rank owns 1/N of the rows of every sharded
level.

Launched by:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                  --master-port P bench.py --gpus N --steps K --warmup W [--transport rccl|ipc] [--smoother gs|jacobi]
         or:  python bench.py --gpus N ...   (no launcher: bench.py spawns its N ranks itself, bench.self_launch)
The rendezvous is bounded (AMGH_RENDEZVOUS_TIMEOUT_S, default 600): a rank whose peers never arrive fails with a message that
names the launcher instead of waiting for ever.  `--host-exec` is the launcher's self-test on a box without GPUs.

One run measures up to three configurations on the SAME hierarchy (A, P, R of every level are shared):
  primary    --smoother (default gs: ruge_stuben defaults, BASELINE.json C4) over --transport (default rccl: RCCL
             send/recv called by the library; falls back to ipc if RCCL cannot be initialised) -> the JSON line's value;
  "jacobi"   the same hierarchy smoothed by Jacobi(2/3): the curve that CAN scale (lexicographic Gauss-Seidel keeps
             256 + 256 + 256/N - 2 dependency levels per shard);
  "ipc"      the primary smoother over the IPC transport (hipIpc peer-mapped send buffers + stream-written flags).
Every configuration is SELF-CHECKING on rank 0 (outside the timed region): the assembled result of the last timed
cycle against the oracle (Jacobi: exact, <= 1e-10) or against the host emulation of the frozen-halo sweeps
(Gauss-Seidel, <= 1e-10; tests/sharded_emulation.py); a failed check fails the run.  Rank 0 also times the CPU
restatement of the reference's cycle (`cpu_baseline`, as bench.py does).

torch is used for ONE thing: the rendezvous the launcher already provides (a gloo group on CPU: the 128-byte RCCL id /
the shared-memory name, small broadcasts).  It never touches the GPU; the timed barriers and the max-over-ranks are the
library's own.  The host hierarchy is built ONCE per node (rank 0); the level matrices of the sharded levels go to
/dev/shm as .npy files and every other rank maps them and reads only its own rows.
"""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

PARITY_TOL = 1e-10


def node_levels(rank, world, bcast, build, shard_min_rows=200_000):
    """The host hierarchy ONCE per node: rank 0 runs `build()` (-> MultiLevel), decides how many levels are sharded,
    and — for world > 1 — exports the sharded levels' matrices as .npy files; every other rank maps them and will read
    only its own rows.  `bcast(obj)` broadcasts a small Python object from rank 0.  Returns (levels, info, tail, dir):
    `tail` (the collapsed levels as a MultiLevel) on rank 0 only; info["ml"] is the whole hierarchy on rank 0."""
    import amg_amd as AMG
    from amg_amd import sharded as SH
    shm = None
    tail = None
    levels = None
    ml = None
    t_phase = {}
    if rank == 0:
        t0 = time.perf_counter()
        ml = build()
        t_phase["hierarchy_s"] = time.perf_counter() - t0
        A = ml.levels[0].A if ml.levels else ml.final_A
        sizes = [l.A.m for l in ml.levels] + [ml.final_A.m]
        lc = SH.num_sharded_levels(sizes, world, shard_min_rows)
        levels = SH.level_arrays(ml, lc)
        tail = AMG.MultiLevel(ml.levels[lc:], ml.final_A, ml.coarse_solver, ml.presmoother, ml.postsmoother,
                              ml.symmetry, method=ml.method)
        info = dict(n_tail=sizes[lc], nnz=A.nnz, nlev=len(ml), lc=lc, sizes=sizes, level_nnz=[l.A.nnz for l in ml.levels], shard_min_rows=shard_min_rows)
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
            t0 = time.perf_counter()
            SH.export_levels(levels, shm)
            t_phase["export_s"] = time.perf_counter() - t0
            t_phase["export_bytes"] = int(need)
        info["shm"] = shm
        info["phases"] = t_phase
    else:
        info = None
    info = bcast(info)
    if rank != 0:
        t0 = time.perf_counter()
        levels = SH.load_levels(info["shm"])
        info = dict(info, load_s=time.perf_counter() - t0)
    info = dict(info, ml=ml)
    return levels, info, tail, shm


def with_smoother(ml, sm):
    """The same hierarchy (A, P, R shared) smoothed by `sm` on every level."""
    import amg_amd as AMG
    lv = [AMG.Level(l.A, l.P, l.R, sm, sm) for l in ml.levels]
    return AMG.MultiLevel(lv, ml.final_A, ml.coarse_solver, sm, sm, ml.symmetry, method=ml.method)


class Run:
    """One measured configuration: a sharded handle, K timed cycles, the assembled result on rank 0."""

    def __init__(self, ctx, label, levels, tail, transport, gs_mode="exact", n_tail=None):
        from amg_amd import sharded as SH
        self.ctx, self.label = ctx, label
        t0 = time.perf_counter()
        self.sh = SH.ShardedHierarchy(levels, ctx["info"]["n_tail"] if n_tail is None else n_tail, tail, ctx["rank"], ctx["world"], ctx["device"],
                                      transport, gs_mode=gs_mode, host_tail=ctx.get("host_tail"))
        self.shard_s = time.perf_counter() - t0
        self.transport = transport[0]
        self.pipelined = self.sh.gs_pipelined()

    def measure(self, b, steps, warmup):
        sh = self.sh
        sh.barrier()
        sh.set_rhs(b[sh.r0:sh.r1])
        for _ in range(warmup):
            sh.precond_apply_d(0)
        sh.barrier()
        sh.stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            sh.precond_apply_d(0)
        sh.barrier()
        elapsed = float(sh.allreduce([time.perf_counter() - t0], "max")[0])
        st = sh.stats(reset=True)
        z_loc = sh._down(sh._x)                         # the LAST timed cycle's result, before anything else runs
        out = {"ms_per_step": 1e3 * elapsed / steps, "elapsed": elapsed, "transport": self.transport, "shard_s": self.shard_s,
               "halo_exchanges_per_cycle": st["halo_exchanges"] / max(1, steps),
               "halo_bytes_sent_per_cycle_max_rank": sh.allreduce([st["halo_bytes_sent"] / max(1, steps)], "max")[0],
               "halo_bytes_sent_per_cycle_all_ranks": sh.allreduce([st["halo_bytes_sent"] / max(1, steps)], "sum")[0]}
        return out, z_loc

    def spmv_ms(self, reps=20):
        """fine-level sharded SpMV (neighbour exchange + local rows) on the level's resident x, timed like the cycle"""
        import amg_amd as AMG
        sh = self.sh
        if sh.lc == 0 or sh.host_exec:
            return None
        y = AMG.DeviceBuffer(max(sh.nloc, 1), self.ctx["device"])
        for _ in range(3):
            sh.lib.amgh_dist_spmv_d(sh.h, 0, None, y.ptr)
        sh.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            sh.lib.amgh_dist_spmv_d(sh.h, 0, None, y.ptr)
        sh.barrier()
        return 1e3 * float(sh.allreduce([time.perf_counter() - t0], "max")[0]) / reps

    def tail_ms(self, reps=5):
        """The part of the cycle that does NOT shard: one visit of the collapsed levels (everything below the sharded ones, coarse
        solve included) on the rank that owns them, timed alone on the handle's stream — `serial_ms_on_rank0` of the line.  Not a
        collective: the owner measures while the other ranks wait at the next barrier."""
        import amg_amd as AMG
        sh = self.sh
        if sh.tail is None or sh.host_exec:
            return None
        nt = sh.tail.ml.levels[0].A.m if sh.tail.ml.levels else sh.tail.ml.final_A.m
        bt = AMG.DeviceBuffer(nt, self.ctx["device"], np.linspace(0.1, 1.0, nt))
        zt = AMG.DeviceBuffer(nt, self.ctx["device"])
        lib = sh.lib
        for _ in range(2):
            if lib.amgh_precond_apply_d(sh.tail.h, bt.ptr, zt.ptr, 0) != 0:
                return None
        sh.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            lib.amgh_precond_apply_d(sh.tail.h, bt.ptr, zt.ptr, 0)
        sh.sync()
        return 1e3 * (time.perf_counter() - t0) / reps

    def close(self):
        self.sh.barrier()
        self.sh.close()


def assemble_on_rank0(ctx, tag, z_loc):
    """Every rank's rows of a result vector -> the whole vector on rank 0 (through the node's shared directory)."""
    d = ctx["gdir"]
    np.save(os.path.join(d, f"z_{tag}_{ctx['rank']}.npy"), z_loc)
    ctx["barrier"]()
    z = None
    if ctx["rank"] == 0:
        z = np.concatenate([np.load(os.path.join(d, f"z_{tag}_{r}.npy")) for r in range(ctx["world"])])
    ctx["barrier"]()
    return z


def check_parity(ctx, ml, b, z, kind, gs_mode="exact"):
    """rank 0: the assembled result of one cycle from x = 0 against the checker (never part of the timed region)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    if kind == "jacobi" or gs_mode in ("exact", "exact-turns"):
        from oracle import oracle as O
        want = O.OracleHierarchy(ml).precond(b)
        what = ("||z - z_oracle|| / ||z_oracle||, oracle V-cycle (" + ("Jacobi smoothers are exact across shards" if kind == "jacobi" else
                "Gauss-Seidel in exact lexicographic order over the whole level: the ranks sweep in turn") + ")")
    else:
        from sharded_emulation import emulate_sharded_cycles
        want = emulate_sharded_cycles(ml, b, ctx["world"], ctx["info"]["lc"], 1)[0]
        what = ("||z - z_emul|| / ||z_emul||, host emulation of the sharded cycle with the oracle's loops "
                "(Gauss-Seidel exact inside a shard, halo frozen per directional sweep)")
    err = float(np.linalg.norm(z - want) / np.linalg.norm(want))
    return {"rel_err": err, "tolerance": PARITY_TOL, "what": what, "ok": bool(err <= PARITY_TOL)}


def amdahl(primary, info, world, N):
    """What a reader of a flat curve needs: the levels below the sharded ones run on ONE rank (exact lexicographic Gauss-Seidel on
    merged dependency-level groups does not shard), so the cycle cannot drop below that serial part whatever N is."""
    serial = primary.get("tail_ms")
    out = {"sharded_levels": primary.get("sharded_levels"), "collapsed_levels_rows": info["sizes"][primary.get("sharded_levels") or 0:],
           "serial_ms_on_rank0": serial}
    if serial is None:
        return out
    t1, src = None, None
    try:   # the single-GPU cycle of the same problem as this repository last measured it (a run at N > 1 does not hold the unsharded layout)
        here = os.path.dirname(os.path.abspath(__file__))
        for name in ("r06_bench_256.json", "r05_bench_256.json"):
            f = os.path.join(here, "profiles", name)
            if N == 256 and os.path.exists(f):
                t1, src = float(json.load(open(f))["ms_per_step"]), "profiles/" + name
                break
    except Exception:  # noqa: BLE001
        pass
    out["amdahl"] = {
        "what": "levels >= sharded_levels (merged-group / dense Gauss-Seidel sweeps, coarse solve) run on rank 0 alone: ms_per_step >= "
                "serial_ms_on_rank0 for every N; with the sharded part scaling perfectly the cycle is serial + (T1 - serial) / N",
        "serial_ms_on_rank0": serial, "sharded_part_ms_this_run": primary["ms_per_step"] - serial,
        "single_gpu_ms": t1, "single_gpu_ms_source": src,
        "best_possible_ms_at_this_N": None if t1 is None else serial + max(0.0, t1 - serial) / world,
        "max_speedup_at_this_N": None if t1 is None else t1 / (serial + max(0.0, t1 - serial) / world),
        "max_speedup_any_N": None if t1 is None else t1 / serial}
    return out


def main_distributed(args):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    os.environ.setdefault("AMGH_IPC_TIMEOUT_S", "120")
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
        # (gloo announces its connections on stdout: rank 0's stdout is the ONE JSON line — the chatter goes to stderr)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        rdv_s = float(os.environ.get("AMGH_RENDEZVOUS_TIMEOUT_S", "600"))
        try:
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=rdv_s))
        except Exception as ex:  # noqa: BLE001  (a rank that never arrives: say who was expected to start it, then fail — never wait for ever)
            raise SystemExit(f"bench_dist.py: rank {rank} of {world} found no peers at {os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']} within "
                             f"{rdv_s:.0f} s ({type(ex).__name__}: {str(ex)[:200]}). Launcher: {os.environ.get('AMGH_BENCH_LAUNCHER', 'external (RANK / WORLD_SIZE were set by the caller)')}; "
                             f"start the ranks with `python bench.py --gpus {world}` (bench.py spawns them) or `python -m torch.distributed.run --nnodes=1 "
                             f"--nproc-per-node {world} --master-addr 127.0.0.1 bench.py --gpus {world}`")
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    def bcast(obj):
        if world == 1:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def all_ok(flag):
        if world == 1:
            return bool(flag)
        box = [None] * world
        dist.all_gather_object(box, bool(flag))
        return all(box)

    def host_barrier():
        if world > 1:
            dist.barrier()

    import __graft_entry__ as g
    if rank == 0:
        g.build(only_missing=True)
    host_barrier()
    import amg_amd as AMG
    from amg_amd import sharded as SH
    from bench import cpu_baseline, spmv_bytes, uniform

    host_exec = bool(getattr(args, "host_exec", False))   # launcher self-test without GPUs: amgh_dist_* executed in host memory
    one_gpu = os.environ.get("AMG_DIST_ONE_GPU") == "1"   # functional check on a single-GPU box: every rank on device 0
    ngpu = 0 if host_exec else int(AMG.hip_lib().amgh_device_count())
    if host_exec:
        one_gpu = False
    elif not one_gpu and ngpu < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        # more ranks than devices (every rank sees the same count): ranks that set a missing device would fail while the
        # others wait in a collective for ever — run the functional mode instead and say so in the line
        one_gpu = True
        if rank == 0:
            print(f"bench_dist: {world} ranks but {ngpu} GPU(s) visible: all ranks share device 0 (functional run, not a scaling "
                  f"measurement)", file=sys.stderr, flush=True)
    if one_gpu:
        local_rank = 0
    transport = getattr(args, "transport", None) or os.environ.get("AMGH_DIST_TRANSPORT", "rccl")
    smoother = getattr(args, "smoother", None) or "gs"
    secondary = not getattr(args, "no_secondary", False)
    if one_gpu and world > 1 and transport == "rccl":
        transport = "ipc"                                  # RCCL refuses two ranks on one device
    N = args.size
    n = N ** 3
    # ---- host hierarchy: once per node --------------------------------------------------------------------------
    t0 = time.perf_counter()
    if host_exec:
        # three levels, both sparse ones sharded whatever their size: what is collapsed onto rank 0 is the coarse solve alone —
        # the host mirror's own Pinv (coarse_solver.jl:9-16), so that nothing of this mode's cycle comes from the checker
        transport, local_rank, secondary = "ipc", -1, False
        levels, info, tail, shm = node_levels(rank, world, bcast,
                                              lambda: AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="host", max_levels=3, coarse_solver=AMG.Pinv),
                                              shard_min_rows=1)
    else:
        levels, info, tail, shm = node_levels(rank, world, bcast,
                                              lambda: AMG.ruge_stuben(AMG.poisson((N, N, N)), setup=getattr(args, "setup", "gpu")))
    t_setup = time.perf_counter() - t0
    ml = info["ml"]                                        # rank 0 only
    gdir = bcast(tempfile.mkdtemp(prefix="amgh_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) if rank == 0 else None)
    ctx = dict(rank=rank, world=world, device=local_rank, info=info, gdir=gdir, barrier=host_barrier)
    if host_exec:
        if rank == 0:
            if tail.levels:
                raise SystemExit("bench_dist.py --host-exec: the hierarchy left sparse levels unsharded (grid too small for this many ranks)")
            pinv = tail.coarse_solver.dense_operator()
            ctx["host_tail"] = lambda bb: pinv @ bb
        else:
            ctx["host_tail"] = True
    jac = AMG.Jacobi(2.0 / 3.0)
    jac_tuple = SH._smoother_tuple(jac)

    def variant(kind):
        """(level arrays, collapsed levels, host hierarchy for the checker) of smoother `kind`"""
        if kind == "gs":
            return levels, tail, ml
        lv = [dict(d, pre=jac_tuple, post=jac_tuple) for d in levels]
        return lv, (with_smoother(tail, jac) if tail is not None else None), (with_smoother(ml, jac) if ml is not None else None)

    def transport_spec(kind):
        if kind == "rccl":
            os.environ.pop("AMGH_IPC_STAGED", None)
            return ("rccl", bcast(SH.rccl_unique_id() if rank == 0 else None))
        # "ipc-staged": the IPC transport without any peer mapping — exchanges staged through host memory and the shared-memory
        # rendezvous (slow, synchronous): the last resort when neither RCCL nor hipIpc mappings work between the node's devices
        if kind == "ipc-staged":
            os.environ["AMGH_IPC_STAGED"] = "1"
        else:
            os.environ.pop("AMGH_IPC_STAGED", None)
        return ("ipc", bcast("/amgh_b_%d_%s" % (os.getpid(), os.urandom(4).hex()) if rank == 0 else None))

    b = uniform(n, 0)
    notes = {}

    def preflight(tkind, pipe=False):
        """A 2-cycle exchange of a small hierarchy with SHARDED levels (shard_min_rows 500) over transport `tkind`, checked against
        the oracle on rank 0, before the timed problem meets that transport for the first time (bounded: the IPC transport's own
        timeout is 30 s here; RCCL has none of its own — a rank that cannot initialise it raises, a rank that hangs in it is the
        launcher's to kill).  pipe = False: 32^3, Jacobi smoothers (the halo exchanges).  pipe = True: 40^3, the default
        Gauss-Seidel smoothers with block layouts forced on the small shards, exact order as ONE sweep pipelined across the ranks:
        the neighbours' mailbox arrays mapped across processes / devices and polled from inside the sweeps — on this hardware,
        before the timed problem depends on it."""
        t0 = time.perf_counter()
        rec = {"transport": tkind, "ok": False, "what": "gs_pipelined" if pipe else "jacobi_exchange"}
        lib = AMG.hip_lib()
        if pipe:
            lib.amgh_debug_set_tunable(b"gs_bw", 2); lib.amgh_debug_set_tunable(b"gs_bw_rows", 64)
        old = os.environ.get("AMGH_IPC_TIMEOUT_S")
        os.environ["AMGH_IPC_TIMEOUT_S"] = "30"
        run = None
        try:
            ns = 40 if pipe else 32
            small = None
            if rank == 0:
                small = AMG.ruge_stuben(AMG.poisson((ns, ns, ns))) if pipe else AMG.ruge_stuben(AMG.poisson((ns, ns, ns)), presmoother=jac, postsmoother=jac)
            lv_s, info_s, tail_s, shm_s = node_levels(rank, world, bcast, lambda: small, shard_min_rows=4000 if pipe else 500)
            ctx_s = dict(ctx, info=info_s)
            err = None
            try:
                run = Run(ctx_s, "preflight", lv_s, tail_s, transport_spec(tkind))
            except AMG.AMGError as e:
                err = str(e)[:200]
            if not all_ok(run is not None):
                rec["error"] = err or "another rank failed to create the handle"
            else:
                nb = ns ** 3
                bs = uniform(nb, 7)
                if pipe:
                    rec["pipelined_by_level"] = run.pipelined
                res, z_loc = run.measure(bs, 2, 0)
                z = assemble_on_rank0(ctx_s, "preflight_" + tkind, z_loc)
                good = True
                if rank == 0:
                    from oracle import oracle as O
                    want = O.OracleHierarchy(info_s["ml"]).precond(bs)
                    rel = float(np.linalg.norm(z - want) / np.linalg.norm(want))
                    rec["rel_err_vs_oracle"] = rel
                    good = rel <= PARITY_TOL
                rec["ok"] = bool(bcast(good if rank == 0 else None))
                if not rec["ok"]:
                    rec["error"] = "the exchanged cycle differs from the oracle"
                if pipe and rec["ok"] and not (run.pipelined and run.pipelined[0]):
                    rec["ok"], rec["error"] = False, "the small shards did not get a pipelined sweep"
            if rank == 0 and shm_s:
                shutil.rmtree(shm_s, ignore_errors=True)
        except Exception as ex:  # noqa: BLE001
            rec["error"] = f"{type(ex).__name__}: {str(ex)[:200]}"
        finally:
            if run is not None:
                try:
                    run.sh.close()
                except Exception:  # noqa: BLE001
                    pass
            if old is None:
                os.environ.pop("AMGH_IPC_TIMEOUT_S", None)
            else:
                os.environ["AMGH_IPC_TIMEOUT_S"] = old
            if pipe:
                lib.amgh_debug_set_tunable(b"gs_bw", 1); lib.amgh_debug_set_tunable(b"gs_bw_rows", 512)
        rec["ok"] = all_ok(rec["ok"])
        rec["seconds"] = round(time.perf_counter() - t0, 2)
        return rec

    preflights = []
    if world > 1 and not getattr(args, "no_preflight", False) and not host_exec:
        order = [transport] + [t for t in ("rccl", "ipc", "ipc-staged") if t != transport and not (one_gpu and t == "rccl")]
        chosen = None
        for tk in order:
            if tk == "ipc-staged" and chosen is not None:
                continue                                   # (the last resort is only tried when nothing else came back right)
            rec = preflight(tk)
            preflights.append(rec)
            if rec["ok"] and chosen is None:
                chosen = tk
        if chosen is None:
            raise SystemExit(f"bench_dist.py: no transport passed the preflight exchange: {preflights}")
        if chosen == "ipc-staged":
            notes["ipc_staged"] = "exchanges staged through host memory (no peer mapping worked between the devices): a functional number, not the transport's speed"
        if smoother == "gs" and chosen != "ipc-staged":
            # the pipelined exact sweep maps memory across processes / devices and polls it from inside kernels: tried on a small
            # problem first; if it does not come back right on this hardware, the timed problem sweeps with the ranks in turn
            rec = preflight(chosen, pipe=True)
            preflights.append(rec)
            if not rec["ok"]:
                os.environ["AMGH_DIST_PIPE"] = "0"
                notes["gs_pipeline_disabled"] = rec.get("error", "preflight failed")
        if chosen != transport:
            notes["transport_requested"] = transport
            notes["transport_fallback_reason"] = next((r.get("error", "preflight failed") for r in preflights if r["transport"] == transport), "preflight failed")
            transport = chosen

    # Gauss-Seidel in exact lexicographic order is ONE dependency chain through the grid.  Where every rank holds the block
    # (dataflow) layout of its shard, the library sweeps the level as one pipeline across the ranks (amgh_dist_set_gs_mode 1):
    # all ranks launch at once, blocks poll the rows they read of the neighbouring rank in that rank's mailboxes — the chain
    # is paid once, the bandwidth-bound middle of the wavefront is shared by the ranks.  Levels without that layout (rows too
    # long, or too few: the merged-group schedules) would be swept with the ranks IN TURN — N turns and N exchanges per sweep
    # for nothing — so the exact curve shards the levels that pipeline (the library's own size rule for block layouts, applied
    # to the whole level) and sweeps the ones below at single-GPU speed on rank 0.  Sharding every level of >= 200 000 rows in
    # exact order is the secondary "gs_exact_all_levels".
    lc_all = info["lc"]
    def pipelines(l):
        rows, nnz = info["sizes"][l], info["level_nnz"][l]
        return rows >= 3_000_000 or (rows >= 1_500_000 and nnz <= 7 * rows)
    lc_exact = 0
    while lc_exact < lc_all and (pipelines(lc_exact) or host_exec):
        lc_exact += 1
    lc_exact = max(min(1, lc_all), lc_exact)
    tail_exact = None
    if rank == 0 and ml is not None and lc_exact < lc_all:
        tail_exact = AMG.MultiLevel(ml.levels[lc_exact:], ml.final_A, ml.coarse_solver, ml.presmoother, ml.postsmoother, ml.symmetry, method=ml.method)

    def run_config(label, kind, tkind, want_spmv=False, gs_mode="exact", all_levels=False):
        lv, tl, ml_k = variant(kind)
        n_tail = None
        if kind == "gs" and gs_mode in ("exact", "exact-turns") and not all_levels and lc_exact < lc_all:
            lv, tl, n_tail = lv[:lc_exact], tail_exact, info["sizes"][lc_exact]
        run, err = None, None
        try:
            run = Run(ctx, label, lv, tl, transport_spec(tkind), gs_mode, n_tail=n_tail)
        except AMG.AMGError as e:
            err = str(e)
        if not all_ok(run is not None):
            if run is not None:
                run.sh.close()
            return None, (err or "another rank failed to create the handle")
        res, z_loc = run.measure(b, args.steps, args.warmup)
        if want_spmv:
            res["spmv_ms"] = run.spmv_ms()
            res["tail_ms"] = run.tail_ms()               # (the owner of the collapsed levels only; the others wait below)
        z = assemble_on_rank0(ctx, label, z_loc)
        if rank == 0:
            res["parity"] = check_parity(ctx, ml_k, b, z, kind, gs_mode)
        res["gs_mode"] = gs_mode if kind == "gs" else None
        res["sharded_levels"] = len(lv)
        res["gs_pipelined_by_level"] = run.pipelined if (kind == "gs" and gs_mode == "exact") else None
        run.close()
        res["value"] = n * args.steps / res["elapsed"]
        ok = bcast(res["parity"]["ok"] if rank == 0 else None)
        return res, (None if ok else "parity check failed")

    # ---- primary configuration ----------------------------------------------------------------------------------
    primary, err = run_config("primary", smoother, transport, want_spmv=True)
    if primary is None and transport == "rccl":
        notes["rccl_error"] = err
        transport = "ipc"
        primary, err = run_config("primary", smoother, transport, want_spmv=True)
    if primary is None:
        raise SystemExit(f"bench_dist.py: no transport could be initialised: {err}")
    if err:
        if rank == 0:
            print(json.dumps({"error": err, "parity": primary.get("parity")}), file=sys.stderr, flush=True)
        raise SystemExit(f"bench_dist.py: {err}: {primary.get('parity')}")
    # ---- secondary configurations (same hierarchy; a failure here never costs the primary line) ------------------
    extra = {}
    if secondary and world > 1:
        todo = []
        if smoother != "jacobi":
            todo.append(("jacobi", "jacobi", transport, "exact"))
            todo.append(("gs_hybrid", "gs", transport, "hybrid"))   # every shard sweeps at once, halo frozen per directional sweep
            todo.append(("gs_exact_turns", "gs", transport, "exact-turns"))   # the same levels with the ranks strictly in turn (what round 4 shipped)
            if lc_exact < lc_all:
                todo.append(("gs_exact_all_levels", "gs", transport, "exact-all"))   # exact order with every large level sharded
        if transport != "ipc":
            todo.append(("ipc", smoother, "ipc", "exact"))
        for label, kind, tkind, mode in todo:
            try:
                res, e2 = run_config(label, kind, tkind, gs_mode="exact" if mode == "exact-all" else mode, all_levels=mode == "exact-all")
                extra[label] = {"error": e2} if res is None else dict(res, **({"error": e2} if e2 else {}))
            except Exception as ex:  # noqa: BLE001
                extra[label] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
                break                                   # the ranks may no longer be in step: stop measuring
    cpu = None
    if rank == 0 and not getattr(args, "no_cpu_baseline", False):
        cpu, _ = cpu_baseline(ml, b, getattr(args, "cpu_budget", 20.0))
    host_barrier()

    if rank == 0:
        alg = spmv_bytes(info["nnz"], n, n)
        spmv_ms = primary.get("spmv_ms")
        piped = any(primary.get("gs_pipelined_by_level") or [])
        smooth_txt = ("ruge_stuben defaults (symmetric Gauss-Seidel pre+post in exact lexicographic order over the whole level — the "
                      "reference's iterate — " + ("as ONE sweep pipelined across the ranks: all ranks launch at once, a block polls the rows "
                      "it reads of the neighbouring rank in that rank's peer-mapped mailboxes; the levels with block layouts are "
                      "sharded on this curve (gs_pipelined_by_level), the ones below are swept on rank 0)" if piped else
                      "the ranks sweeping in turn (no level runs the pipelined sweep in this run: gs_pipelined_by_level))")
                      if smoother == "gs" else "Jacobi(2/3) pre+post (exact across shards)")
        tr_txt = {"rccl": "RCCL send/recv called by libamghip", "ipc": "hipIpc peer-mapped send buffers + stream-written "
                  "flags in shared memory (libamghip's IPC transport)"}[primary["transport"]]
        if host_exec:
            tr_txt = "host memory, shared-memory rendezvous of libamghip's IPC transport"
        strip = lambda r: {k: v for k, v in r.items() if k not in ("elapsed",)}  # noqa: E731
        out = {
            "metric": f"V-cycle unknowns/sec + fine-level SpMV GB/s (% HBM peak), 3-D Poisson {N}^3",
            "value": primary["value"], "unit": "unknowns/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": primary["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "poisson((%d,%d,%d)) 7-point, %s, one V-cycle per step (ldiv!), large levels "
                                   "1-D row-sharded, neighbour exchange of halo entries before every operator (%s, interior "
                                   "rows overlapped), coarser levels collapsed to rank 0" % (N, N, N, smooth_txt, tr_txt),
                       "unknowns": n, "nnz": info["nnz"], "levels": info["nlev"], "sharded_levels": primary.get("sharded_levels", info["lc"]),
                       "sharded_levels_of_the_secondaries": info["lc"],
                       "smoother": smoother, "transport": primary["transport"],
                       "halo_exchanges_per_cycle": primary["halo_exchanges_per_cycle"],
                       "halo_bytes_sent_per_cycle_max_rank": primary["halo_bytes_sent_per_cycle_max_rank"],
                       "halo_bytes_sent_per_cycle_all_ranks": primary["halo_bytes_sent_per_cycle_all_ranks"],
                       "all_ranks_on_one_gpu": bool(one_gpu and world > 1),
                       "host_execution": ("launcher self-test: amgh_dist_* executed in host memory over the shared-memory transport, 3-level "
                                          "hierarchy, Gauss-Seidel across the ranks in turns — functional, NOT a measurement") if host_exec else None,
                       # lexicographic Gauss-Seidel is ONE dependency chain through the whole grid: in exact order the shards of a
                       # level sweep one after the other (the value of this line: the reference's iterate, a flat curve in N); the
                       # curves that CAN scale are secondary["gs_hybrid"] (every shard at once, halo frozen per directional sweep:
                       # another convergent iteration, checked against its own emulation) and secondary["jacobi"] (exact across shards)
                       "gauss_seidel_is_critical_path_bound": smoother == "gs",
                       "gs_mode": primary.get("gs_mode"), "gs_pipelined_by_level": primary.get("gs_pipelined_by_level"),
                       "scaling_curve": "secondary.gs_hybrid / secondary.jacobi" if smoother == "gs" else "value",
                       "parallelism": f"row-shard x{world} ({primary['transport']} halos, libamghip amgh_dist_*)"},
            "roofline": None if spmv_ms is None else {
                "bound": "hbm", "kernel": "csr_stream_kernel<SPMV> on n/N local rows + neighbour halo exchange",
                "achieved": alg / (spmv_ms * 1e-3) / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                "frac": alg / (spmv_ms * 1e-3) / 1e9 / (8000.0 * world), "traffic": None,
                "traffic_source": "not measured in the N > 1 leg (bench.py at N = 1 measures it with rocprofv3 PMC passes)",
                "avg_launch_ms": spmv_ms},
            "parity": primary["parity"],
            "cpu_baseline": cpu,
            "setup_s": t_setup, "shard_s": primary["shard_s"],
            "setup_breakdown": {"hierarchy_s_rank0": info.get("phases", {}).get("hierarchy_s"),
                                "export_s_rank0": info.get("phases", {}).get("export_s"),
                                "export_bytes": info.get("phases", {}).get("export_bytes"),
                                "per_rank_plans_and_upload_s": primary["shard_s"],
                                "note": "rank 0 builds the hierarchy while the other ranks wait; every rank then lays out its own shard "
                                        "(halo plans, smoother schedules, block plans) in per_rank_plans_and_upload_s"},
            "preflight": preflights, "transport_used": primary["transport"],
            "launcher": os.environ.get("AMGH_BENCH_LAUNCHER", "external") + (" (bench.py spawned its ranks)" if os.environ.get("AMGH_BENCH_LAUNCHER") == "self"
                                                                              else " (RANK / WORLD_SIZE set by the caller, e.g. torch.distributed.run)"),
            "ranks": world, "devices_visible": ngpu,
            **amdahl(primary, info, world, N),
            **({"secondary": {k: strip(v) for k, v in extra.items()}} if extra else {}),
            **notes,
        }
    host_barrier()
    if rank == 0:
        for d in (shm, gdir):
            if d:
                shutil.rmtree(d, ignore_errors=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stderr.flush()
        print(json.dumps(out), flush=True)   # the ONE JSON line, after everything RCCL may print
