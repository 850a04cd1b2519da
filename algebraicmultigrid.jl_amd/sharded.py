"""Row-sharded hierarchy on N GPUs — thin ctypes layer over libamghip's `amgh_dist_*` C ABI.

BASELINE.json config C4: fine levels partitioned by contiguous 1-D row ranges, halo entries of every
operator's input vector exchanged by neighbour send/recv (RCCL over xGMI, one process per GPU; the IPC
transport: one process per rank, peer-mapped send buffers and stream-written flags in shared memory; or
the LOCAL transport: N ranks as threads of one process), coarse levels collapsed onto rank 0.  Nothing is
computed here: every rank slices ITS rows out of the level matrices and hands them to the library, which
builds the halo plans, runs the cycle and talks to RCCL itself.  No torch, no CPU fallback.

The hierarchy only has to be BUILT once per node: `export_levels` writes the level matrices of the
sharded levels to a directory of .npy files (e.g. under /dev/shm), every other rank maps them with
`load_levels` and reads just its own rows.
"""
import ctypes as C
import os
import threading

import numpy as np

from ._libs import AMGError, amgh_smoother_t, hip_check, hip_lib
from .device import DeviceBuffer, DeviceHierarchy, require_gpu
from .hierarchy import HermitianSymmetry, MultiLevel

CYCLE_V, CYCLE_W, CYCLE_F = 0, 1, 2
ID_BYTES = 128


def row_cuts(n, nranks):
    """Contiguous 1-D row-range partition: rank p owns [p*n//N, (p+1)*n//N)."""
    return np.array([(p * n) // nranks for p in range(nranks + 1)], dtype=np.int64)


def num_sharded_levels(sizes, nranks, shard_min_rows=200_000):
    """Levels [0, lc) are sharded, level lc and below live on rank 0."""
    lc = 0
    while lc < len(sizes) - 1 and sizes[lc] >= shard_min_rows and sizes[lc] >= 8 * nranks:
        lc += 1
    return lc


def level_arrays(ml, lc):
    """CSR arrays of the sharded levels, as the C ABI takes them (global column indices)."""
    if not isinstance(ml.symmetry, HermitianSymmetry):
        raise AMGError("the sharded path is built for symmetry=HermitianSymmetry()")
    out = []
    for l in range(lc):
        lev = ml.levels[l]
        A = lev.A
        d = dict(n=A.m, nc=lev.P.n, A=A.csr_arrays(), S=None,
                 P=(lev.R.colptr, lev.R.rowval, lev.R.nzval),   # CSR of P = CSC arrays of R
                 R=(lev.P.colptr, lev.P.rowval, lev.P.nzval),
                 pre=_smoother_tuple(lev.presmoother), post=_smoother_tuple(lev.postsmoother),
                 _keep=lev)       # the arrays are views of the matrices' native memory: the level must outlive them
        if not A.is_symmetric():
            d["S"] = (A.colptr, A.rowval, A.nzval)               # column i read as row i (smoother.jl:81-86)
        out.append(d)
    return out


def _smoother_tuple(s):
    return (int(s.kind), int(s.sweep_code), int(s.iter), float(s.omega))


def export_levels(levels, dirpath):
    """Write the arrays of `level_arrays(...)` as .npy files (rank 0 of a node, once)."""
    os.makedirs(dirpath, exist_ok=True)
    meta = []
    for l, d in enumerate(levels):
        for key in ("A", "S", "P", "R"):
            if d[key] is None:
                continue
            for name, arr in zip(("rowptr", "col", "val"), d[key]):
                np.save(os.path.join(dirpath, f"L{l}_{key}_{name}.npy"), np.ascontiguousarray(arr))
        meta.append([d["n"], d["nc"], int(d["S"] is not None), *d["pre"], *d["post"]])
    np.save(os.path.join(dirpath, "meta.npy"), np.array(meta, dtype=np.float64).reshape(len(levels), 11))


def load_levels(dirpath):
    """Map the arrays written by `export_levels` (read-only memmaps: a rank only touches its own rows)."""
    meta = np.load(os.path.join(dirpath, "meta.npy"))
    out = []
    for l, m in enumerate(meta):
        d = dict(n=int(m[0]), nc=int(m[1]), pre=(int(m[3]), int(m[4]), int(m[5]), float(m[6])),
                 post=(int(m[7]), int(m[8]), int(m[9]), float(m[10])))
        for key in ("A", "S", "P", "R"):
            if key == "S" and not int(m[2]):
                d[key] = None
                continue
            d[key] = tuple(np.load(os.path.join(dirpath, f"L{l}_{key}_{name}.npy"), mmap_mode="r")
                           for name in ("rowptr", "col", "val"))
        out.append(d)
    return out


def _rows(csr, r0, r1, dtype=np.float64):
    rp, ci, va = csr
    lo, hi = int(rp[r0]), int(rp[r1])
    rowptr = (np.asarray(rp[r0:r1 + 1], dtype=np.int64) - lo).astype(np.int32)
    return (np.ascontiguousarray(rowptr), np.ascontiguousarray(ci[lo:hi], dtype=np.int32),
            np.ascontiguousarray(va[lo:hi], dtype=dtype))


class LocalGroup:
    """Rendezvous area of N ranks living in one process (amgh_local_group_*)."""

    def __init__(self, nranks, dtype=np.float64):
        self.lib = require_gpu(dtype)      # (the group belongs to the library instance whose handles will use it)
        self.n = int(nranks)
        g = C.c_void_p()
        hip_check(self.lib.amgh_local_group_create(C.byref(g), self.n), "local_group_create")
        self.h = g.value

    def abort(self):
        self.lib.amgh_local_group_abort(self.h)

    def __del__(self):
        try:
            if self.h:
                self.lib.amgh_local_group_destroy(self.h)
                self.h = None
        except Exception:
            pass


def rccl_unique_id():
    """128-byte id made by rank 0 (ncclGetUniqueId); broadcast it to the other ranks by any means."""
    lib = require_gpu()
    buf = (C.c_char * ID_BYTES)()
    hip_check(lib.amgh_dist_unique_id(buf), "dist_unique_id")
    return bytes(buf)


class _HostBuffer:
    """A vector of the host-executed sharded cycle (device < 0 with a host tail): numpy memory behind the same .ptr as a DeviceBuffer."""

    def __init__(self, n, dtype):
        self.a = np.zeros(int(n), dtype=dtype)
        self.ptr = self.a.ctypes.data


class ShardedHierarchy:
    """One rank of the row-sharded MultiLevel.

    levels : list from `level_arrays` / `load_levels` (the sharded levels; this rank reads its rows only)
    sizes  : rows of level lc (the first collapsed level)
    tail   : MultiLevel of the collapsed levels on rank 0, None elsewhere
    transport : ("rccl", id_bytes), ("ipc", "/fresh_shm_name") or ("local", LocalGroup)
    device < 0 (IPC transport only): plans only — the collective setup in host memory, no GPU anywhere; the solve
    entry points are unavailable, `plan_info` is what such a handle is for.
    gs_mode : "exact" (default), "exact-turns" or "hybrid", see set_gs_mode
    dtype : float64 (default) or float32 = the Float32 instance of the library (the hierarchy's values are rounded
    once, every vector and every operation of the sharded cycle is Float32).
    """

    def __init__(self, levels, n_tail, tail, rank, nranks, device, transport, dtype=np.float64, gs_mode="exact", host_tail=None):
        self.rank, self.nranks, self.device = int(rank), int(nranks), int(device)
        # device < 0: plans only — unless the collapsed levels come as a host function (host_tail: b -> x, one visit from x = 0,
        # on the owner; anything callable or None elsewhere): the library then EXECUTES the sharded cycle in host memory
        self.host_exec = self.device < 0 and host_tail is not None
        self.plans_only = self.device < 0 and not self.host_exec
        self.dtype = np.dtype(np.float32 if np.dtype(dtype).itemsize == 4 else np.float64)
        self.lib = hip_lib(self.dtype) if self.device < 0 else require_gpu(self.dtype)
        h = C.c_void_p()
        kind, arg = transport
        if self.device < 0 and kind != "ipc":
            raise AMGError("device < 0 (plans only / host execution) needs the IPC transport")
        if kind == "ipc":
            hip_check(self.lib.amgh_dist_create_ipc(C.byref(h), self.device, self.rank, self.nranks,
                                                    str(arg).encode()), "dist_create_ipc")
        elif kind == "rccl":
            idb = (C.c_char * ID_BYTES).from_buffer_copy(arg)
            hip_check(self.lib.amgh_dist_create_rccl(C.byref(h), self.device, self.rank, self.nranks, idb), "dist_create_rccl")
        elif kind == "local":
            self._group = arg
            hip_check(self.lib.amgh_dist_create_local(C.byref(h), self.device, self.rank, arg.h), "dist_create_local")
        else:
            raise AMGError(f"unknown transport {kind!r}")
        self.h = h.value
        self.lc = len(levels)
        N = self.nranks
        self.cuts = [row_cuts(d["n"], N) for d in levels]
        tail_cuts = np.array([0] + [n_tail] * N, dtype=np.int64)     # everything below: rank 0
        self.cuts.append(tail_cuts)
        for l, d in enumerate(levels):
            rc, cc = self.cuts[l], self.cuts[l + 1]
            r0, r1 = int(rc[self.rank]), int(rc[self.rank + 1])
            c0, c1 = int(cc[self.rank]), int(cc[self.rank + 1])
            A = _rows(d["A"], r0, r1, self.dtype)
            S = _rows(d["S"], r0, r1, self.dtype) if d["S"] is not None else (None, None, None)
            P = _rows(d["P"], r0, r1, self.dtype)
            R = _rows(d["R"], c0, c1, self.dtype)
            pre = amgh_smoother_t(d["pre"][0], d["pre"][1], d["pre"][2], 0, d["pre"][3])
            post = amgh_smoother_t(d["post"][0], d["post"][1], d["post"][2], 0, d["post"][3])
            ptr = lambda a: a.ctypes.data if a is not None else None  # noqa: E731
            hip_check(self.lib.amgh_dist_push_level(
                self.h, d["n"], d["nc"], rc.ctypes.data, cc.ctypes.data, ptr(A[0]), ptr(A[1]), ptr(A[2]),
                ptr(S[0]), ptr(S[1]), ptr(S[2]), ptr(P[0]), ptr(P[1]), ptr(P[2]), ptr(R[0]), ptr(R[1]), ptr(R[2]),
                C.byref(pre), C.byref(post)), "dist_push_level")
        self.tail = None
        if self.host_exec:
            from ._libs import COARSE_FN, COARSE_FN_F32
            fn = host_tail if callable(host_tail) else None
            dt = self.dtype

            def _cb(_user, b_ptr, x_ptr, n):
                try:
                    bb = np.ctypeslib.as_array(b_ptr, shape=(n,)).astype(np.float64)
                    np.ctypeslib.as_array(x_ptr, shape=(n,))[:] = np.asarray(fn(bb), dtype=dt)
                    return 0
                except Exception:  # noqa: BLE001
                    return -1
            self._host_tail_cb = (COARSE_FN if dt.itemsize == 8 else COARSE_FN_F32)(_cb)
            hip_check(self.lib.amgh_dist_set_host_tail(self.h, C.cast(self._host_tail_cb, C.c_void_p), None), "dist_set_host_tail")
        tail_job = None
        if tail is not None and not self.plans_only and not self.host_exec:
            if isinstance(tail, DeviceHierarchy) or self.lc == 0:   # built by the caller already / nothing is sharded: the tail's size IS the partition
                self.tail = tail if isinstance(tail, DeviceHierarchy) else DeviceHierarchy(tail, self.device, 1, self.dtype)
                hip_check(self.lib.amgh_dist_set_tail(self.h, self.tail.h), "dist_set_tail")
            else:
                # the collapsed levels' device hierarchy (seconds of smoother schedules) beside amgh_dist_finalize (seconds of
                # halo plans and shard schedules): independent work — the tail is passed once both are done
                import threading
                box = {}

                def _build():
                    try:
                        box["tail"] = DeviceHierarchy(tail, self.device, 1, self.dtype)
                    except BaseException as e:  # noqa: BLE001
                        box["err"] = e
                tail_job = threading.Thread(target=_build)
                tail_job.start()
        try:
            try:
                hip_check(self.lib.amgh_dist_finalize(self.h), "dist_finalize")
            finally:
                if tail_job is not None:
                    tail_job.join()
            if tail_job is not None:
                if "err" in box:
                    raise box["err"]
                self.tail = box["tail"]
                hip_check(self.lib.amgh_dist_set_tail(self.h, self.tail.h), "dist_set_tail")
        except BaseException:
            # nothing of a half-built rank stays behind: the handle first (it may borrow the tail), then the collapsed levels
            self.lib.amgh_dist_destroy(self.h)
            self.h = None
            if tail_job is not None:
                box.pop("tail", None)      # (DeviceHierarchy.__del__ destroys the collapsed levels' handle)
            self.tail = None
            raise
        self.set_gs_mode(gs_mode)
        r0, r1 = C.c_int64(0), C.c_int64(0)
        hip_check(self.lib.amgh_dist_local_range(self.h, 0, C.byref(r0), C.byref(r1)), "dist_local_range")
        self.r0, self.r1 = r0.value, r1.value
        self.nloc = self.r1 - self.r0
        if self.host_exec:
            self._b, self._x = _HostBuffer(max(self.nloc, 1), self.dtype), _HostBuffer(max(self.nloc, 1), self.dtype)
        elif not self.plans_only:
            self._b = DeviceBuffer(max(self.nloc, 1), self.device, dtype=self.dtype)
            self._x = DeviceBuffer(max(self.nloc, 1), self.device, dtype=self.dtype)

    @classmethod
    def from_multilevel(cls, ml, rank, nranks, device, transport, shard_min_rows=200_000, dtype=np.float64, gs_mode="exact", host_tail=None):
        """Every rank holds (or maps) the whole host hierarchy; only rank 0 needs the collapsed levels."""
        if not isinstance(ml, MultiLevel):
            raise AMGError("ml must be a MultiLevel")
        sizes = [l.A.m for l in ml.levels] + [ml.final_A.m]
        lc = num_sharded_levels(sizes, nranks, shard_min_rows)
        tail = None
        if rank == 0:
            tail = MultiLevel(ml.levels[lc:], ml.final_A, ml.coarse_solver, ml.presmoother, ml.postsmoother,
                              ml.symmetry, method=ml.method)
        # host_tail (device < 0): a factory — called on rank 0 with the collapsed levels' MultiLevel, returns the function b -> x
        ht = None if host_tail is None else (host_tail(tail) if rank == 0 else True)
        return cls(level_arrays(ml, lc), sizes[lc], tail, rank, nranks, device, transport, dtype, gs_mode, host_tail=ht)

    GS_MODES = {"hybrid": 0, "exact": 1, "exact-turns": 2}

    def set_gs_mode(self, mode):
        """Gauss-Seidel / SOR across the shards.  "exact": lexicographic order over the whole level (the reference's iterate) —
        as ONE sweep pipelined across the ranks where every rank holds the dataflow layout of its shard (`gs_pipelined`), else
        with the ranks in turn; "exact-turns": always in turn; "hybrid": every shard at once, halo frozen per directional
        sweep (another convergent iteration).  The same on every rank."""
        if mode not in self.GS_MODES:
            raise AMGError(f"gs_mode must be one of {sorted(self.GS_MODES)}, not {mode!r}")
        self.gs_mode = mode
        hip_check(self.lib.amgh_dist_set_gs_mode(self.h, self.GS_MODES[mode]), "dist_set_gs_mode")

    def gs_pipelined(self):
        """Per sharded level: do its Gauss-Seidel / SOR sweeps run as one sweep pipelined across the ranks under "exact"?"""
        return [int(self.lib.amgh_dist_gs_pipelined(self.h, l)) == 1 for l in range(int(self.lib.amgh_dist_num_sharded_levels(self.h)))] if self.device >= 0 else []

    def pipe_serialized(self):
        """True when ranks of one process sharing one device were found not to run concurrently (their streams share a hardware
        queue) and every level was therefore left to the turn loop (`amgh_dist_pipe_serialized`)."""
        return self.device >= 0 and int(self.lib.amgh_dist_pipe_serialized(self.h)) == 1

    def pipe_protocol_failed(self):
        """True when the mailbox protocol probe between neighbouring ranks (`amgh_dist_pipe_protocol_failed`) did not come back
        right at finalize — mapping, peer access or visibility across devices — and every level was left to the turn loop."""
        return self.device >= 0 and int(self.lib.amgh_dist_pipe_protocol_failed(self.h)) == 1

    def close(self):
        """Destroy the sharded handle first (it borrows the collapsed levels' handle), then the tail."""
        if getattr(self, "h", None):
            self.lib.amgh_dist_destroy(self.h)
            self.h = None
        self.tail = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- solve phase (this rank's rows of the fine vectors) ----------------------------------------
    def local_range(self, level=0):
        r0, r1 = C.c_int64(0), C.c_int64(0)
        hip_check(self.lib.amgh_dist_local_range(self.h, level, C.byref(r0), C.byref(r1)), "dist_local_range")
        return r0.value, r1.value

    def _up(self, buf, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.nloc:
            raise AMGError(f"expected {self.nloc} local entries, got {host.size}")
        if self.nloc and self.host_exec:
            buf.a[:self.nloc] = host
        elif self.nloc:
            hip_check(self.lib.amgh_dev_upload(self.device, buf.ptr, host.ctypes.data, self.dtype.itemsize * self.nloc), "upload")

    def _down(self, buf):
        out = np.empty(self.nloc, dtype=self.dtype)
        if self.nloc and self.host_exec:
            out[:] = buf.a[:self.nloc]
        elif self.nloc:
            hip_check(self.lib.amgh_dev_download(self.device, out.ctypes.data, buf.ptr, self.dtype.itemsize * self.nloc), "download")
        return out

    def set_rhs(self, b_local):
        self._up(self._b, b_local)

    def precond_apply_d(self, cycle=CYCLE_V):
        """ldiv! on the resident right-hand side (set_rhs); result stays on the device.  Enqueue only."""
        hip_check(self.lib.amgh_dist_precond_apply_d(self.h, self._b.ptr, self._x.ptr, cycle), "dist_precond_apply")

    def precond_apply(self, r_local, cycle=CYCLE_V):
        self.set_rhs(r_local)
        self.precond_apply_d(cycle)
        self.sync()
        return self._down(self._x)

    def solve(self, b_local, cycle=CYCLE_V, maxiter=100, abstol=0.0, reltol=None, x0_local=None,
              calculate_residual=True):
        """_solve (multilevel.jl:152-198); returns (x_local, residual history)."""
        reltol = float(np.sqrt(np.finfo(self.dtype).eps)) if reltol is None else float(reltol)
        self.set_rhs(b_local)
        self._up(self._x, np.zeros(self.nloc) if x0_local is None else x0_local)
        hist = np.zeros(maxiter + 1, dtype=self.dtype)
        iters = C.c_int(0)
        hip_check(self.lib.amgh_dist_solve_d(self.h, self._b.ptr, self._x.ptr, cycle, maxiter, abstol, reltol,
                                             int(bool(calculate_residual)), hist.ctypes.data, C.byref(iters)), "dist_solve")
        nh = iters.value + 1 if calculate_residual else 1
        return self._down(self._x), hist[:nh].copy()

    def spmv(self, level, x_local):
        """y = A_level x on this rank's rows (halo exchange included)."""
        r0, r1 = self.local_range(level)
        n = r1 - r0
        if self.host_exec:
            xd, yd = _HostBuffer(max(n, 1), self.dtype), _HostBuffer(max(n, 1), self.dtype)
            xd.a[:n] = np.asarray(x_local, dtype=self.dtype)
            hip_check(self.lib.amgh_dist_spmv_d(self.h, level, xd.ptr, yd.ptr), "dist_spmv")
            return yd.a[:n].copy()
        xd = DeviceBuffer(max(n, 1), self.device, dtype=self.dtype)
        yd = DeviceBuffer(max(n, 1), self.device, dtype=self.dtype)
        isz = self.dtype.itemsize
        if n:
            x_local = np.ascontiguousarray(x_local, dtype=self.dtype)
            hip_check(self.lib.amgh_dev_upload(self.device, xd.ptr, x_local.ctypes.data, isz * n), "upload")
        hip_check(self.lib.amgh_dist_spmv_d(self.h, level, xd.ptr, yd.ptr), "dist_spmv")
        self.sync()
        out = np.empty(n, dtype=self.dtype)
        if n:
            hip_check(self.lib.amgh_dev_download(self.device, out.ctypes.data, yd.ptr, isz * n), "download")
        return out

    def sync(self):
        hip_check(self.lib.amgh_dist_sync(self.h), "dist_sync")

    def barrier(self):
        hip_check(self.lib.amgh_dist_barrier(self.h), "dist_barrier")

    def allreduce(self, values, op="sum"):
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        hip_check(self.lib.amgh_dist_allreduce(self.h, v.ctypes.data, v.size, int(op == "max")), "dist_allreduce")
        return v

    def stats(self, reset=True):
        out = np.zeros(2, dtype=np.int64)
        hip_check(self.lib.amgh_dist_stats(self.h, out.ctypes.data, int(reset)), "dist_stats")
        return {"halo_exchanges": int(out[0]), "halo_bytes_sent": int(out[1])}

    def plan_info(self, level, which=0):
        """Halo plan of x (which = 0; level = lc is the first collapsed level) or of the residual (which = 1)."""
        cnt = np.zeros(5, dtype=np.int64)
        hip_check(self.lib.amgh_dist_plan_info2(self.h, level, which, cnt.ctypes.data, None, None, None, None), "plan_info")
        halo = np.zeros(max(int(cnt[1]), 1), dtype=np.int64)
        send = np.zeros(max(int(cnt[2]), 1), dtype=np.int32)
        sc = np.zeros(self.nranks, dtype=np.int64)
        rc = np.zeros(self.nranks, dtype=np.int64)
        hip_check(self.lib.amgh_dist_plan_info2(self.h, level, which, cnt.ctypes.data, halo.ctypes.data,
                                                send.ctypes.data, sc.ctypes.data, rc.ctypes.data), "plan_info")
        return dict(nloc=int(cnt[0]), halo=halo[:cnt[1]], send_idx=send[:cnt[2]], send_cnt=sc, recv_cnt=rc,
                    interior=(int(cnt[3]), int(cnt[4])))


def run_local_ranks(nranks, fn, devices=None, dtype=np.float64):
    """Run fn(rank, group) on `nranks` threads of this process (LOCAL transport; ctypes releases the GIL inside the
    library, the collectives rendezvous in C++).  Returns the list of results; the first exception is re-raised and
    releases every rank blocked in a collective."""
    group = LocalGroup(nranks, dtype)
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            out[r] = fn(r, group)
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            group.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    real = [e for e in err if e is not None and "invalid state" not in str(e)]
    for e in real + [e for e in err if e is not None]:
        raise e
    return out
