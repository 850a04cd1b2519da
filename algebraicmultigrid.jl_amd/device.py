"""HBM-resident hierarchy: thin ctypes layer over libamghip's C ABI.

No compute happens in Python; every method is one call through include/amghip.h.
If libamghip.so is missing or no GPU is visible this module raises — the solve
phase has no CPU fallback.
"""
import ctypes as C

import numpy as np

from ._libs import COARSE_FN, COARSE_FN_F32, AMGError, gpu_available, hip_check, hip_lib
from .smoothers import Smoother
from .sparse import SparseMatrixCSC

OP_A, OP_P, OP_R = 0, 1, 2
CYCLE_V, CYCLE_W, CYCLE_F = 0, 1, 2
T_LABELS = ["Presmoother", "Residual eval", "Restriction", "Coarse solve", "Prolongation", "Postsmoother"]


def require_gpu(dtype=None):
    """The library instance for element type `dtype` (None / float64: libamghip.so, float32: libamghip_f32.so)."""
    lib = hip_lib(dtype)
    if lib.amgh_device_count() <= 0:
        raise AMGError("no HIP device visible: the AMG solve phase runs on MI355X only (no CPU fallback)")
    return lib


def _ptr(a):
    return a.ctypes.data if a is not None else None


class DeviceBuffer:
    """A device allocation of `n` reals (amgh_dev_alloc / amgh_dev_free); dtype float64 (default) or float32."""

    def __init__(self, n, device=0, host=None, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.lib = require_gpu(self.dtype)
        self.n = int(n)
        self.device = device
        p = C.c_void_p()
        hip_check(self.lib.amgh_dev_alloc(device, self.dtype.itemsize * max(self.n, 1), C.byref(p)), "dev_alloc")
        self.ptr = p.value
        if host is not None:
            self.upload(host)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.size == self.n
        hip_check(self.lib.amgh_dev_upload(self.device, self.ptr, host.ctypes.data, self.dtype.itemsize * self.n), "upload")

    def download(self):
        out = np.empty(self.n, dtype=self.dtype)
        hip_check(self.lib.amgh_dev_download(self.device, out.ctypes.data, self.ptr, self.dtype.itemsize * self.n), "download")
        return out

    def __del__(self):
        try:
            if self.ptr:
                self.lib.amgh_dev_free(self.device, self.ptr)
                self.ptr = None
        except Exception:
            pass


class DeviceCSR:
    """Stand-alone CSR operator on HBM (amgh_csr_*)."""

    def __init__(self, nrows, ncols, rowptr, col, val, device=0, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.lib = require_gpu(self.dtype)
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=self.dtype)
        self.nrows, self.ncols, self.device = int(nrows), int(ncols), device
        h = C.c_void_p()
        hip_check(self.lib.amgh_csr_create(C.byref(h), device, nrows, ncols, _ptr(rowptr), _ptr(col), _ptr(val)),
                  "csr_create")
        self.h = h.value

    def __del__(self):
        try:
            if self.h:
                self.lib.amgh_csr_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def sync(self):
        hip_check(self.lib.amgh_dev_sync(self.device), "sync")

    def spmv(self, x):
        xd = DeviceBuffer(self.ncols, self.device, x, dtype=self.dtype)
        yd = DeviceBuffer(self.nrows, self.device, dtype=self.dtype)
        hip_check(self.lib.amgh_csr_spmv_d(self.h, xd.ptr, yd.ptr, None), "csr_spmv")
        self.sync()
        return yd.download()

    def residual(self, x, b):
        xd = DeviceBuffer(self.ncols, self.device, x, dtype=self.dtype)
        bd = DeviceBuffer(self.nrows, self.device, b, dtype=self.dtype)
        rd = DeviceBuffer(self.nrows, self.device, dtype=self.dtype)
        hip_check(self.lib.amgh_csr_residual_d(self.h, xd.ptr, bd.ptr, rd.ptr, None), "csr_residual")
        self.sync()
        return rd.download()

    def spmv_add(self, x, y):
        xd = DeviceBuffer(self.ncols, self.device, x, dtype=self.dtype)
        yd = DeviceBuffer(self.nrows, self.device, y, dtype=self.dtype)
        hip_check(self.lib.amgh_csr_spmv_add_d(self.h, xd.ptr, yd.ptr, None), "csr_spmv_add")
        self.sync()
        return yd.download()

    def smooth(self, config, x, b):
        """Run `config.iter` sweeps in place on host vector x (returns new x)."""
        n = self.nrows
        xd = DeviceBuffer(self.ncols, self.device, x, dtype=self.dtype)
        bd = DeviceBuffer(n, self.device, b, dtype=self.dtype)
        tmp = DeviceBuffer(self.ncols, self.device, x, dtype=self.dtype) if config.kind == 2 else None
        cur, other = xd, tmp
        for _ in range(config.iter):
            if config.kind == 2:
                hip_check(self.lib.amgh_csr_jacobi_d(self.h, config.omega, cur.ptr, bd.ptr, other.ptr, None), "jacobi")
                cur, other = other, cur
            elif config.kind in (1, 3):
                sor = int(config.kind == 3)
                if config.sweep_code in (0, 2):
                    hip_check(self.lib.amgh_csr_gs_d(self.h, 0, config.omega, sor, cur.ptr, bd.ptr, None), "gs fwd")
                if config.sweep_code in (1, 2):
                    hip_check(self.lib.amgh_csr_gs_d(self.h, 1, config.omega, sor, cur.ptr, bd.ptr, None), "gs bwd")
        self.sync()
        return cur.download()[:n]


def smoother_matrix_csr(A, symmetry):
    """CSR arrays of the matrix a smoother sweeps row-wise.

    HermitianSymmetry(): the "fast" smoothers read CSC column i as row i
    (smoother.jl:81-86,128-134) -> the CSC arrays as they are.
    NoSymmetry(): sweeps over true rows -> CSR of A = CSC arrays of A'."""
    from .hierarchy import HermitianSymmetry
    A = SparseMatrixCSC.coerce(A)
    if symmetry is None or isinstance(symmetry, HermitianSymmetry):
        return A.colptr, A.rowval, A.nzval
    return A.csr_arrays()


def smooth_standalone(config, A, x, b, symmetry=None, dtype=np.float64):
    from .hierarchy import HermitianSymmetry
    A = SparseMatrixCSC.coerce(A)
    if symmetry is not None and not isinstance(symmetry, HermitianSymmetry):
        config.check_no_symmetry(A)
    rp, ci, va = smoother_matrix_csr(A, symmetry)
    op = DeviceCSR(A.m, A.n, rp, ci, va, dtype=dtype)
    x[...] = op.smooth(config, np.asarray(x, dtype=dtype), np.asarray(b, dtype=dtype))


class DeviceHierarchy:
    """MultiLevel on HBM: amgh_create / push_level / set_coarse / finalize."""

    def __init__(self, ml, device=0, nrhs=1, dtype=np.float64):
        from .hierarchy import HermitianSymmetry
        self._open(device, nrhs, isinstance(ml.symmetry, HermitianSymmetry), dtype)
        for lev in ml.levels:
            self.push_begin(lev.A, lev.presmoother, lev.postsmoother)
            self.push_end(lev)
        self.finish(ml)

    # ---- construction, level by level (the setup phase drives these itself when it builds on the GPU: push_begin
    # ---- of a level runs on a worker thread while the host does that level's C/F splitting) ------------------------
    @classmethod
    def incremental(cls, device, nrhs, hermitian, dtype=np.float64):
        self = cls.__new__(cls)
        self._open(device, nrhs, hermitian, dtype)
        return self

    def _open(self, device, nrhs, hermitian, dtype=np.float64):
        self.dtype = np.dtype(dtype)      # the arithmetic type of this handle: libamghip.so or its Float32 instance
        self.lib = require_gpu(self.dtype)
        self.device = device
        self.ml = None
        self.nrhs = int(nrhs)
        self.hermitian = bool(hermitian)
        self.h = None
        h = C.c_void_p()
        hip_check(self.lib.amgh_create(C.byref(h), device, self.nrhs), "create")
        self.h = h.value

    def _level_args(self, A, presmoother, postsmoother):
        """(n, A rows, S rows or NULLs, pre, post) as amgh_push_level_begin / amgh_level_prepare take them; the arrays are
        kept alive by the returned tuple."""
        n = A.m
        Ar, Ac, Av = A.csr_arrays()                  # true A rows
        Av = self._vals(Av)
        if self.hermitian and not A.is_symmetric():
            Sr, Sc, Sv = A.colptr, A.rowval, self._vals(A.nzval)  # column i read as row i
        else:
            Sr = Sc = Sv = None                      # S == A
        for s in (presmoother, postsmoother):
            if not isinstance(s, Smoother):
                raise AMGError(f"unsupported smoother {s!r}")
        pre, post = presmoother.c_struct(), postsmoother.c_struct()
        return (n, _ptr(Ar), _ptr(Ac), _ptr(Av), _ptr(Sr), _ptr(Sc), _ptr(Sv), C.byref(pre), C.byref(post)), (Ar, Ac, Av, Sr, Sc, Sv, pre, post)

    def push_begin(self, A, presmoother, postsmoother):
        """amgh_push_level_begin: A (and S) to HBM, smoother schedules.  Needs nothing but A."""
        import os
        import time
        t_lev = time.perf_counter()
        args, keep = self._level_args(A, presmoother, postsmoother)
        hip_check(self.lib.amgh_push_level_begin(self.h, *args), "push_level_begin")
        del keep
        if os.environ.get("AMGH_VERBOSE"):
            print(f"[amghip] n={A.m} python: push_level_begin total {time.perf_counter() - t_lev:.2f} s",
                  file=__import__("sys").stderr, flush=True)

    def prepare(self, A, presmoother, postsmoother):
        """amgh_level_prepare: the same work into a free-standing level (no handle state: callable from several host
        threads at once).  Returns the level's address for push_prepared / free_prepared."""
        import os
        import time
        t_lev = time.perf_counter()
        args, keep = self._level_args(A, presmoother, postsmoother)
        out = C.c_void_p()
        hip_check(self.lib.amgh_level_prepare_nrhs(self.device, int(self.nrhs), *args, C.byref(out)), "level_prepare")
        del keep
        if os.environ.get("AMGH_VERBOSE"):
            print(f"[amghip] n={A.m} python: level_prepare total {time.perf_counter() - t_lev:.2f} s (clock {t_lev % 1000:.2f} .. {time.perf_counter() % 1000:.2f})",
                  file=__import__("sys").stderr, flush=True)
        return out.value

    def push_prepared(self, level):
        """amgh_push_level_prepared: the prepared level (an address, or a Future of one) becomes the pending level."""
        fut = level if hasattr(level, "result") else None
        if fut is not None:
            level = fut.result()
            fut._taken = True          # (whatever happens below, the address is not the future's any more)
        rc = self.lib.amgh_push_level_prepared(self.h, level)
        if rc != 0:
            self.lib.amgh_level_free(level)
        hip_check(rc, "push_level_prepared")

    def free_prepared(self, level):
        self.lib.amgh_level_free(level)

    def push_end(self, lev):
        """amgh_push_level_end: P and R of the level whose A was pushed last."""
        n, nc = lev.A.m, lev.P.n
        # CSR of P (n x nc) = CSC arrays of R (nc x n), and vice versa.
        Pr, Pc, Pv = lev.R.colptr, lev.R.rowval, self._vals(lev.R.nzval)
        Rr, Rc, Rv = lev.P.colptr, lev.P.rowval, self._vals(lev.P.nzval)
        if lev.R.shape != (nc, n) or lev.P.shape != (n, nc):
            raise AMGError("Level: P must be n x nc and R nc x n")
        import os
        import time
        t0 = time.perf_counter()
        hip_check(self.lib.amgh_push_level_end(self.h, nc, _ptr(Pr), _ptr(Pc), _ptr(Pv), _ptr(Rr), _ptr(Rc), _ptr(Rv)),
                  "push_level_end")
        if os.environ.get("AMGH_VERBOSE"):
            print(f"[amghip] n={n} python: push_level_end {time.perf_counter() - t0:.2f} s (clock {t0 % 1000:.2f} .. {time.perf_counter() % 1000:.2f})",
                  file=__import__("sys").stderr, flush=True)

    def push_abort(self):
        """amgh_push_level_abort: the level begun last turned out to be the coarsest one."""
        hip_check(self.lib.amgh_push_level_abort(self.h), "push_level_abort")

    def _vals(self, v):
        """Matrix values in this handle's arithmetic type (the host mirror stores Float64 throughout)."""
        return v if self.dtype.itemsize == 8 else np.ascontiguousarray(v, dtype=self.dtype)

    def finish(self, ml):
        """Coarsest level + amgh_finalize; `ml` is the hierarchy the pushed levels belong to."""
        self.ml = ml
        fA = ml.final_A
        fr, fc, fv = fA.csr_arrays()
        fv = self._vals(fv)
        cs = ml.coarse_solver
        if getattr(cs, "uses_dense", lambda: True)():
            op = np.asfortranarray(cs.dense_operator(), dtype=self.dtype)
            hip_check(self.lib.amgh_set_coarse(self.h, fA.m, _ptr(fr), _ptr(fc), _ptr(fv), _ptr(op)), "set_coarse")
        else:
            # pluggable host coarse solver: the reference's `(cs)(x, b)` protocol
            def _cb(user, bp, xp, n):
                try:
                    b = np.ctypeslib.as_array(bp, shape=(n,))
                    x = np.ctypeslib.as_array(xp, shape=(n,))
                    x[...] = cs.host_solve(np.asarray(b, dtype=np.float64))
                    return 0
                except Exception:  # never let an exception cross the C boundary
                    return 1
            self._coarse_cb = (COARSE_FN if self.dtype.itemsize == 8 else COARSE_FN_F32)(_cb)
            hip_check(self.lib.amgh_set_coarse_host(self.h, fA.m, _ptr(fr), _ptr(fc), _ptr(fv), self._coarse_cb, None),
                      "set_coarse_host")
        hip_check(self.lib.amgh_finalize(self.h), "finalize")
        # the collapsed coarse tail's operator for V-cycles now, inside the setup (W / F: at their first cycle)
        hip_check(self.lib.amgh_tail_dense_build(self.h, 0), "tail_dense_build")
        self.n = ml.levels[0].A.m if ml.levels else fA.m

    def tail_dense_info(self, cycle=0):
        """(level, rows, build_ms) of the collapsed coarse tail for a cycle type (level -1: none)."""
        lv, rows, ms = C.c_int(-1), C.c_int64(0), C.c_double(0.0)
        hip_check(self.lib.amgh_tail_dense_info(self.h, int(cycle), C.byref(lv), C.byref(rows), C.byref(ms)), "tail_dense_info")
        return lv.value, rows.value, ms.value

    def __del__(self):
        try:
            if self.h:
                self.lib.amgh_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- solve phase ---------------------------------------------------------
    def solve(self, b, x0, cycle, maxiter, abstol, reltol, calculate_residual, log):
        # n x bs blocks travel column-major, as Julia holds them
        b = np.asfortranarray(b, dtype=self.dtype)
        x = np.array(x0, dtype=self.dtype, copy=True, order="F")
        hist = np.zeros(maxiter + 1, dtype=self.dtype)
        iters = C.c_int(0)
        hip_check(self.lib.amgh_solve(self.h, b.ctypes.data, x.ctypes.data, cycle, maxiter, abstol, reltol,
                                      int(bool(calculate_residual)), hist.ctypes.data, C.byref(iters)), "solve")
        n_hist = (iters.value + 1) if calculate_residual else 1
        return x, hist[:n_hist].copy(), iters.value

    def precond_apply(self, r, cycle=CYCLE_V):
        r = np.asfortranarray(r, dtype=self.dtype)
        z = np.empty_like(r, order="F")
        hip_check(self.lib.amgh_precond_apply(self.h, r.ctypes.data, z.ctypes.data, cycle), "precond_apply")
        return z

    def pcg(self, b, cycle=CYCLE_V, use_precond=True, maxiter=None, abstol=0.0, reltol=None):
        b = np.ascontiguousarray(b, dtype=self.dtype)
        maxiter = self.n if maxiter is None else int(maxiter)
        reltol = float(np.sqrt(np.finfo(self.dtype).eps)) if reltol is None else float(reltol)
        x = np.zeros_like(b)
        hist = np.zeros(maxiter + 1, dtype=self.dtype)
        iters = C.c_int(0)
        hip_check(self.lib.amgh_pcg(self.h, b.ctypes.data, x.ctypes.data, cycle, int(bool(use_precond)), maxiter,
                                    abstol, reltol, hist.ctypes.data, C.byref(iters)), "pcg")
        return x, hist[:iters.value + 1].copy(), iters.value

    # ---- per-level hooks -----------------------------------------------------
    def spmv(self, level, which, x):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        nr = self._op_rows(level, which)
        y = np.empty(nr, dtype=self.dtype)
        hip_check(self.lib.amgh_level_spmv(self.h, level, which, x.ctypes.data, y.ctypes.data), "level_spmv")
        return y

    def _op_rows(self, level, which):
        L = len(self.ml.levels)
        if level == L:
            return self.ml.final_A.m
        lev = self.ml.levels[level]
        return lev.P.n if which == OP_R else lev.A.m

    def smooth(self, level, post, x, b):
        x = np.array(x, dtype=self.dtype, copy=True)
        b = np.ascontiguousarray(b, dtype=self.dtype)
        hip_check(self.lib.amgh_level_smooth(self.h, level, int(post), x.ctypes.data, b.ctypes.data), "level_smooth")
        return x

    def bench_op(self, level, which, reps=20, warmup=3):
        ms = C.c_double(0)
        hip_check(self.lib.amgh_bench_op(self.h, level, which, reps, warmup, C.byref(ms)), "bench_op")
        return ms.value

    def device_bytes(self):
        return int(self.lib.amgh_device_bytes(self.h))

    def device_bytes_detail(self):
        out = np.zeros(8, dtype=np.int64)
        hip_check(self.lib.amgh_device_bytes_detail(self.h, out.ctypes.data), "device_bytes_detail")
        keys = ("natural_APR", "level_ordered_csr", "unmerged_slots", "merged_csr", "merged_slots", "prepass_triangles",
                "block_and_vectors", "workspace")
        return dict(zip(keys, map(int, out)))

    def gs_dependency_levels(self, level):
        return int(self.lib.amgh_gs_num_dependency_levels(self.h, level))

    def gs_sweep_steps(self, level, backward=False):
        """Sequential steps of one GS sweep as executed (merged groups / block steps / dependency levels)."""
        return int(self.lib.amgh_gs_num_sweep_steps(self.h, level, int(backward)))

    def gs_sweep_stats(self, level, backward=False):
        """{launches, rows, entries, slot_entries (padding included), tri_entries, levels_per_group} of one sweep."""
        out = np.zeros(6, dtype=np.int64)
        hip_check(self.lib.amgh_gs_sweep_stats(self.h, level, int(backward), out.ctypes.data), "gs_sweep_stats")
        return dict(zip(("launches", "rows", "entries", "slot_entries", "tri_entries", "levels_per_group"), map(int, out)))

    def profile(self, on=True):
        hip_check(self.lib.amgh_profile_enable(self.h, int(on)), "profile_enable")

    def profile_read(self, reset=True):
        L1 = len(self.ml.levels) + 1
        out = np.zeros(6 * L1, dtype=np.float64)
        hip_check(self.lib.amgh_profile_read(self.h, out.ctypes.data, int(reset)), "profile_read")
        return {lab: out[i * L1:(i + 1) * L1].copy() for i, lab in enumerate(T_LABELS)}

    def timer_begin(self):
        hip_check(self.lib.amgh_timer_begin(self.h), "timer_begin")

    def timer_end(self):
        ms = C.c_double(0)
        hip_check(self.lib.amgh_timer_end(self.h, C.byref(ms)), "timer_end")
        return ms.value
