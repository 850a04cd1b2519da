"""ctypes wrapper of the CPU oracle (oracle/amg_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_SO_F32 = os.path.join(_HERE, "liboracle_f32.so")   # the same restatement with real_t = float
vp, i64 = C.c_void_p, C.c_int64


COARSE_FN = C.CFUNCTYPE(C.c_int, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), i64)
COARSE_FN_F32 = C.CFUNCTYPE(C.c_int, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), i64)


class orc_smoother_t(C.Structure):
    _fields_ = [("kind", C.c_int32), ("sweep", C.c_int32), ("iter", C.c_int32), ("pad_", C.c_int32),
                ("omega", C.c_double)]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build(force=False):
    """(Re)build liboracle.so.  It is compiled -O3 -march=native and travels prebuilt from the build container to the
    GPU box: the CPU it was built on is recorded next to it, and a different CPU (or newer sources / Makefile)
    triggers a rebuild instead of running code tuned for another machine."""
    src = os.path.join(_HERE, "amg_oracle.c")
    mk = os.path.join(_HERE, "Makefile")
    tag = _SO + ".cpu"
    built_for = open(tag).read().strip() if os.path.exists(tag) else None
    newest = max(os.path.getmtime(src), os.path.getmtime(mk))
    stale = (not os.path.exists(_SO) or not os.path.exists(_SO_F32) or os.path.getmtime(_SO) < newest
             or os.path.getmtime(_SO_F32) < newest or built_for != cpu_model())
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so", "liboracle_f32.so"])
        with open(tag, "w") as f:
            f.write(cpu_model() + "\n")
    return _SO


_libs = {}


def lib(dtype=np.float64):
    """The restatement for element type `dtype`: float64 (default) or float32."""
    f32 = np.dtype(dtype).itemsize == 4
    _lib = _libs.get(f32)
    if _lib is None:
        build()
        L = C.CDLL(_SO_F32 if f32 else _SO)
        L.orc_create.restype = vp
        L.orc_destroy.argtypes = [vp]
        L.orc_push_level.argtypes = [vp, i64, i64, vp, vp, vp, C.c_int, vp, vp, vp, C.POINTER(orc_smoother_t),
                                     C.POINTER(orc_smoother_t), C.c_int]
        L.orc_set_coarse.argtypes = [vp, i64, vp, vp, vp, vp]
        L.orc_set_coarse_fn.argtypes = [vp, COARSE_FN_F32 if f32 else COARSE_FN, vp]
        L.orc_solve.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, vp, C.POINTER(C.c_int)]
        L.orc_precond.argtypes = [vp, vp, vp, C.c_int]
        L.orc_precond.restype = None
        L.orc_pcg.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp, C.POINTER(C.c_int)]
        L.orc_spmv_arrays.argtypes = [i64, i64, vp, vp, vp, vp, vp, C.c_int]
        L.orc_spmv_arrays.restype = None
        L.orc_smooth_arrays.argtypes = [i64, vp, vp, vp, C.POINTER(orc_smoother_t), C.c_int, vp, vp]
        L.orc_smooth_arrays.restype = i64
        _libs[f32] = _lib = L
    return _lib


def _sm(s):
    return orc_smoother_t(s.kind, s.sweep_code, int(s.iter), 0, float(s.omega))


def spmv(A, x, adjoint=False, dtype=np.float64):
    """mul!(y, A, x) / mul!(y, A', x) on a SparseMatrixCSC-like (colptr,rowval,nzval,m,n)."""
    x = np.ascontiguousarray(x, dtype=dtype)
    y = np.zeros(A.n if adjoint else A.m, dtype=dtype)
    nz = np.ascontiguousarray(A.nzval, dtype=dtype)
    lib(dtype).orc_spmv_arrays(A.m, A.n, A.colptr.ctypes.data, A.rowval.ctypes.data, nz.ctypes.data,
                               x.ctypes.data, y.ctypes.data, int(adjoint))
    return y


def smooth(config, A, x, b, hermitian=True, dtype=np.float64):
    """smooth!(x, setup_smoother(config, A, symmetry), b) -> new x (input untouched)."""
    x = np.array(x, dtype=dtype, copy=True)
    b = np.ascontiguousarray(b, dtype=dtype)
    s = _sm(config)
    nz = np.ascontiguousarray(A.nzval, dtype=dtype)
    rc = lib(dtype).orc_smooth_arrays(A.m, A.colptr.ctypes.data, A.rowval.ctypes.data, nz.ctypes.data,
                                      C.byref(s), int(hermitian), x.ctypes.data, b.ctypes.data)
    if rc != 0:
        raise ArithmeticError(f"SingularException({rc})")
    return x


class OracleHierarchy:
    """The reference's MultiLevel, on the CPU.  Built from the host hierarchy object
    (arrays borrowed).  RS levels hold R as CSC with P = R'; SA levels hold P with R = P'."""

    def __init__(self, ml, kind=None, dtype=np.float64):
        from amg_amd import HermitianSymmetry
        self.dtype = np.dtype(dtype)
        L = self.L = lib(self.dtype)
        vals = lambda a: np.ascontiguousarray(a, dtype=self.dtype)   # noqa: E731  (borrowed as is for float64)
        self.ml = ml
        self.h = L.orc_create()
        herm = int(isinstance(ml.symmetry, HermitianSymmetry))
        self._keep = []
        for lev in ml.levels:
            A = lev.A
            n, nc = A.m, lev.P.n
            # which of P/R is the stored CSC in the reference: RS stores R (unit rows for C points),
            # SA stores P.  Either view is mathematically the same operator; keep the reference's.
            m_is_R = 1 if (kind or getattr(ml, "method", None) or "sa") == "rs" else 0
            M = lev.R if m_is_R else lev.P
            pre, post = _sm(lev.presmoother), _sm(lev.postsmoother)
            Av, Mv = vals(A.nzval), vals(M.nzval)
            L.orc_push_level(self.h, n, nc, A.colptr.ctypes.data, A.rowval.ctypes.data, Av.ctypes.data, m_is_R,
                             M.colptr.ctypes.data, M.rowval.ctypes.data, Mv.ctypes.data, C.byref(pre),
                             C.byref(post), herm)
            self._keep.append((A, M, Av, Mv))
        fA = ml.final_A
        fv = vals(fA.nzval)
        self._keep.append((fA, fv))
        cs = ml.coarse_solver
        if cs.uses_dense():
            self.op = np.asfortranarray(cs.dense_operator(), dtype=self.dtype)
            L.orc_set_coarse(self.h, fA.m, fA.colptr.ctypes.data, fA.rowval.ctypes.data, fv.ctypes.data,
                             self.op.ctypes.data)
        else:  # big coarsest level: the pluggable `(cs)(x, b)` callable
            L.orc_set_coarse(self.h, fA.m, fA.colptr.ctypes.data, fA.rowval.ctypes.data, fv.ctypes.data, None)

            def _cb(user, bp, xp, n):
                x = np.ctypeslib.as_array(xp, shape=(n,))
                x[...] = cs.host_solve(np.asarray(np.ctypeslib.as_array(bp, shape=(n,)), dtype=np.float64))
                return 0
            self._cb = (COARSE_FN_F32 if self.dtype.itemsize == 4 else COARSE_FN)(_cb)
            L.orc_set_coarse_fn(self.h, self._cb, None)
        self.n = ml.levels[0].A.m if ml.levels else fA.m

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def solve(self, b, x0=None, cycle=0, maxiter=100, abstol=0.0, reltol=None, calculate_residual=True):
        reltol = float(np.sqrt(np.finfo(self.dtype).eps)) if reltol is None else reltol
        b = np.ascontiguousarray(b, dtype=self.dtype)
        x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=self.dtype, copy=True)
        hist = np.zeros(maxiter + 1, dtype=self.dtype)
        it = C.c_int(0)
        self.L.orc_solve(self.h, b.ctypes.data, x.ctypes.data, cycle, maxiter, abstol, reltol,
                        int(calculate_residual), hist.ctypes.data, C.byref(it))
        return x, hist[:(it.value + 1) if calculate_residual else 1].copy(), it.value

    def precond(self, r, cycle=0):
        r = np.ascontiguousarray(r, dtype=self.dtype)
        z = np.zeros_like(r)
        self.L.orc_precond(self.h, r.ctypes.data, z.ctypes.data, cycle)
        return z

    def pcg(self, b, cycle=0, use_precond=True, maxiter=None, abstol=0.0, reltol=None):
        reltol = float(np.sqrt(np.finfo(self.dtype).eps)) if reltol is None else reltol
        maxiter = self.n if maxiter is None else maxiter
        b = np.ascontiguousarray(b, dtype=self.dtype)
        x = np.zeros_like(b)
        hist = np.zeros(maxiter + 1, dtype=self.dtype)
        it = C.c_int(0)
        self.L.orc_pcg(self.h, b.ctypes.data, x.ctypes.data, cycle, int(use_precond), maxiter, abstol, reltol,
                      hist.ctypes.data, C.byref(it))
        return x, hist[:it.value + 1].copy(), it.value
