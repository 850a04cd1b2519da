"""Host mirror of the reference's setup-phase interface and hierarchy types.

Names, argument meaning and error behaviour follow the reference
(/root/reference/src): `ruge_stuben` (classical.jl:6-34), `smoothed_aggregation`
(aggregation.jl:66-114), `Level`/`MultiLevel` (multilevel.jl:1-21),
`Classical`/`SymmetricStrength` (strength.jl), `RS` (splitting.jl),
`StandardAggregation` (aggregate.jl), `JacobiProlongation`, `fit_candidates`
(aggregation.jl), `Pinv`/`QRSolver` (coarse_solver.jl), `poisson` (gallery.jl).
The heavy lifting is libamgsetup (C++); the solve phase is libamghip (HIP).
"""
import ctypes as C
import os

import numpy as np

from ._libs import AMGError, amgs_options, setup_lib
from .smoothers import GaussSeidel, Smoother
from .sparse import SparseMatrixCSC


# ---- symmetry tags (utils.jl:1-19) ----------------------------------------
class HermitianSymmetry:
    pass


class NoSymmetry:
    pass


# ---- gallery (gallery.jl) ---------------------------------------------------
def poisson(sz):
    """poisson(n) / poisson((n1,...,nN)): gallery.jl:1-63 (first axis fastest)."""
    dims = (int(sz),) if np.isscalar(sz) else tuple(int(s) for s in sz)
    arr = (C.c_int64 * len(dims))(*dims)
    return SparseMatrixCSC(setup_lib().amgs_poisson(len(dims), arr))


# ---- strength (strength.jl) -------------------------------------------------
class Classical:
    def __init__(self, theta=0.25):
        self.theta = float(theta)

    def __call__(self, At):
        At = SparseMatrixCSC.coerce(At)
        S, T = C.c_void_p(), C.c_void_p()
        rc = setup_lib().amgs_classical_strength(At._h, self.theta, C.byref(S), C.byref(T))
        if rc != 0:
            raise AMGError(setup_lib().amgs_last_error().decode())
        return SparseMatrixCSC(S.value), SparseMatrixCSC(T.value)


class SymmetricStrength:
    def __init__(self, theta=0.0):
        self.theta = float(theta)

    def __call__(self, A, bsr_flag=False):
        if np.iscomplexobj(getattr(A, "data", np.zeros(0))):
            raise AMGError("Symmetric strength not implemented for complex matrices.")  # strength.jl:124-126
        A = SparseMatrixCSC.coerce(A)
        S = SparseMatrixCSC(setup_lib().amgs_symmetric_strength(A._h, self.theta, int(bool(bsr_flag))))
        return S, S


# ---- splitting (splitting.jl) ----------------------------------------------
F_NODE, C_NODE, U_NODE = 0, 1, 2


class RS:
    def __call__(self, S):
        """RS()(S): removes the diagonal of S in place, returns the C/F splitting."""
        S = SparseMatrixCSC.coerce(S)
        out = np.zeros(S.m, dtype=np.int32)
        rc = setup_lib().amgs_rs_splitting(S._h, out.ctypes.data)
        if rc != 0:
            raise AMGError(setup_lib().amgs_last_error().decode())
        # S was modified in place: refresh its views
        S.__init__(S._h, S._owner)
        return out


def direct_interpolation(At, T, splitting):
    """classical.jl:57-68 -> (P, R) with P = R'."""
    At = SparseMatrixCSC.coerce(At)
    T = SparseMatrixCSC.coerce(T)
    sp = np.ascontiguousarray(splitting, dtype=np.int32)
    R = SparseMatrixCSC(setup_lib().amgs_direct_interpolation(At._h, T._h, sp.ctypes.data))
    return R.transpose(), R


# ---- aggregation (aggregate.jl, aggregation.jl) ----------------------------
class StandardAggregation:
    def __call__(self, S):
        S = SparseMatrixCSC.coerce(S)
        return SparseMatrixCSC(setup_lib().amgs_standard_aggregation(S._h))


def fit_candidates(AggOp, B, tol=1e-10):
    """aggregation.jl:161-230.  B 1-D -> Vector method, 2-D -> QR method."""
    AggOp = SparseMatrixCSC.coerce(AggOp)
    B = np.asarray(B, dtype=np.float64)
    vector = B.ndim == 1
    nB = 1 if vector else B.shape[1]
    Bf = np.asfortranarray(B.reshape(B.shape[0], nB))
    bc, nc = C.c_void_p(), C.c_int64()
    Q = SparseMatrixCSC(setup_lib().amgs_fit_candidates(AggOp._h, Bf.ctypes.data, nB, int(vector), tol,
                                                        C.byref(bc), C.byref(nc)))
    n_coarse = nc.value
    buf = np.ctypeslib.as_array(C.cast(bc, C.POINTER(C.c_double)), shape=(max(n_coarse * nB, 1),))[:n_coarse * nB].copy()
    setup_lib().amgs_free(bc)
    R = buf if vector else buf.reshape((n_coarse, nB), order="F")
    return Q, R


class LocalWeighting:
    pass


class JacobiProlongation:
    def __init__(self, omega):
        self.omega = float(omega)

    def __call__(self, A, T, S=None, B=None, degree=1, weighting=None):
        if degree != 1 or (weighting is not None and not isinstance(weighting, LocalWeighting)):
            raise AMGError("only degree=1 LocalWeighting() is built (the reference default)")
        A = SparseMatrixCSC.coerce(A)
        T = SparseMatrixCSC.coerce(T)
        return SparseMatrixCSC(setup_lib().amgs_jacobi_prolongation(A._h, T._h, self.omega))


# ---- coarse solvers (coarse_solver.jl) --------------------------------------
DENSE_COARSE_MAX = 2048  # above this the coarse solve is a host callback, not a dense HBM operator


class CoarseSolver:
    """coarse_solver(A) -> callable (x, b)  (coarse_solver.jl:2).

    `dense_operator()` is what the GPU applies as one GEMV when the coarsest level is small
    (the normal case, n <= max_coarse); for a large coarsest level it returns None and the
    device calls back into `host_solve(b)` (amgh_set_coarse_host)."""

    def __init__(self, A):
        self.A = SparseMatrixCSC.coerce(A)
        self._op = None
        self._lu = None

    def uses_dense(self):
        return self.A.m <= DENSE_COARSE_MAX

    def dense_operator(self):
        raise NotImplementedError

    def host_solve(self, b):
        if self.uses_dense():
            return self.dense_operator() @ b
        if self._lu is None:  # sparse direct factorisation for a big coarsest level
            import scipy.sparse.linalg as spla
            self._lu = spla.splu(self.A.to_scipy())
        return self._lu.solve(np.asarray(b, dtype=np.float64))

    def __call__(self, x, b):
        x[...] = self.host_solve(b)
        return x


class Pinv(CoarseSolver):
    """pinv(Matrix(A)) (coarse_solver.jl:9-16); Julia's default rtol = eps * min(size)."""

    def dense_operator(self):
        if self._op is None:
            M = self.A.toarray()
            if M.size == 0:
                self._op = np.zeros_like(M)
            else:
                self._op = np.linalg.pinv(M, rcond=np.finfo(np.float64).eps * min(M.shape))
        return self._op

    def __repr__(self):
        return "Pinv"


def _qr_basic_operator(M):
    """The operator b -> x of a rank-deficient `qr(A) \\ b` in SuiteSparseQR's sense (what `qr` of a SparseMatrixCSC is in
    Julia, coarse_solver.jl:69,79): Householder QR in which a column whose remaining part has 2-norm <= tol is DEAD
    (Heath's rank detection; SPQR's default tol = 20 (m + n) eps max_j ||A[:, j]||_2), and the BASIC solution — the live
    columns solve R11 y = (Q' b)[:r] by back substitution, the dead ones are zero.  Not the minimum-norm solution of
    `pinv` / `lstsq`: the two differ by a null-space vector.  SPQR also reorders the columns (COLAMD) before it
    factorises; that order is not reproduced here (natural order), so WHICH column of a dependent set is dead can differ
    from Julia's — stated in DESIGN.md section 8 ("unpinned")."""
    M = np.array(M, dtype=np.float64)
    m, n = M.shape
    eps = np.finfo(np.float64).eps
    tol = 20.0 * (m + n) * eps * max(float(np.sqrt((M * M).sum(axis=0)).max()) if n else 0.0, 0.0)
    W = M.copy()                 # becomes R (live rows on top)
    QtI = np.eye(m)              # becomes Q' (reflectors applied to the identity)
    live = []
    r = 0
    for j in range(n):
        if r >= m:
            break
        v = W[r:, j].copy()
        nv = float(np.linalg.norm(v))
        if nv <= tol:
            continue             # dead column: its x stays zero
        alpha = -nv if v[0] >= 0 else nv
        v[0] -= alpha
        vn = float(np.linalg.norm(v))
        if vn > 0.0:
            v /= vn
            W[r:, j:] -= 2.0 * np.outer(v, v @ W[r:, j:])
            QtI[r:, :] -= 2.0 * np.outer(v, v @ QtI[r:, :])
        live.append(j)
        r += 1
    X = np.zeros((n, m))
    if r:
        import scipy.linalg as sla
        R11 = np.triu(W[:r, live])
        X[live, :] = sla.solve_triangular(R11, QtI[:r, :])
    return X


class QRSolver(CoarseSolver):
    """qr(A) \\ b (coarse_solver.jl:66-81).  For the full-rank coarse matrices AMG
    produces this is A^-1 b; the dense operator is formed by a Householder-QR
    solve against the identity.  Rank-deficient input: the BASIC solution of a rank-revealing
    Householder QR (`_qr_basic_operator`: SuiteSparseQR's semantics — dead columns get zero), not the
    minimum-norm one."""

    def dense_operator(self):
        if self._op is None:
            M = self.A.toarray()
            n = M.shape[0]
            if n == 0:
                self._op = np.zeros_like(M)
            else:
                Q, R = np.linalg.qr(M)
                d = np.abs(np.diag(R))
                if d.min() <= np.finfo(np.float64).eps * n * max(d.max(), 1e-300):
                    self._op = _qr_basic_operator(M)
                else:
                    import scipy.linalg as sla
                    self._op = sla.solve_triangular(R, Q.T)
        return self._op

    def __repr__(self):
        return "QRSolver"


class _LinearSolveWrapperInternal(CoarseSolver):
    """LinearSolveWrapperInternal (coarse_solver.jl:24-42): a third-party factorisation, set up once (`init(linprob,
    alg)`), solved per right-hand-side column (`p.linsolve.b = b[:, i]; x[:, i] = solve!(p.linsolve).u`).  Always the
    host-callback protocol (amgh_set_coarse_host): the library calls `(cs)(x, b)` on every coarse solve."""

    def __init__(self, A, alg):
        super().__init__(A)
        self.alg = alg
        self._fact = alg.factorize(self.A.to_scipy())

    def uses_dense(self):
        return False

    def host_solve(self, b):
        b = np.asarray(b, dtype=np.float64)
        if b.ndim == 2:
            return np.column_stack([self._fact(b[:, i]) for i in range(b.shape[1])])
        return self._fact(b)

    def __repr__(self):
        return repr(self.alg)


class LinearSolveWrapper:
    """LinearSolveWrapper(alg) (coarse_solver.jl:50-58): pass a linear-solver algorithm of another package as the coarse
    solver, `ruge_stuben(A; coarse_solver = LinearSolveWrapper(alg))`.  `alg` needs `factorize(A_scipy_csc) -> solve(b)`;
    `SuperLUFactorization` / `UMFPACKLikeFactorization` below wrap SciPy's sparse direct solvers the way LinearSolve.jl
    wraps KLU / UMFPACK."""

    def __init__(self, alg):
        if not hasattr(alg, "factorize"):
            raise AMGError("LinearSolveWrapper: alg must provide factorize(A) -> callable solve(b)")
        self.alg = alg

    def __call__(self, A):
        return _LinearSolveWrapperInternal(A, self.alg)


class SuperLUFactorization:
    """SciPy's SuperLU (sparse LU with partial pivoting) as a LinearSolveWrapper algorithm."""

    def factorize(self, A):
        import scipy.sparse.linalg as spla
        return spla.splu(A.tocsc()).solve

    def __repr__(self):
        return "SuperLUFactorization()"


class DenseLUFactorization:
    """LAPACK getrf / getrs on the dense coarse matrix (LinearSolve.jl's LUFactorization on a small matrix)."""

    def factorize(self, A):
        import scipy.linalg as sla
        lu = sla.lu_factor(A.toarray())
        return lambda b: sla.lu_solve(lu, b)

    def __repr__(self):
        return "DenseLUFactorization()"


_default_coarse_solver = QRSolver  # coarse_solver.jl:84


# ---- hierarchy types (multilevel.jl:1-114) ----------------------------------
class Level:
    def __init__(self, A, P, R, presmoother, postsmoother):
        self.A = SparseMatrixCSC.coerce(A)
        self.P = SparseMatrixCSC.coerce(P)
        self.R = SparseMatrixCSC.coerce(R)
        self.presmoother = presmoother
        self.postsmoother = postsmoother

    def __repr__(self):
        return f"Level with R {self.R.shape} | A {self.A.shape} | P {self.P.shape}"


class MultiLevel:
    """levels, final_A, coarse_solver, presmoother, postsmoother (multilevel.jl:14-21).

    The workspace of the reference (`MultiLevelWorkspace`) lives in HBM inside
    the libamghip handle, created lazily by `device()`.
    """

    def __init__(self, levels, final_A, coarse_solver, presmoother, postsmoother, symmetry=None, _hier=None,
                 method=None):
        self.method = method  # "rs" | "sa" | None (assembled by the caller)
        self.levels = list(levels)
        self.final_A = SparseMatrixCSC.coerce(final_A)
        self.coarse_solver = coarse_solver
        self.presmoother = presmoother
        self.postsmoother = postsmoother
        self.symmetry = symmetry if symmetry is not None else HermitianSymmetry()
        self._hier = _hier
        self._dev = {}

    def __len__(self):
        return len(self.levels) + 1

    def device(self, device=0, nrhs=1, dtype=None):
        """The HBM-resident hierarchy (libamghip handle) for workspace block size `nrhs`
        (the reference's `Val{bs}`, multilevel.jl:28-35); built on first use.  dtype: the arithmetic type of the
        handle — float64 (default), or float32 = the Float32 instance of the library (eltype(A) == Float32)."""
        f32 = dtype is not None and np.dtype(dtype).itemsize == 4
        key = (device, int(nrhs), "f32") if f32 else (device, int(nrhs))
        if key not in self._dev:
            from .device import DeviceHierarchy
            self._dev[key] = DeviceHierarchy(self, device, int(nrhs), np.float32 if f32 else np.float64)
        return self._dev[key]

    def __repr__(self):
        total = self.final_A.nnz + sum(l.A.nnz for l in self.levels)
        rows = []
        for i, l in enumerate(self.levels):
            rows.append("   %2d   %10d   %10d [%5.2f%%]" % (i + 1, l.A.m, l.A.nnz, 100.0 * l.A.nnz / total))
        rows.append("   %2d   %10d   %10d [%5.2f%%]" % (len(self.levels) + 1, self.final_A.m, self.final_A.nnz,
                                                      100.0 * self.final_A.nnz / max(total, 1)))
        return ("Multilevel Solver\n-----------------\n"
                f"Operator Complexity: {round(operator_complexity(self), 3)}\n"
                f"Grid Complexity: {round(grid_complexity(self), 3)}\n"
                f"No. of Levels: {len(self)}\n"
                f"Coarse Solver: {self.coarse_solver}\n"
                "Level     Unknowns     NonZeros\n-----     --------     --------\n" + "\n".join(rows) + "\n")


def operator_complexity(ml):
    if ml.levels:
        return (sum(l.A.nnz for l in ml.levels) + ml.final_A.nnz) / ml.levels[0].A.nnz
    return 1.0


def grid_complexity(ml):
    if ml.levels:
        return (sum(l.A.m for l in ml.levels) + ml.final_A.m) / ml.levels[0].A.m
    return 1.0


class _Hier:
    """Owns an amgs_hier*; matrices borrowed from it keep it alive."""

    def __init__(self, h):
        if not h:
            raise AMGError("libamgsetup: " + setup_lib().amgs_last_error().decode())
        self.h = h

    def __del__(self):
        try:
            setup_lib().amgs_hier_free(self.h)
        except Exception:
            pass

    def get(self, level, which):
        return SparseMatrixCSC(setup_lib().amgs_hier_get(self.h, level, which), owner=self)


def _build_multilevel(hier, presmoother, postsmoother, coarse_solver, symmetry, method, eltype=None):
    L = setup_lib().amgs_hier_num_levels(hier.h)
    levels = [Level(hier.get(l, 0), hier.get(l, 1), hier.get(l, 2), presmoother, postsmoother) for l in range(L)]
    final_A = hier.get(L, 0)
    if eltype is not None:  # the element type of the matrix the caller handed in (multilevel.jl:154 promotion)
        final_A.eltype = eltype
        for lev in levels:
            lev.A.eltype = eltype
    if not isinstance(symmetry, HermitianSymmetry):
        # NoSymmetry smoothers need a nonzero stored diagonal (smoother.jl:239-241)
        for lev in levels:
            for s in (presmoother, postsmoother):
                s.check_no_symmetry(lev.A)
    cs = coarse_solver(final_A)
    return MultiLevel(levels, final_A, cs, presmoother, postsmoother, symmetry, _hier=hier, method=method)


def _unwrap(A, symmetry):
    return SparseMatrixCSC.coerce(A), symmetry


def _huge_empty(count, dtype):
    """np.empty whose pages — untouched so far — are advised to be 2 MiB ones (transparent huge pages are in `madvise`
    mode on the target hosts): a download writes hundreds of megabytes front to back, and the sequential C/F splitting
    then chases through them at random; with 4 KiB pages both pay a page fault / a TLB miss per 512 x fewer bytes."""
    a = np.empty(int(count), dtype=dtype)
    if a.nbytes >= (8 << 20) and not os.environ.get("AMGS_NO_HUGEPAGES"):
        try:
            lo = (a.ctypes.data + (2 << 20) - 1) & ~((2 << 20) - 1)
            hi = (a.ctypes.data + a.nbytes) & ~((2 << 20) - 1)
            if hi > lo:
                C.CDLL(None, use_errno=True).madvise(C.c_void_p(lo), C.c_size_t(hi - lo), 14)   # MADV_HUGEPAGE: advice only
        except Exception:   # noqa: BLE001 - no libc / no madvise: plain pages
            pass
    return a


class _DMat:
    """SparseMatrixCSC on HBM (amgh_dmat_*): the operand type of the GPU half of the setup phase."""

    def __init__(self, handle, lib):
        self.h, self.lib = handle, lib
        self.m, self.n, self.nnz = (int(lib.amgh_dmat_rows(handle)), int(lib.amgh_dmat_cols(handle)),
                                    int(lib.amgh_dmat_nnz(handle)))

    @classmethod
    def upload(cls, A, lib, device=0):
        h = C.c_void_p()
        from ._libs import hip_check
        hip_check(lib.amgh_dmat_upload(C.byref(h), device, A.m, A.n, A.colptr.ctypes.data, A.rowval.ctypes.data,
                                       A.nzval.ctypes.data), "dmat_upload")
        return cls(h.value, lib)

    def download(self, values=True):
        from ._libs import hip_check
        cp = _huge_empty(self.n + 1, np.int32)
        rv = _huge_empty(self.nnz, np.int32)
        nz = _huge_empty(self.nnz, np.float64) if values else None
        hip_check(self.lib.amgh_dmat_download(self.h, cp.ctypes.data, rv.ctypes.data, nz.ctypes.data if values else None),
                  "dmat_download")
        return cp, rv, nz

    def to_host(self):
        # straight into the host library's arrays (no staging copy; the device arrays are a valid CSC by construction)
        from ._libs import hip_check
        h = setup_lib().amgs_mat_alloc(self.m, self.n, self.nnz)
        if not h:
            raise AMGError(setup_lib().amgs_last_error().decode())
        out = SparseMatrixCSC(h)
        hip_check(self.lib.amgh_dmat_download(self.h, out.colptr.ctypes.data, out.rowval.ctypes.data if self.nnz else None,
                                              out.nzval.ctypes.data if self.nnz else None), "dmat_download")
        return out

    def __del__(self):
        try:
            if self.h:
                self.lib.amgh_dmat_free(self.h)
                self.h = None
        except Exception:
            pass


class _Pipeline:
    """Calls executed in order on ONE host thread (ctypes releases the GIL inside the library); the first exception
    stops the queue and is re-raised by close()."""

    def __init__(self):
        import queue
        import threading
        self.q = queue.Queue()
        self.exc = None
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if self.exc is not None:
                continue
            fn, args = item
            try:
                fn(*args)
            except BaseException as e:  # noqa: BLE001 - handed to the closing thread
                self.exc = e

    def submit(self, fn, *args):
        self.q.put((fn, args))

    def close(self, reraise=True):
        self.q.put(None)
        self.t.join()
        if reraise and self.exc is not None:
            raise self.exc


def _ruge_stuben_gpu(A, theta, max_levels, max_coarse, hermitian, device=0, builder=None, smoothers=None):
    """extend_hierarchy_rs! (classical.jl:36-55) with strength, interpolation, transposes and R*A*P on the GPU
    (amgh_setup_*, include/amghip.h) and the sequential C/F splitting on the host (amgs_rs_cf_splitting_patterns).
    Returns [(A, P, R), ...], final_A as host matrices — bitwise what the host library builds.

    With `builder` (a DeviceHierarchy.incremental) the solve-phase handle is filled on the way: the upload of a level's
    A and the construction of its smoother schedules (amgh_push_level_begin) need only A, so a second host thread does
    them — and the level's P, R (amgh_push_level_end) once they exist — while this thread goes on with strength / C/F
    splitting / interpolation / R*A*P of this and the following levels."""
    from ._libs import hip_check, hip_lib
    lib, L = hip_lib(), setup_lib()
    if lib.amgh_device_count() <= 0:
        raise AMGError("ruge_stuben(setup='gpu'): no HIP device visible")

    def call2(fn, *args, what=""):
        a, b = C.c_void_p(), C.c_void_p()
        hip_check(fn(*args, C.byref(a), C.byref(b)), what)
        return _DMat(a.value, lib), _DMat(b.value, lib)

    def spgemm(X, Y):
        c = C.c_void_p()
        rc = lib.amgh_setup_spgemm(X.h, Y.h, C.byref(c))
        if rc == -5:   # a column of the product outgrew the LDS table: this one product on the host
            return _DMat.upload(X.to_host() @ Y.to_host(), lib, device)
        hip_check(rc, "setup_spgemm")
        return _DMat(c.value, lib)

    def prime_symmetry(M_host, dM):
        """issymmetric(M) and copy(M') from the device while M is there (milliseconds), so that the upload of the
        hierarchy (which asks every level's A for its CSR arrays) does not transpose on the host again."""
        t = C.c_void_p()
        hip_check(lib.amgh_setup_transpose(dM.h, C.byref(t)), "setup_transpose")
        dT = _DMat(t.value, lib)
        same = C.c_int(0)
        hip_check(lib.amgh_dmat_equal(dM.h, dT.h, C.byref(same)), "dmat_equal")
        M_host._sym = bool(same.value)
        if not M_host._sym:
            M_host._T = dT.to_host()
            M_host._T._sym = False
            M_host._T._T = M_host
        return dT

    import os
    import time
    timing = {} if os.environ.get("AMG_SETUP_TIMING") else None
    t_last = [time.perf_counter()]

    def tick(label):
        if timing is not None:
            now = time.perf_counter()
            timing[label] = timing.get(label, 0.0) + now - t_last[0]
            t_last[0] = now

    out = []
    hostA = [A]                 # hostA[l]: the level matrix on the host (from level 1 on: filled by the download thread)
    dA = _DMat.upload(A, lib, device)
    tick("upload A")
    pipe = _Pipeline() if builder is not None else None
    # With a builder, a third thread brings P, R and the next level's A to the host (the host MultiLevel and the schedule
    # builder want them; the next level's strength / splitting / interpolation work on the device copies and on the
    # S, T patterns) and feeds the builder in order; without one everything runs on this thread.
    dl = _Pipeline() if pipe is not None else None

    def run(fn, *args):
        if dl is not None:
            dl.submit(fn, *args)
        else:
            fn(*args)

    def set_symmetry(l, dT, same):
        """issymmetric(A_l) and copy(A_l') as the device found them (see prime_symmetry)."""
        M = hostA[l]
        M._sym = same
        if not same:
            M._T = dT.to_host()
            M._T._sym = False
            M._T._T = M

    # Schedules of up to three levels at a time (AMG_PREPARE_THREADS; amgh_level_prepare touches no handle): level l+1's matrix exists long
    # before the schedules of the much larger level l are done.  The levels join the handle in order (`pipe`).
    pool = None
    prepared = []
    if pipe is not None:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=int(__import__("os").environ.get("AMG_PREPARE_THREADS", "3")))

    def prepare_level(l):
        while len(prepared) <= l:
            prepared.append(None)
        if prepared[l] is None:
            prepared[l] = pool.submit(builder.prepare, hostA[l], *smoothers)

    def begin_level(l):
        prepare_level(l)
        pipe.submit(builder.push_prepared, prepared[l])

    def drop_prepared():
        """After a failure: levels that were prepared and never reached the handle."""
        pool.shutdown(wait=True)
        for fut in prepared:
            if fut is not None and fut.exception() is None and not getattr(fut, "_taken", False):
                builder.free_prepared(fut.result())

    def check_smoothers(M):
        """NoSymmetry smoothers need a nonzero stored diagonal (smoother.jl:239-241): validated level by level as the
        matrices appear, not after the whole hierarchy (and its HBM copy) has been built."""
        if not hermitian and smoothers is not None:
            for s_ in smoothers:
                s_.check_no_symmetry(M)

    def finish_level(l, dP, dR, dRAP, nAT, nsame):
        check_smoothers(hostA[l])
        hostA.append(dRAP.to_host())
        if nAT is not None:
            set_symmetry(l + 1, nAT, nsame)
            # (starting the next level's schedules HERE, ~0.7 s earlier, was measured and is worse: two large levels
            # under construction at once slow each other down by more than the overlap — every synchronous copy and
            # hipFree of one waits for the other's kernels; begin_level starts them when the level is begun)
        out.append((hostA[l], dP.to_host(), dR.to_host()))
        if pipe is not None:
            pipe.submit(builder.push_end, Level(*out[-1], *smoothers))

    def symmetry_of(dM):
        """(copy(M'), issymmetric(M)) on the device."""
        t = C.c_void_p()
        hip_check(lib.amgh_setup_transpose(dM.h, C.byref(t)), "setup_transpose")
        dT = _DMat(t.value, lib)
        same = C.c_int(0)
        hip_check(lib.amgh_dmat_equal(dM.h, dT.h, C.byref(same)), "dmat_equal")
        return dT, bool(same.value)

    lvl, n = 0, A.m
    dAT = None
    try:
        if A._sym is None and max_levels > 1 and n > max_coarse:
            dAT, same = symmetry_of(dA)
            set_symmetry(0, dAT, same)
        check_smoothers(A)
        while lvl + 1 < max_levels and n > max_coarse:
            for q in (dl, pipe):      # a worker thread has failed: stop computing further levels
                if q is not None and q.exc is not None:
                    raise q.exc
            if hermitian:
                dAt = dA
            else:
                if dAT is None:
                    t = C.c_void_p()
                    hip_check(lib.amgh_setup_transpose(dA.h, C.byref(t)), "setup_transpose")
                    dAT = _DMat(t.value, lib)
                dAt = dAT
            tick("transpose / symmetry")
            if pipe is not None:
                run(begin_level, lvl)
            s_, t_, sn, tn = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            hip_check(lib.amgh_setup_classical_strength(dAt.h, theta, C.byref(s_), C.byref(t_), C.byref(sn), C.byref(tn)),
                      "setup_classical_strength")
            dS, dT, dSn, dTn = (_DMat(v.value, lib) for v in (s_, t_, sn, tn))
            tick("strength")
            Sp, Sj, _ = dSn.download(values=False)
            Tp, Tj, _ = dTn.download(values=False)
            tick("download S, T patterns")
            splitting = np.empty(n, dtype=np.int32)
            if L.amgs_rs_cf_splitting_patterns(n, Sp.ctypes.data, Sj.ctypes.data, Tp.ctypes.data, Tj.ctypes.data,
                                               splitting.ctypes.data) != 0:
                raise AMGError(L.amgs_last_error().decode())
            del dS, dSn, dTn
            tick("C/F splitting (host)")
            dR, dP = call2(lib.amgh_setup_direct_interpolation, dAt.h, dT.h, splitting.ctypes.data, what="setup_direct_interpolation")
            if dR.m == 0:        # size(P, 2) == 0: stop coarsening (classical.jl:43)
                if pipe is not None:
                    run(pipe.submit, builder.push_abort)
                break
            tick("interpolation")
            dRAP = spgemm(spgemm(dR, dA), dP)
            tick("R*A*P")
            nAT = nsame = None
            if lvl + 2 < max_levels and dRAP.m > max_coarse:      # the product is the next level: issymmetric / copy(A') now
                nAT, nsame = symmetry_of(dRAP)
            run(finish_level, lvl, dP, dR, dRAP, nAT, nsame)
            tick("download P, R, RAP")
            dA, dAT, n, lvl = dRAP, nAT, dRAP.m, lvl + 1
            del dP, dR, dRAP, nAT, dAt
    except BaseException:
        for q in (dl, pipe):      # never tear the handle down under a running push
            if q is not None:
                q.close(reraise=False)
        if pool is not None:
            drop_prepared()
        raise
    if dl is not None:
        try:
            dl.close()
            tick("wait for the downloads")
            pipe.close()
            tick("wait for the smoother schedules")
        except BaseException:
            pipe.close(reraise=False)
            drop_prepared()
            raise
        pool.shutdown(wait=True)
    A_host = hostA[-1]
    if A_host._sym is None:
        prime_symmetry(A_host, dA)
    if timing is not None:
        import sys
        print("ruge_stuben(setup='gpu') seconds: " + ", ".join(f"{k} {v:.2f}" for k, v in timing.items()),
              file=sys.stderr, flush=True)
    return out, A_host


def _smoothed_aggregation_gpu(A, B, theta, omega, improve_iters, hermitian, max_levels, max_coarse, device=0):
    """extend_hierarchy_sa! (aggregation.jl:116-157) with its two heavy steps on the GPU: the prolongation smoothing
    P = T - (omega D^-1 A) T (amgh_setup_jacobi_prolongation: row sums, scaling, SpGEMM, subtraction) and R*A*P
    (transpose + two SpGEMMs), and SymmetricStrength.  The sequential aggregation, improve_candidates (Gauss-Seidel on the
    candidates) and fit_candidates (a QR per aggregate) run in the host library on the level matrix, which is
    downloaded anyway.  improve_candidates needs only A and B, strength + aggregation only A: the two chains run side
    by side on two host threads (half the OpenMP threads each), and a third thread downloads P and R — nothing on the
    way down reads them — while the next level is under construction.
    Returns [(A, P, R), ...], final_A — bitwise what amgs_smoothed_aggregation builds."""
    from ._libs import hip_check, hip_lib
    lib, L = hip_lib(), setup_lib()
    if lib.amgh_device_count() <= 0:
        raise AMGError("smoothed_aggregation(setup='gpu'): no HIP device visible")

    def spgemm(X, Y):
        c = C.c_void_p()
        rc = lib.amgh_setup_spgemm(X.h, Y.h, C.byref(c))
        if rc == -5:   # a column of the product outgrew the LDS table: this one product on the host
            return _DMat.upload(X.to_host() @ Y.to_host(), lib, device)
        hip_check(rc, "setup_spgemm")
        return _DMat(c.value, lib)

    import os
    import time
    timing = {} if os.environ.get("AMG_SETUP_TIMING") else None
    t_last = [time.perf_counter()]

    def tick(label):
        if timing is not None:
            now = time.perf_counter()
            timing[label] = timing.get(label, 0.0) + now - t_last[0]
            t_last[0] = now

    vector = B is None or B.ndim == 1
    n0 = A.m
    Bcur = np.ones(n0) if B is None else np.array(B, dtype=np.float64, order="F", copy=True)
    out = []
    A_host, dA = A, _DMat.upload(A, lib, device)
    tick("upload A")
    bsr_flag = False
    strength = SymmetricStrength(theta)
    import threading
    threads_all = int(L.amgs_set_threads(0))
    side = _Pipeline()            # improve_candidates beside strength + aggregation
    fetch = _Pipeline()           # downloads of P and R
    fetched = []

    def improve(A_h, Bf, nB, box, share):
        L.amgs_set_threads_here(share)
        if L.amgs_improve_candidates(A_h._h, Bf.ctypes.data, nB, int(improve_iters)) != 0:
            box.append(AMGError(L.amgs_last_error().decode()))

    def fetch_pr(slot, dP, dR):
        fetched[slot] = (dP.to_host(), dR.to_host())

    try:
        while len(out) + 1 < max_levels and A_host.m > max_coarse:
            n = A_host.m
            nB = 1 if vector else Bcur.shape[1]
            Bf = np.asfortranarray(Bcur.reshape(n, nB))
            if Bf is Bcur or np.shares_memory(Bf, Bcur):
                Bf = Bf.copy(order="F")            # the sweeps work in place; a level that ends the loop leaves B alone
            share = max(1, threads_all // 2) if threads_all > 1 and A_host.nnz > 1 << 20 else 0
            failed, done = [], threading.Event()
            if share:
                side.submit(improve, A_host, Bf, nB, failed, share)
                side.submit(done.set)
                L.amgs_set_threads_here(threads_all - share)
            try:    # (whatever strength or aggregation raise, this thread gets its full thread count back)
                # SymmetricStrength on the GPU (strength.jl:77-122, amgh_setup_symmetric_strength: bitwise the host
                # library's); the sequential aggregation reads pattern and values on the host
                if hermitian:
                    dAs = dA
                else:
                    t_ = C.c_void_p()
                    hip_check(lib.amgh_setup_transpose(dA.h, C.byref(t_)), "setup_transpose")
                    dAs = _DMat(t_.value, lib)
                s_ = C.c_void_p()
                hip_check(lib.amgh_setup_symmetric_strength(dAs.h, theta, int(bool(bsr_flag)), C.byref(s_)), "setup_symmetric_strength")
                dS = _DMat(s_.value, lib)
                tick("symmetric strength (GPU)")
                S = dS.to_host()
                del dS, dAs
                tick("download S")
                AggOp = SparseMatrixCSC(L.amgs_standard_aggregation(S._h))
                tick("aggregation (host)")
                if share:
                    while not done.wait(0.05):
                        if side.exc is not None:
                            break
            finally:
                if share:
                    L.amgs_set_threads_here(threads_all)
            if share:
                if side.exc is not None:
                    raise side.exc
            else:
                improve(A_host, Bf, nB, failed, threads_all)
            if failed:
                raise failed[0]
            tick("improve_candidates (host)")
            if AggOp.m == 0:
                break
            if vector:
                # one candidate: the tentative prolongator on the GPU (aggregation.jl:161-193, bitwise the host library's)
                dAgg = _DMat.upload(AggOp, lib, device)
                b1 = np.ascontiguousarray(Bf[:, 0])
                Bc = np.empty(AggOp.m, dtype=np.float64)
                t_ = C.c_void_p()
                hip_check(lib.amgh_setup_fit_candidates_vector(dAgg.h, b1.ctypes.data, 1e-10, C.byref(t_), Bc.ctypes.data),
                          "setup_fit_candidates_vector")
                dT, T = _DMat(t_.value, lib), None
                del dAgg
                tick("fit_candidates (GPU)")
            else:
                T, Bc = fit_candidates(AggOp, Bf)
                tick("fit_candidates (host)")
                dT = _DMat.upload(T, lib, device)
                tick("upload T")
            p = C.c_void_p()
            rc = lib.amgh_setup_jacobi_prolongation(dA.h, dT.h, omega, C.byref(p))
            if rc == -5:
                if T is None:
                    T = dT.to_host()
                dP = _DMat.upload(SparseMatrixCSC(L.amgs_jacobi_prolongation(A_host._h, T._h, omega)), lib, device)
            else:
                hip_check(rc, "setup_jacobi_prolongation")
                dP = _DMat(p.value, lib)
            tick("prolongation smoothing (GPU)")
            if dP.n == 0:
                break
            r = C.c_void_p()
            hip_check(lib.amgh_setup_transpose(dP.h, C.byref(r)), "setup_transpose")
            dR = _DMat(r.value, lib)
            dRAP = spgemm(spgemm(dR, dA), dP)
            tick("R*A*P (GPU)")
            fetched.append(None)
            fetch.submit(fetch_pr, len(fetched) - 1, dP, dR)
            out.append(A_host)
            A_host, dA = dRAP.to_host(), dRAP
            tick("download RAP")
            Bcur = Bc
            bsr_flag = True
    except BaseException:
        side.close(reraise=False)
        fetch.close(reraise=False)
        raise
    side.close()
    fetch.close()
    out = [(a, *pr) for a, pr in zip(out, fetched)]
    tick("wait for the downloads of P, R")
    if timing is not None:
        import sys
        print("smoothed_aggregation(setup='gpu') seconds: " + ", ".join(f"{k} {v:.2f}" for k, v in timing.items()),
              file=sys.stderr, flush=True)
    return out, A_host


def ruge_stuben(A, strength=None, symmetry=None, CF=None, presmoother=None, postsmoother=None,
                max_levels=10, max_coarse=10, coarse_solver=None, setup=None, device=None, **kwargs):
    """ruge_stuben(A; strength=Classical(0.25), symmetry=HermitianSymmetry(), CF=RS(),
    presmoother=GaussSeidel(), postsmoother=GaussSeidel(), max_levels=10, max_coarse=10,
    coarse_solver=QRSolver)   — classical.jl:6-34.

    setup = "host" (libamgsetup, C++/OpenMP) | "gpu" (strength, interpolation and R*A*P on the MI355X, C/F splitting
    on the host; same hierarchy bit for bit).  Default: the environment variable AMG_SETUP, else "host".
    device = a GPU ordinal (with setup="gpu"): the solve-phase hierarchy `ml.device(device)` is built on the way, each
    level's smoother schedules on worker threads while the host does that level's C/F splitting (up to
    AMG_PREPARE_THREADS = 3 levels under construction at once: their transient buffers add up — for a problem close to
    the HBM capacity, e.g. 512^3 memory-lean, build first and call ml.device() afterwards)."""
    if kwargs.get("B") is not None:  # classical.jl:18
        raise AMGError("near null space `B` is only supported for smoothed aggregation AMG, not Ruge-Stüben AMG.")
    strength = strength if strength is not None else Classical(0.25)
    symmetry = symmetry if symmetry is not None else HermitianSymmetry()
    presmoother = presmoother if presmoother is not None else GaussSeidel()
    postsmoother = postsmoother if postsmoother is not None else GaussSeidel()
    coarse_solver = coarse_solver if coarse_solver is not None else _default_coarse_solver
    if not isinstance(strength, Classical) or (CF is not None and not isinstance(CF, RS)):
        raise AMGError("ruge_stuben: only strength=Classical(θ), CF=RS() are built")
    A = SparseMatrixCSC.coerce(A)
    o = amgs_options()
    setup_lib().amgs_default_options_rs(C.byref(o))
    o.theta = strength.theta
    o.max_levels = int(max_levels)
    o.max_coarse = int(max_coarse)
    o.hermitian = int(isinstance(symmetry, HermitianSymmetry))
    import os
    setup = setup if setup is not None else os.environ.get("AMG_SETUP", "host")
    if setup == "gpu":
        builder = None
        if device is not None:
            from .device import DeviceHierarchy
            builder = DeviceHierarchy.incremental(int(device), 1, bool(o.hermitian))
        lv, final_A = _ruge_stuben_gpu(A, strength.theta, int(max_levels), int(max_coarse), bool(o.hermitian),
                                       device=int(device or 0), builder=builder, smoothers=(presmoother, postsmoother))
        levels = [Level(a, p, r, presmoother, postsmoother) for a, p, r in lv]
        for m_ in [final_A] + [l.A for l in levels]:
            m_.eltype = A.eltype
        if not isinstance(symmetry, HermitianSymmetry):
            for lev in levels:
                for s_ in (presmoother, postsmoother):
                    s_.check_no_symmetry(lev.A)
        ml = MultiLevel(levels, final_A, coarse_solver(final_A), presmoother, postsmoother, symmetry, method="rs")
        if builder is not None:
            builder.finish(ml)
            ml._dev[(int(device), 1)] = builder
        return ml
    if setup != "host":
        raise AMGError("ruge_stuben: setup must be 'host' or 'gpu'")
    hier = _Hier(setup_lib().amgs_ruge_stuben(A._h, C.byref(o)))
    return _build_multilevel(hier, presmoother, postsmoother, coarse_solver, symmetry, "rs", A.eltype)


def smoothed_aggregation(A, B=None, symmetry=None, strength=None, aggregate=None, smooth=None,
                         presmoother=None, postsmoother=None, improve_candidates=None, max_levels=10,
                         max_coarse=10, diagonal_dominance=False, keep=False, verbose=False,
                         coarse_solver=None, setup=None, **kwargs):
    """smoothed_aggregation(A; B=nothing, symmetry, strength=SymmetricStrength(),
    aggregate=StandardAggregation(), smooth=JacobiProlongation(4/3), presmoother, postsmoother,
    improve_candidates=GaussSeidel(iter=4), max_levels, max_coarse, coarse_solver) — aggregation.jl:66-114.

    setup = "host" (libamgsetup) | "gpu" (prolongation smoothing and R*A*P on the MI355X, strength / aggregation /
    candidates on the host; same hierarchy bit for bit).  Default: the environment variable AMG_SETUP, else "host"."""
    if np.iscomplexobj(getattr(A, "data", np.zeros(0))):
        raise AMGError("Symmetric strength not implemented for complex matrices.")
    strength = strength if strength is not None else SymmetricStrength()
    symmetry = symmetry if symmetry is not None else HermitianSymmetry()
    smooth = smooth if smooth is not None else JacobiProlongation(4.0 / 3.0)
    presmoother = presmoother if presmoother is not None else GaussSeidel()
    postsmoother = postsmoother if postsmoother is not None else GaussSeidel()
    improve_candidates = improve_candidates if improve_candidates is not None else GaussSeidel(iter=4)
    coarse_solver = coarse_solver if coarse_solver is not None else _default_coarse_solver
    if not isinstance(strength, SymmetricStrength) or not isinstance(smooth, JacobiProlongation):
        raise AMGError("smoothed_aggregation: only SymmetricStrength(θ) / JacobiProlongation(ω) are built")
    if aggregate is not None and not isinstance(aggregate, StandardAggregation):
        raise AMGError("smoothed_aggregation: only StandardAggregation() is built")
    if not isinstance(improve_candidates, GaussSeidel) or improve_candidates.sweep_name != "symmetric":
        raise AMGError("smoothed_aggregation: improve_candidates must be GaussSeidel(SymmetricSweep(), iter)")
    A = SparseMatrixCSC.coerce(A)
    n = A.m
    o = amgs_options()
    setup_lib().amgs_default_options_sa(C.byref(o))
    o.theta = strength.theta
    o.max_levels = int(max_levels)
    o.max_coarse = int(max_coarse)
    o.hermitian = int(isinstance(symmetry, HermitianSymmetry))
    o.sa_omega = smooth.omega
    o.sa_improve_iters = int(improve_candidates.iter)
    Bptr, nB = None, 1
    if B is not None:
        B = np.asarray(B, dtype=np.float64)
        if B.shape[0] != n:  # @assert size(A,1) == size(B,1), aggregation.jl:87
            raise AssertionError("size(A, 1) == size(B, 1)")
        o.sa_B_is_vector = int(B.ndim == 1)
        nB = 1 if B.ndim == 1 else B.shape[1]
        Bf = np.asfortranarray(B.reshape(n, nB))
        Bptr = Bf.ctypes.data
    import os
    setup = setup if setup is not None else os.environ.get("AMG_SETUP", "host")
    if setup == "gpu":
        lv, final_A = _smoothed_aggregation_gpu(A, None if B is None else B, strength.theta, smooth.omega,
                                                int(improve_candidates.iter), bool(o.hermitian), int(max_levels),
                                                int(max_coarse))
        levels = [Level(a, p, r, presmoother, postsmoother) for a, p, r in lv]
        for m_ in [final_A] + [l.A for l in levels]:
            m_.eltype = A.eltype
        if not isinstance(symmetry, HermitianSymmetry):
            for lev in levels:
                for s_ in (presmoother, postsmoother):
                    s_.check_no_symmetry(lev.A)
        ml = MultiLevel(levels, final_A, coarse_solver(final_A), presmoother, postsmoother, symmetry, method="sa")
        if verbose:
            print(ml)
        return ml
    if setup != "host":
        raise AMGError("smoothed_aggregation: setup must be 'host' or 'gpu'")
    hier = _Hier(setup_lib().amgs_smoothed_aggregation(A._h, Bptr, nB, C.byref(o)))
    ml = _build_multilevel(hier, presmoother, postsmoother, coarse_solver, symmetry, "sa", A.eltype)
    if verbose:
        print(ml)
    return ml
