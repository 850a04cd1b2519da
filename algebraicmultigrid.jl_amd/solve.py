"""Solve-phase entry points — host mirror of multilevel.jl:116-264 and
preconditioner.jl:1-24.  Every call goes through libamghip (HIP, gfx950)."""
import numpy as np

from ._libs import AMGError
from .device import CYCLE_F, CYCLE_V, CYCLE_W
from .hierarchy import MultiLevel, ruge_stuben, smoothed_aggregation
from .sparse import SparseMatrixCSC


class Cycle:
    code = None


class V(Cycle):
    code = CYCLE_V


class W(Cycle):
    code = CYCLE_W


class F(Cycle):
    code = CYCLE_F


def _cycle_code(cycle):
    if cycle is None:
        return CYCLE_V
    if isinstance(cycle, type) and issubclass(cycle, Cycle):
        cycle = cycle()
    if not isinstance(cycle, Cycle):
        raise AMGError("cycle must be V(), W() or F()")
    return cycle.code


_SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))


def _arith_dtype(ml, b):
    """The arithmetic type of a solve: Float32 when the hierarchy AND the right-hand side are Float32 (then every
    operation of the reference's generic code is a Float32 one: multilevel.jl:154,166, test/runtests.jl:244-259) —
    the Float32 instance of the library runs it; any mix promotes to Float64 as promote_type does."""
    A0 = ml.levels[0].A if ml.levels else ml.final_A
    return np.dtype(np.float32 if (A0.eltype == np.float32 and np.asarray(b).dtype == np.float32) else np.float64)


def _solve_inplace(x, ml, b, cycle=None, maxiter=100, abstol=0.0, reltol=None, verbose=False, log=False,
                   calculate_residual=True, **kwargs):
    """`_solve!(x, ml, b, cycle; maxiter, abstol, reltol, verbose, log, calculate_residual)`
    (multilevel.jl:158-198).  x is the initial guess and is overwritten.
    reltol defaults to sqrt(eps(real(eltype(b)))) (multilevel.jl:162)."""
    if not isinstance(ml, MultiLevel):
        raise AMGError("ml must be a MultiLevel")
    dt = _arith_dtype(ml, b)
    if reltol is None:
        reltol = float(np.sqrt(np.finfo(np.float32 if np.asarray(b).dtype == np.float32 else np.float64).eps))
    b = np.asarray(b, dtype=dt)
    if b.ndim not in (1, 2):
        raise AMGError("b must be a vector or an n x bs matrix")
    n = ml.levels[0].A.m if ml.levels else ml.final_A.m
    if b.shape[0] != n or x.shape != b.shape:
        raise AMGError("DimensionMismatch: x, b must have length size(A, 1)")
    bs = 1 if b.ndim == 1 else b.shape[1]   # workspace block size (`Val{bs}`, multilevel.jl:28-35)
    xs, hist, iters = ml.device(nrhs=bs, dtype=dt).solve(b, x, _cycle_code(cycle), int(maxiter), float(abstol),
                                                         float(reltol), calculate_residual, log)
    x[...] = xs
    if verbose and calculate_residual:
        for i in range(iters):
            print("Norm of residual at iteration %6d is %.4e" % (i + 1, hist[i]))
    return (x, hist) if log else x


def _solve(ml, b, cycle=None, **kwargs):
    """`_solve(ml, b[, cycle]; kwargs...)`: x = zeros, then `_solve!` (multilevel.jl:152-157).

    Element type: the reference returns `promote_type(eltype(ml.workspace), eltype(b))`
    (multilevel.jl:154; test/runtests.jl:244-259).  Float32 hierarchy + Float32 right-hand side: the whole solve runs
    in Float32 on the library's Float32 instance; any mix: Float64."""
    dt = _arith_dtype(ml, b)
    b_in = np.asarray(b)
    x = np.zeros(b_in.shape, dtype=dt)
    if dt == np.float64 and b_in.dtype == np.float32 and "reltol" not in kwargs:
        kwargs = dict(kwargs, reltol=float(np.sqrt(np.finfo(np.float32).eps)))   # eltype(b) sets the default tolerance
    return _solve_inplace(x, ml, b_in if dt == np.float32 else np.asarray(b_in, dtype=np.float64), cycle, **kwargs)


# ---- CommonSolve-style adapter (multilevel.jl:241-264) ------------------------
class AMGAlg:
    pass


class RugeStubenAMG(AMGAlg):
    pass


class SmoothedAggregationAMG(AMGAlg):
    pass


_SOLVE_KW = ("maxiter", "abstol", "reltol", "verbose", "log", "calculate_residual")
_RS_KW = ("strength", "symmetry", "CF", "presmoother", "postsmoother", "max_levels", "max_coarse", "coarse_solver", "B")
_SA_KW = ("B", "symmetry", "strength", "aggregate", "smooth", "presmoother", "postsmoother", "improve_candidates",
          "max_levels", "max_coarse", "diagonal_dominance", "keep", "verbose", "coarse_solver")


class AMGSolver:
    """`AMGSolver(ml, b)` — what `init` returns and `solve!` consumes (multilevel.jl:241-245)."""

    def __init__(self, ml, b):
        self.ml = ml
        self.b = b


def init(alg, A, b, **kwargs):
    """CommonSolve `init(::RugeStubenAMG | ::SmoothedAggregationAMG, A, b; kwargs...)` (multilevel.jl:256-261):
    the setup phase; returns the AMGSolver that `solve_` (Julia: `solve!`) runs."""
    if isinstance(alg, type):
        alg = alg()
    if isinstance(alg, RugeStubenAMG):
        ml = ruge_stuben(A, **{k: v for k, v in kwargs.items() if k in _RS_KW})
    elif isinstance(alg, SmoothedAggregationAMG):
        ml = smoothed_aggregation(A, **{k: v for k, v in kwargs.items() if k in _SA_KW})
    else:
        raise AMGError("alg must be RugeStubenAMG() or SmoothedAggregationAMG()")
    return AMGSolver(ml, b)


def solve_(solt, cycle=None, **kwargs):
    """CommonSolve `solve!(solt::AMGSolver, args...; kwargs...)` = `_solve(solt.ml, solt.b, ...)` (multilevel.jl:262-264)."""
    if not isinstance(solt, AMGSolver):
        raise AMGError("solve_: expected the AMGSolver returned by init")
    return _solve(solt.ml, solt.b, cycle, **{k: v for k, v in kwargs.items() if k in _SOLVE_KW})


def solve(A, b, alg, cycle=None, **kwargs):
    """solve(A, b, RugeStubenAMG() | SmoothedAggregationAMG(); kwargs...) = init + solve! : the same kwargs go
    to the setup and to `_solve`, each side ignoring what it does not know (multilevel.jl:252-255)."""
    return solve_(init(alg, A, b, **kwargs), cycle, **kwargs)


# ---- LinearSolve `precs` builders (precs.jl:7-38) -------------------------------
class _PreconBuilder:
    """Callable `(builder)(A, p) -> (Pl, Pr)`: a left AMG preconditioner and the identity on the right
    (precs.jl:16-18, 36-38).  `blocksize` is the workspace block size (`Val{bs}`, multilevel.jl:28-35)."""
    _setup = None

    def __init__(self, blocksize=1, **kwargs):
        self.blocksize = int(blocksize)
        self.kwargs = kwargs

    def __call__(self, A, p=None):
        ml = type(self)._setup(SparseMatrixCSC.coerce(A), **self.kwargs)
        if self.blocksize > 1:
            # allocate the n x bs workspace now, as Val{bs} does at setup — in the hierarchy's own eltype (a Float32
            # matrix gets the Float32 instance of the library, as eltype(workspace) follows eltype(A), multilevel.jl:36)
            A0 = ml.levels[0].A if ml.levels else ml.final_A
            f32 = np.dtype(getattr(A0, "eltype", np.float64)).itemsize == 4
            ml.device(nrhs=self.blocksize, dtype=np.float32 if f32 else None)
        return aspreconditioner(ml), Identity()


class Identity:
    """LinearAlgebra.I as a right preconditioner."""

    def ldiv(self, b, x=None):
        if x is None:
            return np.array(b, copy=True)   # LinearAlgebra.I keeps the eltype (a Float32 Krylov vector stays Float32)
        x[...] = b
        return x

    solve = ldiv


class RugeStubenPreconBuilder(_PreconBuilder):
    """RugeStubenPreconBuilder(; blocksize = 1, kwargs...) (precs.jl:27-38)."""
    _setup = staticmethod(ruge_stuben)


class SmoothedAggregationPreconBuilder(_PreconBuilder):
    """SmoothedAggregationPreconBuilder(; blocksize = 1, kwargs...) (precs.jl:7-18)."""
    _setup = staticmethod(smoothed_aggregation)


# ---- preconditioner facade (preconditioner.jl) --------------------------------
class Preconditioner:
    def __init__(self, ml, cycle=None, init="zero"):
        self.ml = ml
        self.cycle = cycle if cycle is not None else V()
        self.init = init

    def ldiv(self, b, x=None):
        """ldiv!(x, p, b): x .= 0 then exactly one cycle without residual (preconditioner.jl:12-19)."""
        dt = _arith_dtype(self.ml, b)
        b = np.asarray(b, dtype=dt)
        if self.init == "zero":
            bs = 1 if b.ndim == 1 else b.shape[1]
            z = self.ml.device(nrhs=bs, dtype=dt).precond_apply(b, _cycle_code(self.cycle))
        else:
            z = b.copy()
            _solve_inplace(z, self.ml, b, self.cycle, maxiter=1, calculate_residual=False)
        if x is None:
            return z
        x[...] = z
        return x

    def solve(self, b):
        """`p \\ b`"""
        return self.ldiv(b)

    def __rmatmul__(self, other):
        raise TypeError("use p.ldiv(b) / p.solve(b)")

    def mul(self, x):
        """mul!(b, p, x) = A₁ x (preconditioner.jl:20), on the GPU."""
        L = len(self.ml.levels)
        return self.ml.device().spmv(0 if L else L, 0, x)


def aspreconditioner(ml, cycle=None):
    return Preconditioner(ml, cycle)


def cg(A, b, Pl=None, abstol=0.0, reltol=_SQRT_EPS, maxiter=None, log=False):
    """IterativeSolvers.jl `cg(A, b; Pl, abstol, reltol, maxiter, log)` as the reference's tests use it
    (cycle_tests.jl:25, runtests.jl:186,204).  Runs entirely on device (amgh_pcg); A must be the
    fine-level operator of Pl's hierarchy."""
    if not isinstance(Pl, Preconditioner):
        raise AMGError("cg: Pl must be aspreconditioner(ml)")
    ml = Pl.ml
    A = SparseMatrixCSC.coerce(A)
    fine = ml.levels[0].A if ml.levels else ml.final_A
    if A is not fine and not (A.shape == fine.shape and A.nnz == fine.nnz and np.array_equal(A.colptr, fine.colptr)
                              and np.array_equal(A.rowval, fine.rowval) and np.array_equal(A.nzval, fine.nzval)):
        raise AMGError("cg: A must be the operator the preconditioner was built from")
    b = np.asarray(b, dtype=np.float64)
    maxiter = A.n if maxiter is None else int(maxiter)
    x, hist, iters = ml.device().pcg(b, _cycle_code(Pl.cycle), True, maxiter, float(abstol), float(reltol))
    if log:
        return x, {"iters": iters, "resnorm": hist[1:], "isconverged": bool(hist[-1] <= max(reltol * hist[0], abstol))}
    return x
