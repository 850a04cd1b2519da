"""Smoother configurations — host mirror of smoother.jl:1-49,92-99,173-180.

`GaussSeidel(sweep, iter)`, `Jacobi(ω; iter)`, `SOR(ω, sweep, iter)` and the
in-place convenience call `(config)(A, x, b, symmetry=HermitianSymmetry())`
(smoother.jl:33-38), which runs the sweep on the GPU through libamghip's
stand-alone CSR operators.
"""
import numpy as np

from ._libs import AMGError, amgh_smoother_t

KIND_NONE, KIND_GS, KIND_JACOBI, KIND_SOR = 0, 1, 2, 3
SWEEP_FORWARD, SWEEP_BACKWARD, SWEEP_SYMMETRIC = 0, 1, 2


class Sweep:
    code = None
    name = None


class ForwardSweep(Sweep):
    code, name = SWEEP_FORWARD, "forward"


class BackwardSweep(Sweep):
    code, name = SWEEP_BACKWARD, "backward"


class SymmetricSweep(Sweep):
    code, name = SWEEP_SYMMETRIC, "symmetric"


def _sweep(s):
    if isinstance(s, type) and issubclass(s, Sweep):
        s = s()
    if not isinstance(s, Sweep):
        raise AMGError("sweep must be ForwardSweep(), BackwardSweep() or SymmetricSweep()")
    return s


class SingularException(ArithmeticError):
    """LinearAlgebra.SingularException(col) — thrown by the NoSymmetry smoothers' setup
    when a diagonal entry is missing or zero (smoother.jl:239-241)."""

    def __init__(self, col):
        super().__init__(f"SingularException({col})")
        self.col = col


class Smoother:
    kind = KIND_NONE
    iter = 1
    omega = 1.0
    sweep_code = SWEEP_SYMMETRIC

    def c_struct(self):
        return amgh_smoother_t(self.kind, self.sweep_code, int(self.iter), 0, float(self.omega))

    def check_no_symmetry(self, A):
        """DiagonalIndices(A) check of the NoSymmetry family (smoother.jl:226-257)."""
        d = A.diagonal()
        stored = np.zeros(A.m, dtype=bool)
        cols = np.repeat(np.arange(A.n, dtype=np.int64), np.diff(A.colptr))
        stored[A.rowval[A.rowval == cols]] = True
        bad = np.nonzero(~stored | (d == 0))[0]
        if bad.size:
            raise SingularException(int(bad[0]) + 1)

    def __call__(self, A, x, b, symmetry=None):
        """In-place `smooth!` on freshly set-up smoother (smoother.jl:33-38). x is updated in place."""
        from .device import smooth_standalone
        smooth_standalone(self, A, x, b, symmetry)
        return None


class GaussSeidel(Smoother):
    """GaussSeidel(; iter=1) = symmetric sweep; GaussSeidel(sweep; iter=1); GaussSeidel(sweep, iter)."""
    kind = KIND_GS

    def __init__(self, sweep=None, iter=1):
        s = _sweep(sweep if sweep is not None else SymmetricSweep())
        self.sweep = s
        self.sweep_code = s.code
        self.sweep_name = s.name
        self.iter = int(iter)

    def __repr__(self):
        return f"GaussSeidel({type(self.sweep).__name__}(), {self.iter})"


class Jacobi(Smoother):
    """Jacobi(ω; iter=1) (smoother.jl:97)."""
    kind = KIND_JACOBI

    def __init__(self, omega=0.5, iter=1):
        self.omega = float(omega)
        self.iter = int(iter)

    def check_no_symmetry(self, A):  # JacobiSmoother skips zero diagonals (smoother.jl:162-168)
        return None

    def __repr__(self):
        return f"Jacobi({self.omega}, iter={self.iter})"


class SOR(Smoother):
    """SOR(ω; iter=1) symmetric; SOR(ω, sweep); SOR(ω, sweep, iter) (smoother.jl:173-180)."""
    kind = KIND_SOR

    def __init__(self, omega, sweep=None, iter=1):
        s = _sweep(sweep if sweep is not None else SymmetricSweep())
        self.omega = float(omega)
        self.sweep = s
        self.sweep_code = s.code
        self.sweep_name = s.name
        self.iter = int(iter)

    def __repr__(self):
        return f"SOR({self.omega}, {type(self.sweep).__name__}(), {self.iter})"
