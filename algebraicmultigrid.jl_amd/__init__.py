"""algebraicmultigrid.jl_amd — MI355X-native AMG solve path.

Host mirror of AlgebraicMultigrid.jl's interface for ONE hot path: the cycling
loop (`multilevel.jl`) and the relaxation sweeps (`smoother.jl`), executed by
hand-written HIP kernels for gfx950 behind the C ABI of include/amghip.h.
The directory name contains a dot, so import it through the repo-root alias:

    import amg_amd as AMG
    A  = AMG.poisson((256, 256, 256))
    ml = AMG.ruge_stuben(A)
    x  = AMG._solve(ml, A @ np.ones(A.m))
    p  = AMG.aspreconditioner(ml); x = AMG.cg(A, b, Pl=p)
"""
from ._libs import AMGError, gpu_available, hip_lib, setup_lib  # noqa: F401
from .sparse import SparseMatrixCSC  # noqa: F401
from .smoothers import (BackwardSweep, ForwardSweep, GaussSeidel, Jacobi, SingularException, SOR,  # noqa: F401
                        SymmetricSweep)
from .hierarchy import (Classical, DenseLUFactorization, HermitianSymmetry, JacobiProlongation, Level,  # noqa: F401
                        LinearSolveWrapper, LocalWeighting, SuperLUFactorization,
                        MultiLevel, NoSymmetry, Pinv, QRSolver, RS, StandardAggregation, SymmetricStrength,
                        direct_interpolation, fit_candidates, grid_complexity, operator_complexity, poisson,
                        ruge_stuben, smoothed_aggregation)
from .solve import (AMGSolver, F, Identity, Preconditioner, RugeStubenAMG, RugeStubenPreconBuilder,  # noqa: F401
                    SmoothedAggregationAMG, SmoothedAggregationPreconBuilder, V, W, _solve, _solve_inplace,
                    aspreconditioner, cg, init, solve, solve_)
from .device import DeviceBuffer, DeviceCSR, DeviceHierarchy  # noqa: F401
from . import sharded  # noqa: F401

__all__ = [n for n in dir() if not n.startswith("__")]
