"""Problems of the multi-process IPC tests, built identically by every worker process and by the parent test."""
import numpy as np

import amg_amd as AMG


def uniform(n, seed=0):
    """conftest.uniform (splitmix64 stream) without importing pytest / torch into the worker processes"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def build_case(case):
    """-> (MultiLevel, b, shard_min_rows, [(result key, arguments)])"""
    if case == "jacobi":
        A = AMG.poisson((32, 24, 20))
        jac = AMG.Jacobi(2.0 / 3.0, iter=2)
        ml = AMG.ruge_stuben(A, presmoother=jac, postsmoother=jac)
        plan = [("solve_v", dict(cycle=0, reltol=1e-8, maxiter=60)), ("solve_w", dict(cycle=1, reltol=1e-8, maxiter=60)),
                ("solve_f", dict(cycle=2, reltol=1e-8, maxiter=60)), ("ldiv", {}), ("spmv", {})]
        return ml, uniform(A.m, 5), 500, plan
    if case == "jacobi_overlap":
        # shards with more than 65 536 interior rows: the interior / boundary split runs beside the exchange
        A = AMG.poisson((64, 64, 72))
        jac = AMG.Jacobi(2.0 / 3.0)
        ml = AMG.ruge_stuben(A, presmoother=jac, postsmoother=jac)
        return ml, uniform(A.m, 11), 20000, [("cycles", dict(cycles=2)), ("spmv", {})]
    if case in ("gs", "die"):
        A = AMG.poisson((40, 40, 48))
        ml = AMG.ruge_stuben(A)
        return ml, uniform(A.m, 6), 4000, [("cycles", dict(cycles=3)), ("solve_v", dict(reltol=1e-10, maxiter=60))]
    if case == "c4":
        # BASELINE.json config C4 at FULL size: poisson((256,256,256)), ruge_stuben defaults (every process builds the
        # hierarchy itself, data-parallel half on the GPU)
        A = AMG.poisson((256, 256, 256))
        ml = AMG.ruge_stuben(A, setup="gpu")
        return ml, uniform(A.m, 0), 200_000, [("cycles", dict(cycles=1))]
    if case == "sor_w":
        A = AMG.poisson((32, 32, 32))
        ml = AMG.ruge_stuben(A, presmoother=AMG.SOR(1.2, AMG.ForwardSweep()), postsmoother=AMG.SOR(1.2, AMG.BackwardSweep()))
        return ml, uniform(A.m, 9), 2000, [("cycles", dict(cycles=2, cycle=1))]
    raise ValueError(case)


def assemble(parts, key):
    return np.concatenate([p[key] for p in parts], axis=-1)
