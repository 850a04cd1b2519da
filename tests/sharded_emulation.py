"""Host emulation of the row-sharded cycle on the GLOBAL matrices with the oracle's loops (checker of the -m gpu
tests and of bench_dist.py's parity field; imports nothing but numpy and the oracle)."""
import numpy as np

from oracle import oracle as O


# ---- global host emulation of the row-sharded cycle (full size) ------------------------------------------------------
def emulate_sharded_cycles(ml, b, nranks, lc, cycles, cyc=0):
    """Iterates after 1..cycles cycles (x0 = 0) of the row-sharded cycle as libamghip's amgh_dist_* runs it, emulated
    on the host with the ORACLE's loops on the global matrices — no renumbering, so it runs at 256^3 in seconds:
      * residual / restriction / prolongation: the global oracle products (a shard computes the same rows);
      * Jacobi: the global oracle sweep (exact across shards);
      * Gauss-Seidel / SOR: per directional sweep, every rank p sweeps a copy of the frozen x with a matrix whose
        rows outside [r0_p, r1_p) are EMPTY (a row without a diagonal keeps its x, smoother.jl:87) and keeps its own
        rows of the result: exact lexicographic order inside the shard, halo frozen at the start of the sweep;
      * levels >= lc: the oracle hierarchy of the collapsed levels."""
    import ctypes as C
    import amg_amd as AMG
    L = O.lib()
    sizes = [lv.A.m for lv in ml.levels] + [ml.final_A.m]
    tail = AMG.MultiLevel(ml.levels[lc:], ml.final_A, ml.coarse_solver, ml.presmoother, ml.postsmoother, ml.symmetry,
                          method=ml.method)
    oh_tail = O.OracleHierarchy(tail)
    cuts = [[(p * sizes[l]) // nranks for p in range(nranks + 1)] for l in range(lc)]
    masked = {}

    def masked_colptr(l, p):
        if (l, p) not in masked:
            cp = ml.levels[l].A.colptr
            r0, r1 = cuts[l][p], cuts[l][p + 1]
            m = np.empty_like(cp)
            m[:r0] = cp[r0]
            m[r0:r1 + 1] = cp[r0:r1 + 1]
            m[r1 + 1:] = cp[r1]
            masked[(l, p)] = m
        return masked[(l, p)]

    def sweep(l, kind, direction, omega, x, bb):
        A = ml.levels[l].A
        s = O.orc_smoother_t(kind, direction, 1, 0, float(omega))
        if kind == 2:
            assert L.orc_smooth_arrays(A.m, A.colptr.ctypes.data, A.rowval.ctypes.data, A.nzval.ctypes.data, C.byref(s), 1,
                                       x.ctypes.data, bb.ctypes.data) == 0
            return
        frozen = x.copy()
        for p in range(nranks):
            w = frozen.copy() if nranks > 1 else x
            cp = masked_colptr(l, p)
            assert L.orc_smooth_arrays(A.m, cp.ctypes.data, A.rowval.ctypes.data, A.nzval.ctypes.data, C.byref(s), 1,
                                       w.ctypes.data, bb.ctypes.data) == 0
            r0, r1 = cuts[l][p], cuts[l][p + 1]
            x[r0:r1] = w[r0:r1]

    def smooth(l, sm, x, bb):
        for _ in range(sm.iter):
            if sm.kind == 2:
                sweep(l, 2, 0, sm.omega, x, bb)
            elif sm.kind in (1, 3):
                if sm.sweep_code in (0, 2):
                    sweep(l, sm.kind, 0, sm.omega, x, bb)
                if sm.sweep_code in (1, 2):
                    sweep(l, sm.kind, 1, sm.omega, x, bb)

    def cycle(l, x, bb, c):
        if l == lc:
            xo, _, _ = oh_tail.solve(bb, x0=x, cycle=c, maxiter=1, calculate_residual=False)
            x[:] = xo
            return
        lev = ml.levels[l]
        smooth(l, lev.presmoother, x, bb)
        r = bb - O.spmv(lev.A, x)
        bc = O.spmv(lev.R, r)
        xc = np.zeros(sizes[l + 1])
        cycle(l + 1, xc, bc, c)
        if c == 1:
            cycle(l + 1, xc, bc, 1)
        elif c == 2:
            cycle(l + 1, xc, bc, 0)
        x += O.spmv(lev.P, xc)
        smooth(l, lev.postsmoother, x, bb)

    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b)
    out = []
    for _ in range(cycles):
        cycle(0, x, b, cyc)
        out.append(x.copy())
    return out
