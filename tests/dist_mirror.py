"""TEST-SIDE mirror of the row-sharded V-cycle (BASELINE.json config C4) — an independent emulation in
Python of what libamghip's `amgh_dist_*` does natively; the product is algebraicmultigrid.jl_amd/sharded.py over the
C ABI, this file only serves the tests (gloo world_size-2 run, host emulation of the frozen-halo sweeps).

The reference is single-process; this is the one place the new build adds a real exchange step:
levels with many rows are partitioned by contiguous 1-D row ranges across the ranks (for the
first-axis-fastest 3-D Poisson ordering: z-slabs), every operator application is preceded by an
exchange of the halo entries of its input vector, and the coarse levels are collapsed onto rank 0.

  * One process per GPU; collectives through `torch.distributed` (backend "nccl" = RCCL over xGMI
    on the GPU box, "gloo" in the CPU tests).  Halo entries travel in ONE all-gather per exchange:
    every rank contributes the boundary entries any other rank needs (padded to the largest
    contribution), then gathers its own halo out of the result with a precomputed index.
  * All partitioning / halo bookkeeping is computed redundantly on every rank from the replicated
    host hierarchy (setup is deterministic), so setup needs no communication at all.
  * Local compute goes through an `ops` backend: `HipOps` = libamghip's stand-alone CSR operators
    on the rank's GPU.  (The CPU tests plug in their own backend; the product ships only HipOps.)
  * Jacobi, residual, restriction and prolongation are exactly the single-GPU arithmetic (same
    per-row order).  Gauss-Seidel cannot be both exact and parallel across a 1-D row partition in
    lexicographic order (rank p's first row depends on rank p-1's last row), so on sharded levels
    it is the usual processor-block hybrid: exact lexicographic GS inside a shard, halo values
    frozen at the start of each directional sweep.  Collapsed levels run the exact smoother.
"""
import numpy as np

from amg_amd import AMGError, HermitianSymmetry, MultiLevel

CYCLE_V, CYCLE_W, CYCLE_F = 0, 1, 2


def row_ranges(n, nranks):
    """Contiguous 1-D row-range partition: rank p owns [p*n//N, (p+1)*n//N)."""
    cuts = [(p * n) // nranks for p in range(nranks + 1)]
    return [(cuts[p], cuts[p + 1]) for p in range(nranks)]


class VectorPlan:
    """Halo plan of one distributed vector: which off-range entries this rank reads (union over
    all operators that consume the vector) and which of its own entries other ranks read."""

    def __init__(self, ranges, rank, needs_per_rank):
        self.ranges = ranges
        self.rank = rank
        N = len(ranges)
        self.r0, self.r1 = ranges[rank]
        self.nloc = self.r1 - self.r0
        self.halo_globals = needs_per_rank[rank]              # sorted unique globals outside my range
        self.nhalo = int(self.halo_globals.size)
        starts = np.array([r[0] for r in ranges] + [ranges[-1][1]], dtype=np.int64)
        # send list of every rank: its entries needed by anybody else
        send_lists = []
        for p in range(N):
            lo, hi = ranges[p]
            parts = [nd[(nd >= lo) & (nd < hi)] for q, nd in enumerate(needs_per_rank) if q != p]
            send_lists.append(np.unique(np.concatenate(parts)) if parts else np.zeros(0, dtype=np.int64))
        self.max_send = max(1, max(int(s.size) for s in send_lists))
        self.total_send = sum(int(s.size) for s in send_lists)   # identical on every rank
        self.send_idx = (send_lists[rank] - self.r0).astype(np.int32)   # local indices I contribute
        # where each of my halo entries sits in the all-gathered buffer
        owner = np.searchsorted(starts, self.halo_globals, side="right") - 1
        pos = np.zeros(self.nhalo, dtype=np.int64)
        for p in range(N):
            m = owner == p
            if m.any():
                pos[m] = np.searchsorted(send_lists[p], self.halo_globals[m])
        self.unpack_idx = (owner * self.max_send + pos).astype(np.int32)

    def localize(self, cols):
        """Map global column indices to positions in [local | halo]."""
        cols = np.asarray(cols)
        out = np.empty(cols.shape, dtype=np.int32)
        own = (cols >= self.r0) & (cols < self.r1)
        out[own] = cols[own] - self.r0
        out[~own] = self.nloc + np.searchsorted(self.halo_globals, cols[~own])
        return out


def _needs(rowptr, col, row_ranges_, col_ranges_):
    """For every rank: sorted unique column indices its rows reference outside its own column range."""
    out = []
    for (r0, r1), (c0, c1) in zip(row_ranges_, col_ranges_):
        c = col[rowptr[r0]:rowptr[r1]]
        off = c[(c < c0) | (c >= c1)]
        out.append(np.unique(off).astype(np.int64))
    return out


def _local_block(rowptr, col, val, r0, r1, plan):
    lo, hi = int(rowptr[r0]), int(rowptr[r1])
    rp = (np.asarray(rowptr[r0:r1 + 1], dtype=np.int64) - lo).astype(np.int32)
    return rp, plan.localize(col[lo:hi]), np.ascontiguousarray(val[lo:hi], dtype=np.float64)


class DistMultiLevel:
    """The reference's MultiLevel, row-sharded over `comm.world_size` ranks."""

    def __init__(self, ml, comm, ops, shard_min_rows=200_000):
        if not isinstance(ml, MultiLevel):
            raise AMGError("ml must be a MultiLevel")
        if not isinstance(ml.symmetry, HermitianSymmetry):
            raise AMGError("the sharded path is built for symmetry=HermitianSymmetry()")
        self.ml, self.comm, self.ops = ml, comm, ops
        N, rank = comm.world_size, comm.rank
        self.N, self.rank = N, rank
        L = len(ml.levels)
        sizes = [l.A.m for l in ml.levels] + [ml.final_A.m]
        # levels [0, lc) are sharded; level lc (and below) lives on rank 0
        lc = 0
        while lc < L and sizes[lc] >= shard_min_rows and sizes[lc] >= 8 * N:
            lc += 1
        self.lc = lc
        self.ranges = [row_ranges(sizes[l], N) for l in range(lc)]
        if lc <= L:
            self.ranges.append([(0, sizes[lc])] + [(sizes[lc], sizes[lc])] * (N - 1))
        self.levels = []
        csr = {}
        for l in range(lc):
            lev = ml.levels[l]
            A = lev.A
            csr[l] = dict(A=A.csr_arrays(), S=(A.colptr, A.rowval, A.nzval),
                          P=(lev.R.colptr, lev.R.rowval, lev.R.nzval),   # CSR of P = CSC arrays of R
                          R=(lev.P.colptr, lev.P.rowval, lev.P.nzval), symmetric=A.is_symmetric())
        # halo plans: x_l is read by A_l (and S_l) and by P_{l-1}; res_l is read by R_l
        self.xplan, self.rplan = [], []
        for l in range(lc + 1):
            needs = [np.zeros(0, dtype=np.int64) for _ in range(N)]
            if l < lc:
                for key in ("A",) + (() if csr[l]["symmetric"] else ("S",)):
                    nd = _needs(csr[l][key][0], csr[l][key][1], self.ranges[l], self.ranges[l])
                    needs = [np.union1d(a, b) for a, b in zip(needs, nd)]
            if l >= 1:
                nd = _needs(csr[l - 1]["P"][0], csr[l - 1]["P"][1], self.ranges[l - 1], self.ranges[l])
                needs = [np.union1d(a, b) for a, b in zip(needs, nd)]
            self.xplan.append(VectorPlan(self.ranges[l], rank, needs))
            if l < lc:
                nd = _needs(csr[l]["R"][0], csr[l]["R"][1], self.ranges[l + 1], self.ranges[l])
                self.rplan.append(VectorPlan(self.ranges[l], rank, nd))
        # local operators + vectors
        for l in range(lc):
            xp, rp, xpc = self.xplan[l], self.rplan[l], self.xplan[l + 1]
            r0, r1 = self.ranges[l][rank]
            c0, c1 = self.ranges[l + 1][rank]
            d = dict(n=r1 - r0, nc=c1 - c0, pre=ml.levels[l].presmoother, post=ml.levels[l].postsmoother)
            d["A"] = ops.make_csr(r1 - r0, xp.nloc + xp.nhalo, *_local_block(*csr[l]["A"], r0, r1, xp))
            d["S"] = d["A"] if csr[l]["symmetric"] else ops.make_csr(
                r1 - r0, xp.nloc + xp.nhalo, *_local_block(*csr[l]["S"], r0, r1, xp))
            d["P"] = ops.make_csr(r1 - r0, xpc.nloc + xpc.nhalo, *_local_block(*csr[l]["P"], r0, r1, xpc))
            d["R"] = ops.make_csr(c1 - c0, rp.nloc + rp.nhalo, *_local_block(*csr[l]["R"], c0, c1, rp))
            kinds = {ml.levels[l].presmoother.kind, ml.levels[l].postsmoother.kind}
            ops.prepare(d["S"], jacobi=2 in kinds, gs=bool(kinds & {1, 3}))   # schedules built here, not in the first cycle
            d["res"] = ops.zeros(rp.nloc + rp.nhalo)
            d["tmp"] = ops.zeros(max(r1 - r0, 1))
            self.levels.append(d)
        self.x = [ops.zeros(p.nloc + p.nhalo) for p in self.xplan]
        self.b = [ops.zeros(max(p.nloc, 1)) for p in self.xplan]
        self._xfer = {}
        for name, plans in (("x", self.xplan), ("r", self.rplan)):
            for l, p in enumerate(plans):
                self._xfer[(name, l)] = dict(send_idx=ops.index(p.send_idx), unpack_idx=ops.index(p.unpack_idx),
                                             send=ops.zeros(p.max_send), recv=ops.zeros(p.max_send * N))
        # collapsed levels: an ordinary single-GPU hierarchy on rank 0
        self.coarse = None
        if rank == 0:
            sub = MultiLevel(ml.levels[lc:], ml.final_A, ml.coarse_solver, ml.presmoother, ml.postsmoother,
                             ml.symmetry, method=ml.method)
            self.coarse = ops.make_hierarchy(sub)
        self.n_global = sizes[0]
        self.halo_bytes_per_cycle = 0

    # ---- halo exchange: ONE all-gather per vector ------------------------------------------------
    def exchange(self, name, l, vec):
        plan = (self.xplan if name == "x" else self.rplan)[l]
        if self.N == 1 or plan.total_send == 0:   # same decision on every rank: the all-gather is collective
            return
        t = self._xfer[(name, l)]
        ns = plan.send_idx.size
        if ns:
            self.ops.gather(t["send_idx"], vec, t["send"], ns)
        self.comm.all_gather(t["recv"], t["send"])
        if plan.nhalo:
            self.ops.gather(t["unpack_idx"], t["recv"], self.ops.view(vec, plan.nloc, plan.nhalo), plan.nhalo)

    # ---- smooth!(x, smoother, b) on a sharded level ------------------------------------------------
    def smooth(self, l, s, xzero=False, b_kept=False):
        """xzero: x (local part AND halo) is zero on every rank: the first halo exchange would move zeros.
        b_kept: the previous smooth! call of this cycle swept the same b with Gauss-Seidel / SOR (its level-ordered
        copy inside the operator is still valid).  Returns whether this call leaves such a copy."""
        d, x, b = self.levels[l], self.x[l], self.b[l]
        n = d["n"]
        fresh = xzero   # halo already consistent with the neighbours' x
        for _ in range(s.iter):
            if s.kind == 2:      # Jacobi: exact
                if not fresh:
                    self.exchange("x", l, x)
                fresh = False
                self.ops.jacobi(d["S"], s.omega, x, b, d["tmp"])
                self.ops.copy(self.ops.view(x, 0, n), d["tmp"], n)
            elif s.kind in (1, 3):  # Gauss-Seidel / SOR: exact inside the shard, halo frozen per sweep
                sor = s.kind == 3
                if s.sweep_code in (0, 2):
                    if not fresh:
                        self.exchange("x", l, x)
                    fresh = False
                    self.ops.gs(d["S"], False, s.omega, sor, x, b, b_kept)
                    b_kept = True
                if s.sweep_code in (1, 2):
                    if not fresh:
                        self.exchange("x", l, x)
                    fresh = False
                    self.ops.gs(d["S"], True, s.omega, sor, x, b, b_kept)
                    b_kept = True
        return b_kept and s.kind in (1, 3) and s.iter > 0

    # ---- __solve! (multilevel.jl:214-239) ---------------------------------------------------------
    def cycle(self, l, cyc, xzero=False):
        if l == self.lc:
            if self.rank == 0:
                self.ops.coarse_cycle(self.coarse, self.x[l], self.b[l], cyc)
            return
        d = self.levels[l]
        x, b = self.x[l], self.b[l]
        b_kept = self.smooth(l, d["pre"], xzero)
        self.exchange("x", l, x)
        self.ops.residual(d["A"], x, b, d["res"])
        self.exchange("r", l, d["res"])
        self.ops.spmv(d["R"], d["res"], self.b[l + 1])
        self.ops.zero(self.x[l + 1], self.xplan[l + 1].nloc + self.xplan[l + 1].nhalo)   # coarse_x .= 0, halo included
        self._next(l + 1, cyc)
        self.exchange("x", l + 1, self.x[l + 1])
        self.ops.spmv_add(d["P"], self.x[l + 1], x)
        self.smooth(l, d["post"], False, b_kept)

    def _next(self, l, cyc):  # __solve_next! (multilevel.jl:200-212); x is zero on the first visit only
        self.cycle(l, cyc, True)
        if cyc == CYCLE_W:
            self.cycle(l, CYCLE_W)
        elif cyc == CYCLE_F:
            self.cycle(l, CYCLE_V)

    # ---- entry points -----------------------------------------------------------------------------
    def set_rhs(self, b_local):
        """b_local: this rank's rows of the fine right-hand side (host array)."""
        self.ops.upload(self.b[0], b_local)

    def local_range(self, l=0):
        return self.ranges[l][self.rank]

    def precond_apply(self, cyc=CYCLE_V):
        """ldiv!: x = 0, one cycle (preconditioner.jl:12-19).  Result stays in self.x[0][:nloc]."""
        self.ops.zero(self.x[0], self.xplan[0].nloc + self.xplan[0].nhalo)
        self.cycle(0, cyc, True) if self.lc > 0 else self._collapsed_only(cyc)

    def _collapsed_only(self, cyc):
        if self.rank == 0:
            self.ops.coarse_cycle(self.coarse, self.x[0], self.b[0], cyc)

    def norm_residual(self):
        """||b - A x|| over all ranks (multilevel.jl:188-190)."""
        if self.lc == 0:
            val = self.ops.coarse_resnorm2(self.coarse, self.x[0], self.b[0]) if self.rank == 0 else 0.0
            return float(np.sqrt(self.comm.all_reduce_sum(val)))
        d = self.levels[0]
        self.exchange("x", 0, self.x[0])
        self.ops.residual(d["A"], self.x[0], self.b[0], d["res"])
        return float(np.sqrt(self.comm.all_reduce_sum(self.ops.dot(d["res"], d["res"], d["n"]))))

    def solve(self, b_local, cyc=CYCLE_V, maxiter=100, abstol=0.0, reltol=None):
        """_solve (multilevel.jl:152-198) with x0 = 0; returns (x_local, residual history)."""
        reltol = float(np.sqrt(np.finfo(np.float64).eps)) if reltol is None else reltol
        self.set_rhs(b_local)
        nloc = self.xplan[0].nloc
        self.ops.zero(self.x[0], nloc)
        normb = float(np.sqrt(self.comm.all_reduce_sum(self.ops.dot(self.b[0], self.b[0], nloc))))
        hist = [normb]
        if normb != 0:
            abstol = max(reltol * normb, abstol)
        normres, itr = normb, 1
        while itr <= maxiter and normres > abstol:
            self.cycle(0, cyc) if self.lc > 0 else self._collapsed_only(cyc)
            normres = self.norm_residual()
            hist.append(normres)
            itr += 1
        return self.ops.download(self.x[0], nloc), np.array(hist)


# ---- communicators -----------------------------------------------------------------------------------
class TorchComm:
    """torch.distributed (RCCL on the GPU box, gloo in the CPU tests)."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank()
        self.world_size = dist.get_world_size()

    def all_gather(self, recv, send):
        self.dist.all_gather_into_tensor(recv, send)

    def all_reduce_sum(self, value):
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        return float(t.item())

    def barrier(self):
        self.dist.barrier()


class SingleComm:
    rank, world_size = 0, 1

    def all_gather(self, recv, send):
        recv[:send.numel()] = send

    def all_reduce_sum(self, value):
        return float(value)

    def barrier(self):
        pass


# ---- the product backend: libamghip on this rank's GPU, torch tensors as device memory ----------------
class HipOps:
    def __init__(self, device):
        import ctypes as C

        import torch

        from amg_amd._libs import hip_check
        from amg_amd.device import require_gpu
        self.C, self.torch, self.check = C, torch, hip_check
        self.lib = require_gpu()
        self.device = int(device)
        torch.cuda.set_device(self.device)
        self.dev = torch.device("cuda", self.device)
        self._scratch = torch.zeros(1100, dtype=torch.float64, device=self.dev)
        self._keep = []

    def _stream(self):
        return self.torch.cuda.current_stream(self.dev).cuda_stream

    def zeros(self, n):
        return self.torch.zeros(max(int(n), 1), dtype=self.torch.float64, device=self.dev)

    def index(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        return self.torch.from_numpy(a if a.size else np.zeros(1, dtype=np.int32)).to(self.dev)

    def view(self, v, off, n):
        return v[off:off + n]

    def upload(self, v, host):
        host = np.ascontiguousarray(host, dtype=np.float64)
        v[:host.size].copy_(self.torch.from_numpy(host))

    def download(self, v, n):
        return v[:n].cpu().numpy()

    def zero(self, v, n):
        v[:n].zero_()

    def copy(self, dst, src, n):
        dst[:n].copy_(src[:n])

    def make_csr(self, nrows, ncols, rowptr, col, val):
        from amg_amd.device import DeviceCSR
        return DeviceCSR(nrows, ncols, rowptr, col, val, self.device)

    def prepare(self, op, jacobi, gs):
        if op.nrows:
            self.check(self.lib.amgh_csr_prepare(op.h, int(jacobi), int(gs)), "csr_prepare")

    def make_hierarchy(self, ml):
        from amg_amd.device import DeviceHierarchy
        dev = DeviceHierarchy(ml, self.device)
        self.check(self.lib.amgh_set_stream(dev.h, self._stream()), "set_stream")
        return dev

    def spmv(self, op, x, y):
        if op.nrows:
            self.check(self.lib.amgh_csr_spmv_d(op.h, x.data_ptr(), y.data_ptr(), self._stream()), "spmv")

    def residual(self, op, x, b, r):
        if op.nrows:
            self.check(self.lib.amgh_csr_residual_d(op.h, x.data_ptr(), b.data_ptr(), r.data_ptr(), self._stream()), "residual")

    def spmv_add(self, op, x, y):
        if op.nrows:
            self.check(self.lib.amgh_csr_spmv_add_d(op.h, x.data_ptr(), y.data_ptr(), self._stream()), "spmv_add")

    def jacobi(self, op, omega, xin, b, xout):
        if op.nrows:
            self.check(self.lib.amgh_csr_jacobi_d(op.h, omega, xin.data_ptr(), b.data_ptr(), xout.data_ptr(),
                                                  self._stream()), "jacobi")

    def gs(self, op, backward, omega, sor, x, b, reuse_b=False):
        if op.nrows:
            self.check(self.lib.amgh_csr_gs_ex_d(op.h, int(backward), omega, int(sor), x.data_ptr(), b.data_ptr(),
                                                 self._stream(), 1 if reuse_b else 0), "gs")

    def gather(self, idx, src, dst, n):
        self.check(self.lib.amgh_gather_d(self.device, n, idx.data_ptr(), src.data_ptr(), dst.data_ptr(),
                                          self._stream()), "gather")

    def dot(self, x, y, n):
        out = self.C.c_double(0)
        self.check(self.lib.amgh_dot_d(self.device, n, x.data_ptr(), y.data_ptr(), self._scratch.data_ptr(),
                                       self.C.byref(out), self._stream()), "dot")
        return out.value

    def coarse_cycle(self, dev, x, b, cyc):
        self.check(self.lib.amgh_set_stream(dev.h, self._stream()), "set_stream")
        self.check(self.lib.amgh_cycle_d(dev.h, 0, x.data_ptr(), b.data_ptr(), cyc), "cycle")

    def coarse_resnorm2(self, dev, x, b):
        n = dev.n
        r = self.zeros(n)
        self.check(self.lib.amgh_level_residual_d(dev.h, 0, x.data_ptr(), b.data_ptr(), r.data_ptr()), "residual")
        return self.dot(r, r, n)
