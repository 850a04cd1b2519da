"""Test-side backends for the Python mirror of the row-sharded driver (tests/dist_mirror.py).

CpuOps   : local operators on CPU torch tensors, arithmetic by the ORACLE (tests only) — lets the
           partition / halo / collapse logic run under gloo with world_size 2 on a box without GPU.
ThreadComm: N virtual ranks as threads in ONE process sharing one GPU (gpurun gives a single GPU):
           the all-gather is a barrier + device-to-device copies.  Used by the -m gpu tests to drive
           the real HipOps backend with N = 2, 4 shards.
"""
import threading

import numpy as np
import torch

from oracle import oracle as O


class _Csr:
    def __init__(self, nrows, ncols, rowptr, col, val):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        # the oracle works on CSC arrays: CSR of M == CSC of M'  (m = ncols, n = nrows)
        self.m, self.n = self.ncols, self.nrows
        self.colptr, self.rowval, self.nzval = self.rowptr, self.col, self.val


class CpuOps:
    """Reference arithmetic on CPU tensors: y = M x with M given by CSR rows == oracle's A' product."""

    def zeros(self, n):
        return torch.zeros(max(int(n), 1), dtype=torch.float64)

    def index(self, a):
        a = np.ascontiguousarray(a, dtype=np.int64)
        return torch.from_numpy(a if a.size else np.zeros(1, dtype=np.int64))

    def view(self, v, off, n):
        return v[off:off + n]

    def upload(self, v, host):
        v[:len(host)] = torch.from_numpy(np.ascontiguousarray(host, dtype=np.float64))

    def download(self, v, n):
        return v[:n].numpy().copy()

    def zero(self, v, n):
        v[:n] = 0

    def copy(self, dst, src, n):
        dst[:n] = src[:n]

    def make_csr(self, nrows, ncols, rowptr, col, val):
        return _Csr(nrows, ncols, rowptr, col, val)

    def prepare(self, op, jacobi, gs):
        pass

    def make_hierarchy(self, ml):
        return O.OracleHierarchy(ml)

    def _mul(self, op, x):
        return O.spmv(op, x.numpy()[:op.ncols], adjoint=True) if op.nrows else np.zeros(0)

    def spmv(self, op, x, y):
        y[:op.nrows] = torch.from_numpy(self._mul(op, x))

    def residual(self, op, x, b, r):
        r[:op.nrows] = b[:op.nrows] - torch.from_numpy(self._mul(op, x))

    def spmv_add(self, op, x, y):
        y[:op.nrows] = y[:op.nrows] + torch.from_numpy(self._mul(op, x))

    def jacobi(self, op, omega, xin, b, xout):
        n = op.nrows
        xi, bb = xin.numpy(), b.numpy()
        out = xi[:n].copy()
        for i in range(n):
            rs, d = 0.0, 0.0
            for j in range(op.rowptr[i], op.rowptr[i + 1]):
                if op.col[j] == i:
                    d = op.val[j]
                else:
                    rs += op.val[j] * xi[op.col[j]]
            if d != 0:
                out[i] = (1.0 - omega) * xi[i] + omega * ((bb[i] - rs) / d)
        xout[:n] = torch.from_numpy(out)

    def gs(self, op, backward, omega, sor, x, b, reuse_b=False):
        n = op.nrows
        xv, bb = x.numpy(), b.numpy()
        order = range(n - 1, -1, -1) if backward else range(n)
        for i in order:
            rs, d = 0.0, 0.0
            for j in range(op.rowptr[i], op.rowptr[i + 1]):
                if op.col[j] == i:
                    d = op.val[j]
                else:
                    rs += op.val[j] * xv[op.col[j]]
            if d != 0:
                xv[i] = (1 - omega) * xv[i] + (omega / d) * (bb[i] - rs) if sor else (bb[i] - rs) / d

    def gather(self, idx, src, dst, n):
        dst[:n] = src[idx[:n]]

    def dot(self, x, y, n):
        return float(x[:n] @ y[:n])

    def coarse_cycle(self, oh, x, b, cyc):
        n = oh.n
        xo, _, _ = oh.solve(b.numpy()[:n], x0=x.numpy()[:n], cycle=cyc, maxiter=1, calculate_residual=False)
        x[:n] = torch.from_numpy(xo)

    def coarse_resnorm2(self, oh, x, b):
        A = oh.ml.levels[0].A if oh.ml.levels else oh.ml.final_A
        r = b.numpy()[:oh.n] - O.spmv(A, x.numpy()[:oh.n])
        return float(r @ r)


class OracleOps(CpuOps):
    """CpuOps with the sweeps done by the oracle's C loops (orc_smooth_arrays) instead of Python loops, so that the
    host emulation of the sharded cycle also runs at full size.  A shard's block is n_loc x (n_loc + n_halo): it is
    padded to a square matrix whose halo rows are EMPTY — a row without a diagonal keeps its x (smoother.jl:87), which
    is exactly "halo frozen during the sweep"."""

    def _square(self, op):
        if not hasattr(op, "_sq"):
            n = op.ncols
            rp = np.concatenate([op.rowptr, np.full(n - op.nrows, op.rowptr[-1], dtype=np.int32)])
            op._sq = (n, np.ascontiguousarray(rp, dtype=np.int32), op.col, op.val)
        return op._sq

    def _sweep(self, op, kind, sweep, omega, x, b):
        import ctypes as C
        n, rp, ci, va = self._square(op)
        s = O.orc_smoother_t(kind, sweep, 1, 0, float(omega))
        xv = x.numpy()
        bb = np.zeros(n)
        bb[:op.nrows] = b.numpy()[:op.nrows]
        assert xv.size >= n
        rc = O.lib().orc_smooth_arrays(n, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, C.byref(s), 1,
                                       xv.ctypes.data, bb.ctypes.data)
        assert rc == 0

    def jacobi(self, op, omega, xin, b, xout):
        if not op.nrows:
            return
        tmp = xin.clone()
        self._sweep(op, 2, 0, omega, tmp, b)
        xout[:op.nrows] = tmp[:op.nrows]

    def gs(self, op, backward, omega, sor, x, b, reuse_b=False):
        if op.nrows:
            self._sweep(op, 3 if sor else 1, 1 if backward else 0, omega, x, b)


class ThreadComm:
    """One of N virtual ranks living in threads of one process (single GPU)."""

    class Shared:
        def __init__(self, n):
            self.n = n
            self.barrier = threading.Barrier(n)
            self.slots = [None] * n
            self.vals = [0.0] * n
            # the virtual ranks only emulate SPMD control flow: exactly one of them runs between
            # two collectives (no concurrent calls into HIP / torch from several Python threads)
            self.turn = threading.Lock()

        def wait(self):
            self.turn.release()
            try:
                self.barrier.wait()
            finally:
                self.turn.acquire()

    def __init__(self, shared, rank):
        self.sh, self.rank, self.world_size = shared, rank, shared.n

    def all_gather(self, recv, send):
        if send.is_cuda:
            torch.cuda.synchronize()
        self.sh.slots[self.rank] = send
        self.sh.wait()
        m = send.numel()
        for p in range(self.world_size):
            recv[p * m:(p + 1) * m].copy_(self.sh.slots[p])
        if send.is_cuda:
            torch.cuda.synchronize()
        self.sh.wait()

    def all_reduce_sum(self, value):
        self.sh.vals[self.rank] = float(value)
        self.sh.wait()
        s = sum(self.sh.vals)
        self.sh.wait()
        return s

    def barrier(self):
        self.sh.wait()


def run_virtual_ranks(n, fn):
    """Run fn(comm) on n threads; returns the list of results (exceptions re-raised)."""
    shared = ThreadComm.Shared(n)
    out, err = [None] * n, [None] * n

    def work(r):
        shared.turn.acquire()
        try:
            out[r] = fn(ThreadComm(shared, r))
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            shared.barrier.abort()
        finally:
            shared.turn.release()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out


from sharded_emulation import emulate_sharded_cycles  # noqa: E402,F401  (kept importable from here)
