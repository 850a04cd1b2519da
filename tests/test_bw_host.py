"""Wavefront of blocks, host side (no GPU): the plan libamghip builds for single-column fine levels — acyclic block partition
from monotone potentials, launches = depths of the quotient DAG, in-block steps, packed rows with uint16 column offsets,
the quotient as reciprocal + one fma correction (csrc/hip/gs_blocks.hpp) — is executed on the host from the very records the
device kernel reads (`amgh_debug_bw_sweep_host`) and compared BIT FOR BIT with the scalar lexicographic sweep of
smoother.jl:61-90 — the oracle's loop for symmetric operators, and the same loop written out in Python for the rows of a
non-symmetric one (where the reference's NoSymmetry sweep multiplies by a stored inverse diagonal instead of dividing).  The device kernel then only has to do the same arithmetic (tests/test_gpu_abi_surface.py,
tools/block_wave_bench).
The same call validates the plan's quotient graph — what orders the blocks of the one-launch (chained) sweep: every external
position of a block inside a block of its predecessor / successor list, tickets strictly ordered (AMGH_ESTATE otherwise).
"""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import amg_amd as AMG
from conftest import uniform
from oracle import oracle as O


def _bw_sweep(A, x, b, target_rows, backward, omega=1.0, dtype=np.float64):
    lib = AMG.hip_lib("float32" if dtype == np.float32 else "float64")
    rp, ci, va = A.csr_arrays()
    rp = np.ascontiguousarray(rp, dtype=np.int32)
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    va = np.ascontiguousarray(va, dtype=dtype)
    x = np.array(x, dtype=dtype, copy=True)
    b = np.ascontiguousarray(b, dtype=dtype)
    st = np.zeros(4, dtype=np.int64)
    rc = lib.amgh_debug_bw_sweep_host(A.m, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, int(target_rows), int(backward),
                                      float(omega), x.ctypes.data, b.ctypes.data, st.ctypes.data)
    return rc, x, dict(zip(("blocks", "launches", "sum_depth", "ext"), map(int, st)))


def _scalar_sweep(A, x, b, backward, omega=1.0, dtype=np.float64):
    """gs! / sor_step! of smoother.jl:61-90,193-221 on the operator's own rows: z += a_ij x_j in stored order, one rounding
    per operation; rows with a zero diagonal keep their x"""
    rp, ci, va = A.csr_arrays()
    va = np.asarray(va, dtype=dtype)
    x = np.array(x, dtype=dtype, copy=True)
    b = np.asarray(b, dtype=dtype)
    n = A.m
    one, om = dtype(1), dtype(omega)
    for i in (range(n - 1, -1, -1) if backward else range(n)):
        acc, d = dtype(0), dtype(0)
        for j in range(rp[i], rp[i + 1]):
            if ci[j] == i:
                d = va[j]
            else:
                acc = dtype(acc + dtype(va[j] * x[ci[j]]))
        if d != 0:
            x[i] = dtype((b[i] - acc) / d) if omega == 1.0 else dtype(dtype((one - om) * x[i]) + dtype(dtype(om / d) * dtype(b[i] - acc)))
    return x


def _short_rows(n, seed, symmetric, zero_diag=()):
    """random operator with at most 12 off-diagonal entries per row (what the block records take)"""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for i in range(n):
        for j in rng.choice(n, size=int(rng.integers(0, 5)), replace=False):
            if j != i:
                rows.append(i); cols.append(int(j)); vals.append(-rng.random())
    M = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    M.sum_duplicates()
    if symmetric:
        M = M + M.T
    keep = sp.lil_matrix(M.shape)
    M = M.tolil()
    for i in range(n):
        for j, v in list(zip(M.rows[i], M.data[i]))[:12]:
            keep[i, j] = v
    d = np.asarray(abs(keep.tocsr()).sum(axis=1)).ravel() + 1.0
    for r in zero_diag:
        d[r] = 0.0
    K = (keep.tocsr() + sp.diags(d)).tocsc()
    K.eliminate_zeros()
    return AMG.SparseMatrixCSC.from_scipy(K)


CASES = [
    ("poisson 3-D", lambda: AMG.poisson((14, 12, 10)), 64),
    ("poisson 3-D, larger blocks", lambda: AMG.poisson((20, 18, 16)), 512),
    ("poisson 2-D", lambda: AMG.poisson((40, 33)), 96),
    ("poisson 1-D (a chain: one step per row, blocks of at most 64 rows)", lambda: AMG.poisson(300), 32),
    ("random symmetric pattern, zero diagonals", lambda: _short_rows(900, 5, True, zero_diag=(0, 17, 899)), 128),
    ("random non-symmetric pattern", lambda: _short_rows(700, 6, False), 64),
]


@pytest.mark.parametrize("name,make,target", CASES, ids=[c[0] for c in CASES])
def test_block_wave_sweep_is_the_scalar_sweep_bit_for_bit(name, make, target):
    A = make()
    n = A.m
    x0, b = uniform(n, 3) - 0.5, uniform(n, 4)
    for back, s in ((0, AMG.GaussSeidel(AMG.ForwardSweep())), (1, AMG.GaussSeidel(AMG.BackwardSweep()))):
        rc, x, st = _bw_sweep(A, x0, b, target, back)
        assert rc == 0, (name, rc)
        ref = _scalar_sweep(A, x0, b, back)
        assert np.array_equal(x, ref), (name, back, float(np.max(np.abs(x - ref))))
        if A.is_symmetric():
            assert np.array_equal(x, O.smooth(s, A, x0, b, hermitian=True)), (name, back)       # the oracle's gs! loop, bit for bit
        else:
            assert np.allclose(x, O.smooth(s, A, x0, b, hermitian=False), rtol=1e-13, atol=1e-15)   # (NoSymmetry: x = D^-1 (b - ...))
        assert st["blocks"] >= 1 and 1 <= st["launches"] <= st["blocks"] and st["sum_depth"] >= st["launches"]
    # two sweeps in a row (forward then backward) from the result of the first
    rc, x1, _ = _bw_sweep(A, x0, b, target, 0)
    rc, x2, _ = _bw_sweep(A, x1, b, target, 1)
    assert np.array_equal(x2, _scalar_sweep(A, _scalar_sweep(A, x0, b, 0), b, 1))


def test_block_wave_sor_and_float32():
    A = AMG.poisson((12, 11, 9))
    n = A.m
    x0, b = uniform(n, 8) - 0.5, uniform(n, 9)
    for back, sw in ((0, AMG.ForwardSweep()), (1, AMG.BackwardSweep())):
        rc, x, _ = _bw_sweep(A, x0, b, 64, back, omega=1.3)
        assert rc == 0 and np.array_equal(x, _scalar_sweep(A, x0, b, back, omega=1.3))
        rc, x32, _ = _bw_sweep(A, x0, b, 64, back, dtype=np.float32)
        ref32 = _scalar_sweep(A, x0, b, back, dtype=np.float32)
        assert rc == 0 and x32.dtype == np.float32 and np.array_equal(x32, ref32)


def test_block_wave_plan_refuses_long_rows_and_counts_launches():
    ml = AMG.ruge_stuben(AMG.poisson((16, 16, 16)))
    A1 = ml.levels[1].A                                   # 19-point-like rows: up to 18 off-diagonal entries, the longest the records take
    x0, b1 = uniform(A1.m, 31) - 0.5, uniform(A1.m, 32)
    for back in (0, 1):
        rc, x, _ = _bw_sweep(A1, x0, b1, 64, back)
        assert rc == 0 and np.array_equal(x, _scalar_sweep(A1, x0, b1, back))
    A2 = ml.levels[2].A                                   # longer rows still
    assert np.diff(A2.colptr).max() - 1 > 18
    rc, _, _ = _bw_sweep(A2, np.zeros(A2.m), np.ones(A2.m), 64, 0)
    assert rc == -5                                       # AMGH_EUNSUPPORTED
    rc, _, _ = _bw_sweep(AMG.poisson(400), np.zeros(400), np.ones(400), 256, 0)
    assert rc == -5                                       # a chain in blocks of more than 124 rows: more steps than a block's walk takes
    A = AMG.poisson((16, 16, 16))
    rc, _, st = _bw_sweep(A, np.zeros(A.m), np.ones(A.m), 64, 0)
    assert rc == 0 and st["blocks"] == 64 and st["launches"] == 3 * 4 - 2        # 4^3 blocks of 4^3 rows: a wavefront of cubes
    assert st["sum_depth"] == st["launches"] * (3 * 4 - 2) and st["ext"] > 0


def _dict_sweep(A, x, b, target_rows, backward, dtype=np.float64):
    lib = AMG.hip_lib("float32" if dtype == np.float32 else "float64")
    rp, ci, va = A.csr_arrays()
    rp = np.ascontiguousarray(rp, dtype=np.int32)
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    va = np.ascontiguousarray(va, dtype=dtype)
    x = np.array(x, dtype=dtype, copy=True)
    b = np.ascontiguousarray(b, dtype=dtype)
    st = np.zeros(5, dtype=np.int64)
    rc = lib.amgh_debug_bw_dict_sweep_host(A.m, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, int(target_rows), int(backward),
                                           x.ctypes.data, b.ctypes.data, st.ctypes.data)
    return rc, x, dict(zip(("on", "dict_rows", "dict_max", "dict_bytes", "plain_bytes"), map(int, st)))


def test_dictionary_layout_executed_on_the_host_is_the_scalar_sweep_bit_for_bit():
    """The dictionary layout of the dataflow records (csrc/hip/gs_flow.hpp FlowDict) without a GPU: a sweep executed on the
    host from the column records, the dictionary indices in the publish words and the blocks' dictionaries of value rows
    equals the scalar lexicographic sweep (smoother.jl:61-90) bit for bit — a 7-point grid (a handful of distinct value rows
    per block: boundary rows differ from interior ones), a grid with perturbed rows (longer dictionaries), a 2-D grid,
    Float32; an operator whose rows all differ has no such layout (the plain records stay)."""
    cases = [("poisson 3-D", AMG.poisson((14, 12, 10)), 64), ("poisson 2-D", AMG.poisson((40, 30)), 64)]
    S = AMG.poisson((12, 12, 12)).to_scipy().tolil()
    rng = np.random.default_rng(4)
    for i in rng.integers(0, S.shape[0], 200):
        S[i, i] = 6.0 + rng.random()
    cases.append(("perturbed rows", AMG.SparseMatrixCSC.from_scipy(S.tocsc()), 128))
    for name, A, rows in cases:
        x0, b = uniform(A.m, 3) - 0.5, uniform(A.m, 4)
        for dtype in (np.float64, np.float32):
            for backward in (False, True):
                rc, xd, st = _dict_sweep(A, x0, b, rows, backward, dtype)
                assert rc == 0 and st["on"] == 1, (name, st)
                assert st["dict_bytes"] * 2 < st["plain_bytes"] and st["dict_rows"] * 4 <= A.m, (name, st)
                assert np.array_equal(xd, _scalar_sweep(A, x0, b, backward, dtype=dtype)), (name, dtype, backward)
        if name == "perturbed rows":
            assert st["dict_max"] > 8, st
    R = AMG.poisson((10, 10, 10)).to_scipy().tocsr()
    R.data = R.data * (1.0 + 0.01 * np.random.default_rng(5).random(R.data.size))
    Ar = AMG.SparseMatrixCSC.from_scipy(((R + R.T) * 0.5).tocsc())
    x0 = uniform(Ar.m, 6)
    rc, xd, st = _dict_sweep(Ar, x0, uniform(Ar.m, 7), 64, False)
    assert rc == 0 and st["on"] == 0 and np.array_equal(xd, x0), st
