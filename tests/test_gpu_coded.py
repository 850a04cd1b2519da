"""Value-coded columns (include/amghip.h amgh_debug_coded_ops): the SpMV-type launches of the level-ordered cycle — residual
(multilevel.jl:219-220), restriction (:221), prolongation (:233-234) — stream one 32-bit word per entry (column | code << 24)
where an operator has at most 256 distinct values; the products and their order are the plain kernel's, so the cycle is the
same bit for bit, and the oracle's at 1e-10."""
import numpy as np
import pytest

import amg_amd as AMG
from amg_amd import DeviceHierarchy
from bench import uniform
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def rel(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y)) / max(np.linalg.norm(y), 1e-300)


def _set(lib, name, v):
    assert lib.amgh_debug_set_tunable(name, v) == 0


@pytest.mark.parametrize("bs", [1, 3, 8])
def test_coded_columns_cycle_is_the_plain_cycle_bit_for_bit(bs):
    lib = AMG.hip_lib()
    A = AMG.poisson((64, 64, 66))                       # 270 336 rows: the big-operator configuration of the stream kernel
    ml = AMG.ruge_stuben(A)
    n = A.m
    B = np.stack([uniform(n, 70 + c) - 0.1 * c for c in range(bs)], axis=1)
    b = B[:, 0].copy() if bs == 1 else B
    _set(lib, b"trim_coded", 0)                          # (both copies kept: this test switches between them at run time)
    try:
        dev = DeviceHierarchy(ml, 0, bs)
    finally:
        _set(lib, b"trim_coded", 1)
    assert lib.amgh_debug_coded_ops(dev.h, 0) == 5       # A (2 values) and P (2 weights); R has 135 168 rows: below the size that pays
    assert lib.amgh_debug_coded_ops(dev.h, len(ml.levels) - 1) == 0   # a small level: plain columns
    z = dev.precond_apply(b)
    try:
        _set(lib, b"stream_code", 0)
        assert lib.amgh_debug_coded_ops(dev.h, 0) == 0
        z_plain = dev.precond_apply(b)
        dev_plain = DeviceHierarchy(ml, 0, bs)           # ... and a hierarchy that never built them
        z_never = dev_plain.precond_apply(b)
    finally:
        _set(lib, b"stream_code", 1)
    assert np.array_equal(z, z_plain) and np.array_equal(z, z_never)
    assert lib.amgh_debug_coded_ops(dev_plain.h, 0) == 0
    oh = O.OracleHierarchy(ml)
    zc = z if bs == 1 else z[:, bs - 1]
    assert rel(zc, oh.precond(b if bs == 1 else B[:, bs - 1])) <= 1e-10
    x, _, its = dev.solve(b, np.zeros_like(b), 0, 4, 0.0, 0.0, False, False)
    try:
        _set(lib, b"stream_code", 0)
        x_plain = dev.solve(b, np.zeros_like(b), 0, 4, 0.0, 0.0, False, False)[0]
    finally:
        _set(lib, b"stream_code", 1)
    assert np.array_equal(x, x_plain)


def test_operators_of_many_values_keep_their_plain_columns():
    """Every entry its own value (a symmetric perturbation of the grid): more than 256 distinct values — nothing is coded, the
    cycle is the oracle's; -0.0 and 0.0, which compare equal, are two codes (values are told apart by their bits)."""
    lib = AMG.hip_lib()
    S = AMG.poisson((64, 64, 66)).to_scipy().tocsr()
    rng = np.random.default_rng(9)
    R = S.copy()
    R.data = R.data * (1.0 + 0.01 * rng.random(R.data.size))
    R = ((R + R.T) * 0.5).tocsc()
    A = AMG.SparseMatrixCSC.from_scipy(R)
    ml = AMG.ruge_stuben(A)
    dev = DeviceHierarchy(ml, 0, 1)
    assert lib.amgh_debug_coded_ops(dev.h, 0) & 1 == 0
    b = uniform(A.m, 5)
    assert rel(dev.precond_apply(b), O.OracleHierarchy(ml).precond(b)) <= 1e-10


def test_trimmed_footprint_keeps_only_the_coded_columns():
    """The default footprint: an operator of the level-ordered cycle that has value-coded columns keeps ONLY them (amgh_finalize
    releases its 12-byte columns and values; the level-ordered A only where the level sweeps the block layout as a dataflow).
    Same cycle bit for bit as with both copies; the run-time tunable stream_code = 0 has nothing to switch to on such an
    operator; the stand-alone hooks (amgh_level_spmv) read the coded columns; fewer bytes on the device."""
    lib = AMG.hip_lib()
    A = AMG.poisson((64, 64, 66))
    ml = AMG.ruge_stuben(A)
    n = A.m
    b = uniform(n, 91) - 0.3
    _set(lib, b"gs_bw", 2)                               # (the fine level on the block layout, as at full size: its A is trimmed too)
    try:
        dev = DeviceHierarchy(ml, 0, 1)
        _set(lib, b"trim_coded", 0)
        try:
            both = DeviceHierarchy(ml, 0, 1)
        finally:
            _set(lib, b"trim_coded", 1)
    finally:
        _set(lib, b"gs_bw", 1)
    assert lib.amgh_debug_coded_ops(dev.h, 0) == 5 and lib.amgh_debug_bw_mode(dev.h, 0) == 3
    assert dev.device_bytes() < both.device_bytes() - 12 * (A.nnz + ml.levels[0].P.nnz) * 0.9
    z = dev.precond_apply(b)
    assert np.array_equal(z, both.precond_apply(b))
    try:
        _set(lib, b"stream_code", 0)
        assert lib.amgh_debug_coded_ops(dev.h, 0) == 5 and lib.amgh_debug_coded_ops(both.h, 0) == 0
        assert np.array_equal(dev.precond_apply(b), z) and np.array_equal(both.precond_apply(b), z)
    finally:
        _set(lib, b"stream_code", 1)
    assert rel(z, O.OracleHierarchy(ml).precond(b)) <= 1e-10
    x = uniform(n, 92)
    for which in (0, 1, 2):                               # A, P, R through the hooks
        y = dev.spmv(0, which, x[: (ml.levels[0].P.n if which == 1 else n)])
        assert np.array_equal(y, both.spmv(0, which, x[: (ml.levels[0].P.n if which == 1 else n)]))
