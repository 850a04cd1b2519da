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
    dev = DeviceHierarchy(ml, 0, bs)
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
