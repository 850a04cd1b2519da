"""The C-ABI libraries load and export every symbol their headers declare (no GPU needed)."""
import ctypes
import os
import re

import pytest

import amg_amd as AMG
from conftest import ROOT


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"_[a-z0-9_]+)\s*\(", txt)))


def test_amghip_exports_every_declared_symbol():
    lib = AMG.hip_lib()
    names = declared("amghip.h", "amgh")
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n


def test_float32_instance_exports_the_solve_phase_symbols():
    """libamghip_f32.so = the same source with amgh_real = float: every declared entry point — the row-sharded
    hierarchy (amgh_dist_*, amgh_local_group_*) included — except the GPU half of the setup (amgh_dmat_*, amgh_setup_*)."""
    lib = AMG.hip_lib("float32")
    assert lib is not AMG.hip_lib()
    f64_only = ("amgh_dmat_", "amgh_setup_")
    names = [n for n in declared("amghip.h", "amgh") if not n.startswith(f64_only)]
    assert len(names) >= 80 and any(n.startswith("amgh_dist_create_ipc") for n in names)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.amgh_strerror(0) == b"ok" and lib.amgh_device_count() >= 0
    h = ctypes.c_void_p()
    assert lib.amgh_create(ctypes.byref(h), 0, 65) == -5


def test_float32_oracle_is_float32_arithmetic():
    """liboracle_f32.so (real_t = float) against liboracle.so on one symmetric Gauss-Seidel sweep: Float32 result,
    Float32-sized distance from the Float64 one, and exactly what a Float32 scalar loop gives."""
    import numpy as np
    from oracle import oracle as O
    A = AMG.poisson((9, 8))
    n = A.m
    x0 = (np.arange(n) % 7 - 3.0) / 4.0
    b = np.cos(np.arange(n))
    s = AMG.GaussSeidel(AMG.ForwardSweep())
    x64 = O.smooth(s, A, x0, b)
    x32 = O.smooth(s, A, x0.astype(np.float32), b.astype(np.float32), dtype=np.float32)
    assert x32.dtype == np.float32
    d = np.linalg.norm(x32 - x64) / np.linalg.norm(x64)
    assert 1e-9 < d < 1e-5
    # the scalar loop of smoother.jl:61-90 in numpy Float32 scalars (column i read as row i, A symmetric)
    cp, rv, nz = A.colptr, A.rowval, A.nzval.astype(np.float32)
    x = x0.astype(np.float32).copy()
    bf = b.astype(np.float32)
    for i in range(n):
        rsum, d_ = np.float32(0), np.float32(0)
        for j in range(cp[i], cp[i + 1]):
            if rv[j] == i:
                d_ = nz[j]
            else:
                rsum = np.float32(rsum + np.float32(nz[j] * x[rv[j]]))
        if d_ != 0:
            x[i] = np.float32(np.float32(bf[i] - rsum) / d_)
    assert np.array_equal(x, x32)


def test_amgsetup_exports_every_declared_symbol():
    lib = AMG.setup_lib()
    names = declared("amgsetup.h", "amgs")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n


def test_strerror_and_argument_checks_without_gpu():
    lib = AMG.hip_lib()
    assert lib.amgh_strerror(0) == b"ok"
    assert b"invalid argument" in lib.amgh_strerror(-2)
    assert lib.amgh_device_count() >= 0
    assert lib.amgh_create(None, 0, 1) == -2           # NULL out-pointer
    h = ctypes.c_void_p()
    assert lib.amgh_create(ctypes.byref(h), 0, 0) == -5    # block size must be 1..64
    assert lib.amgh_create(ctypes.byref(h), 0, 65) == -5


@pytest.mark.skipif(AMG.gpu_available(), reason="only meaningful on a box without a GPU")
def test_product_path_fails_loudly_without_gpu():
    import numpy as np
    A = AMG.poisson(100)
    ml = AMG.ruge_stuben(A)
    with pytest.raises(AMG.AMGError):
        AMG._solve(ml, np.ones(100))
    with pytest.raises(AMG.AMGError):
        AMG.GaussSeidel()(A, np.ones(100), np.zeros(100))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "algebraicmultigrid.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.lower() or f == "__init__.py" and False, os.path.join(dp, f)
