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
