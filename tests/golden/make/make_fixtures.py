#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the DATA fixtures the reference's own tests hold.

Run in the build container only (needs /root/reference and, for the two JLD2
files, libhdf5 from /opt/conda).  The outputs are data — sparse-matrix arrays,
dense vectors and the known-answer vectors quoted in the reference's tests —
never source text.  Everything is stored 0-based, CSC (colptr/rowval/nzval), as
the reference holds its matrices.

    python tests/golden/make/make_fixtures.py

Sources (relative to /root/reference/test):
  thing.jl randlap.jl test.jl ref_S_test.jl ref_R.jl onetoall.jl   one-line CSC literals
  ref_split_test.txt                                               expected C/F vector
  bug.jld2 (G)  lin_elastic_2d.jld2 (A, b, B)                      HDF5
  runtests.jl:154-223                                              46-entry solution vectors
"""
import os
import re
import subprocess
import sys

import numpy as np

REF = "/root/reference/test"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(HERE)


def parse_jl_csc(path):
    txt = open(path).read()
    m, n = map(int, re.search(r"Gm, Gn = (\d+), (\d+)", txt).groups())

    def arr(name, dtype):
        body = re.search(name + r" = \[(.*?)\]", txt, re.S).group(1)
        return np.array([dtype(t) for t in body.replace("\n", " ").split(",") if t.strip()], dtype=dtype)

    colptr = arr("Gcolptr", int).astype(np.int32) - 1
    rowval = arr("Growval", int).astype(np.int32) - 1
    nzval = arr("Gnzval", float).astype(np.float64)
    assert colptr.shape[0] == n + 1 and colptr[-1] == rowval.shape[0] == nzval.shape[0]
    return dict(m=m, n=n, colptr=colptr, rowval=rowval, nzval=nzval)


def build_extractor():
    exe = "/tmp/extract_jld2"
    subprocess.check_call(
        ["gcc", os.path.join(HERE, "extract_jld2.c"), "-I/opt/conda/include", "-L/opt/conda/lib",
         "-Wl,-rpath,/opt/conda/lib", "-lhdf5", "-o", exe])
    return exe


def jld2_sparse(exe, path, name):
    toks = subprocess.check_output([exe, path, "sparse", name]).decode().split("\n")
    m, n = map(int, toks[0].split())
    colptr = np.array(toks[2].split(), dtype=np.int64).astype(np.int32) - 1
    rowval = np.array(toks[4].split(), dtype=np.int64).astype(np.int32) - 1
    nzval = np.array(toks[6].split(), dtype=np.float64)
    return dict(m=m, n=n, colptr=colptr, rowval=rowval, nzval=nzval)


def jld2_dense(exe, path, name):
    toks = subprocess.check_output([exe, path, "dense", name]).decode().split("\n")
    hdr = list(map(int, toks[0].split()))
    dims = hdr[1:]
    vals = np.array(toks[2].split(), dtype=np.float64)
    # HDF5 dims are the reverse of Julia's; memory is Julia column-major.
    return vals.reshape(dims[::-1], order="F") if len(dims) > 1 else vals


def golden_vectors():
    """The five 46-entry known-answer vectors of runtests.jl 'Preconditioning non-SPD problem'."""
    lines = open(os.path.join(REF, "runtests.jl")).read()
    blocks = re.findall(r"diff = x - \[(.*?)\]", lines, re.S)
    assert len(blocks) == 5, len(blocks)
    out = {}
    names = ["solve_Aones_fwd_maxiter1", "solve_b_fwd_maxiter1", "cg_fwd", "cg_sym_reltol1e-6", "solve_b_sym_maxiter1"]
    for nm, b in zip(names, blocks):
        v = np.array([float(t) for t in re.split(r"[,\s;]+", b.strip()) if t], dtype=np.float64)
        assert v.shape[0] == 46
        out[nm] = v
    return out


def main():
    for nm in ["thing", "randlap", "test", "ref_S_test", "ref_R", "onetoall"]:
        np.savez_compressed(os.path.join(OUT, nm + ".npz"), **parse_jl_csc(os.path.join(REF, nm + ".jl")))
    split = np.loadtxt(os.path.join(REF, "ref_split_test.txt")).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "ref_split_test.npz"), splitting=split)
    exe = build_extractor()
    G = jld2_sparse(exe, os.path.join(REF, "bug.jld2"), "G")
    np.savez_compressed(os.path.join(OUT, "bug.npz"), **G)
    A = jld2_sparse(exe, os.path.join(REF, "lin_elastic_2d.jld2"), "A")
    b = jld2_dense(exe, os.path.join(REF, "lin_elastic_2d.jld2"), "b")
    B = jld2_dense(exe, os.path.join(REF, "lin_elastic_2d.jld2"), "B")
    assert A["m"] == 208 and b.shape == (208,) and B.shape == (208, 3), (b.shape, B.shape)
    np.savez_compressed(os.path.join(OUT, "lin_elastic_2d.npz"), b=b, B=B, **A)
    np.savez_compressed(os.path.join(OUT, "thing_solutions.npz"), **golden_vectors())
    print("fixtures written to", OUT)


if __name__ == "__main__":
    sys.exit(main())
