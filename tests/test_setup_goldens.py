"""Host setup phase (libamgsetup) against the known-answer tests of the reference
(/root/reference/test/runtests.jl, sa_tests.jl, test_regression.jl, nns_test.jl)."""
import numpy as np
import pytest
import scipy.sparse as sp

import amg_amd as AMG
from conftest import load_csc, load_npz


def test_classical_strength_poisson5():  # runtests.jl:22-29
    S, T = AMG.Classical(0.2)(AMG.poisson(5))
    exp = np.array([[1, .5, 0, 0, 0], [.5, 1, .5, 0, 0], [0, .5, 1, .5, 0], [0, 0, .5, 1, .5], [0, 0, 0, .5, 1.]])
    assert np.array_equal(S.toarray(), exp)


def test_classical_strength_graph():  # runtests.jl:30-32
    S, T = AMG.Classical(0.25)(load_csc("test"))
    diff = S.toarray() - load_csc("ref_S_test").toarray()
    assert diff.max() < 1e-10 and np.abs(diff).max() < 1e-10


def test_rs_splitting():  # runtests.jl:39-48
    assert list(AMG.RS()(AMG.poisson(7))) == [0, 1, 0, 1, 0, 1, 0]
    S, T = AMG.Classical(0.25)(load_csc("thing"))
    exp = [0, 0, 1, 0, 1, 1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 0, 1,
           0, 1, 0, 0, 0, 0, 1, 1, 0, 1, 0]
    assert list(AMG.RS()(S)) == exp
    assert np.array_equal(AMG.RS()(load_csc("ref_S_test")), load_npz("ref_split_test")["splitting"])


def test_direct_interpolation():  # runtests.jl:56-64
    A = AMG.poisson(5)
    P, R = AMG.direct_interpolation(A, AMG.poisson(5), [1, 0, 1, 0, 1])
    exp = np.array([[1, 0, 0], [.5, .5, 0], [0, 1, 0], [0, .5, .5], [0, 0, 1.]])
    assert np.array_equal(P.toarray(), exp)
    assert np.array_equal(R.toarray(), exp.T)
    ml = AMG.ruge_stuben(load_csc("thing"))  # runtests.jl:65-67
    assert ml.levels[1].A.m == 19


def test_multilevel_poisson1000():  # runtests.jl:77-88, README.md:38-45 (config C1)
    ml = AMG.ruge_stuben(AMG.poisson(1000))
    assert len(ml) == 8
    assert [l.A.m for l in ml.levels] == [1000, 500, 250, 125, 62, 31, 15]
    assert [l.A.nnz for l in ml.levels] == [2998, 1498, 748, 373, 184, 91, 43]
    assert ml.final_A.m == 7 and ml.final_A.nnz == 19
    assert round(AMG.operator_complexity(ml), 3) == 1.986
    assert round(AMG.grid_complexity(ml), 2) == 1.99


def test_multilevel_randlap():  # runtests.jl:90-102
    ml = AMG.ruge_stuben(load_csc("randlap"))
    assert len(ml) == 3
    assert [l.A.m for l in ml.levels] == [100, 17]
    assert [l.A.nnz for l in ml.levels] == [2066, 289]
    assert ml.final_A.m == 2 and ml.final_A.nnz == 4
    assert round(AMG.operator_complexity(ml), 3) == 1.142
    assert round(AMG.grid_complexity(ml), 3) == 1.190


def test_small_problems_have_no_levels():  # test_regression.jl:41-57 (issue #31)
    for sz in (10, 5, 2):
        for f in (AMG.ruge_stuben, AMG.smoothed_aggregation):
            ml = f(AMG.poisson(sz))
            assert ml.levels == [] and ml.final_A.shape == (sz, sz)
            assert AMG.operator_complexity(ml) == 1 and AMG.grid_complexity(ml) == 1


def test_rs_rejects_near_null_space():  # classical.jl:18
    with pytest.raises(AMG.AMGError):
        AMG.ruge_stuben(AMG.poisson(20), B=np.ones(20))


def symmetric_soc(A, theta):  # the in-test oracle of sa_tests.jl:3-23
    A = sp.csc_matrix(A)
    D = np.abs(A.diagonal())
    C = A.tocoo()
    i, j, v = C.row, C.col, C.data
    mask = (i != j) & (np.abs(v ** 2) >= theta * theta * D[i] * D[j])
    S = sp.csc_matrix((v[mask], (i[mask], j[mask])), shape=A.shape) + sp.diags(D)
    S = sp.csc_matrix(abs(S))
    S.sort_indices()
    Sd = S.toarray()
    for c in range(Sd.shape[1]):  # scale_cols_by_largest_entry!
        m = max(0.0, Sd[:, c].max()) if S[:, c].nnz else 0.0
        if S[:, c].nnz:
            Sd[:, c] = Sd[:, c] / m
    return Sd


def _cases():
    rng = np.random.default_rng(0)
    cases = [sp.csc_matrix(rng.random((s, s))) for s in (2, 3, 5)]
    cases += [AMG.poisson(s).to_scipy() for s in (2, 3, 5, 7, 10, 11, 19)]
    return cases


def test_symmetric_strength_vs_slow_reference():  # sa_tests.jl:26-39
    for M in _cases():
        for theta in (0.0, 0.1, 0.5, 1.0, 10.0):
            S, _ = AMG.SymmetricStrength(theta)(M)
            ref = symmetric_soc(M, theta)
            assert ((ref - S.toarray()) ** 2).sum() < 1e-6


def stand_agg(C):  # sa_tests.jl:64-135 restated with dense lookups (eps = 0)
    C = sp.csc_matrix(C)
    n = C.shape[0]
    Cd = C.toarray()

    def N(i):
        return [j for j in range(n) if abs(Cd[j, i]) > 0]

    def NT(i):
        return [j for j in range(n) if abs(Cd[i, j]) > 0]

    R = {i for i in range(n) if N(i) != [i] or NT(i) != [i]}
    j = 0
    agg = -np.ones(n, dtype=int)
    for i in range(n):
        Ni = set(N(i))
        if Ni <= R:
            R -= Ni
            for x in Ni:
                agg[x] = j
            j += 1
    old_R = set(R)
    for i in range(n):
        if i not in R:
            continue
        best, cand = -np.inf, -1
        for k in range(C.indptr[i], C.indptr[i + 1]):
            x = C.indices[k]
            if x not in old_R and best < C.data[k]:
                best, cand = C.data[k], x
        if cand >= 0:
            agg[i] = agg[cand]
            R.discard(i)
    for i in range(n):
        if i not in R:
            continue
        Ni = (set(N(i)) & R) | {i}
        R -= Ni
        for x in Ni:
            agg[x] = j
        j += 1
    out = np.zeros((agg.max() + 1 if (agg > -1).any() else 0, n))
    for x in range(n):
        if agg[x] > -1:
            out[agg[x], x] = 1.0
    return out


def test_standard_aggregation_vs_slow_reference():  # sa_tests.jl:191-206
    for M in _cases():
        for theta in (0.0, 0.02, 0.1, 1.0):
            Cm = sp.csc_matrix(symmetric_soc(M + M.T, theta))
            calc = AMG.StandardAggregation()(Cm).toarray()
            ref = stand_agg(Cm)
            assert calc.shape == ref.shape and ((calc - ref) ** 2).sum() < 1e-6


def test_standard_aggregation_corner_cases():  # sa_tests.jl:140-188
    S = sp.csc_matrix((np.ones(6), ([0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2])), shape=(4, 4))
    Agg = AMG.StandardAggregation()(S).toarray()
    assert Agg.shape[0] == 2 and (Agg.sum(axis=0) == 1).all()
    S_iso = sp.identity(5, format="csc")
    Agg = AMG.StandardAggregation()(S_iso)
    assert Agg.nnz == 0
    ml = AMG.smoothed_aggregation(sp.diags([2.0] * 20).tocsc())
    assert len(ml) == 1 and ml.final_A.shape == (20, 20)
    A_iso = sp.diags([[-0.5] * 4, [1.0, 1.0, 100.0, 1.0, 1.0], [-0.5] * 4], [-1, 0, 1]).tocsc()
    S5, _ = AMG.SymmetricStrength(0.25)(A_iso)
    Agg = AMG.StandardAggregation()(S5).toarray()
    assert Agg.shape[0] == 2 and Agg[:, 2].sum() == 0


def test_fit_candidates_vector():  # sa_tests.jl:209-268
    def agg(m, n, colptr, rowval):
        return AMG.SparseMatrixCSC.from_arrays(m, n, np.array(colptr) - 1, np.array(rowval) - 1, np.ones(len(rowval)))
    cases = [(agg(2, 5, range(1, 7), [1, 1, 1, 2, 2]), np.ones(5)),
             (agg(2, 5, range(1, 7), [2, 2, 1, 1, 1]), np.ones(5)),
             (agg(3, 9, range(1, 11), [1, 1, 1, 2, 2, 2, 3, 3, 3]), np.ones(9)),
             (agg(3, 9, range(1, 11), [3, 2, 1, 1, 2, 3, 2, 1, 3]), np.arange(1.0, 10.0)),
             (agg(2, 5, [1, 2, 3, 3, 4, 5], [1, 1, 2, 2]), np.array([1, 1, 5, 2, 3.0])),
             (agg(3, 9, [1, 2, 3, 3, 4, 5, 6, 6, 7, 8], [1, 1, 2, 2, 2, 3, 3]), np.arange(1.0, 10.0))]
    for Agg, B in cases:
        B = B.copy()
        B[np.diff(Agg.colptr) == 0] = 0
        Q, Bc = AMG.fit_candidates(Agg, B)
        Qd = Q.toarray()
        assert np.allclose(B, Qd @ Bc) and np.allclose(Qd @ (Qd.T @ B), B)


def test_fit_candidates_matrix():  # nns_test.jl:28-107
    def aggT(rows, cols, m, n):
        return sp.csc_matrix((np.ones(len(rows)), (np.array(rows) - 1, np.array(cols) - 1)), shape=(m, n))
    r9 = list(range(1, 10))
    cases = [(aggT([1, 2, 3, 4, 5], [1, 1, 1, 2, 2], 5, 2), np.ones((5, 1))),
             (aggT(r9, [3, 2, 1, 1, 2, 3, 2, 1, 3], 9, 3), np.arange(9.0).reshape(9, 1)),
             (aggT([1, 2, 3, 4], [1, 1, 2, 2], 4, 2), np.c_[np.ones(4), np.arange(4.0)]),
             (aggT(r9, [1, 1, 1, 2, 2, 2, 3, 3, 3], 9, 3), np.c_[np.ones(9), np.arange(9.0)]),
             (aggT(r9, [1, 1, 2, 2, 3, 3, 4, 4, 4], 9, 4), np.c_[np.ones(9), np.arange(9.0)]),
             (aggT([1, 2, 3, 4], [1, 1, 2, 2], 4, 2), np.c_[np.ones(4), 1e-20 * np.arange(4.0)]),
             (aggT([1, 2, 4, 5], [1, 1, 2, 2], 5, 2), np.c_[np.ones(5), np.arange(1.0, 6.0)]),
             (aggT([1, 2, 4, 5], [1, 1, 2, 2], 5, 2), np.c_[np.ones(5), np.arange(1.0, 6.0), np.arange(5.0, 0, -1)]),
             (aggT([2, 3, 4, 5, 6], [1, 1, 2, 2, 2], 7, 2), np.c_[np.ones(7), np.arange(1.0, 8.0)])]
    for AT, fine in cases:
        fine = fine.copy()
        fine[np.asarray(AT.sum(axis=1)).ravel() == 0, :] = 0.0
        Q, R = AMG.fit_candidates(sp.csc_matrix(AT.T), fine)
        Qd = Q.toarray()
        assert np.allclose(fine, Qd @ R) and np.allclose(fine, Qd @ (Qd.T @ fine))


def test_b_as_vector_equals_b_as_matrix():  # nns_test.jl:6-24 (setup part)
    A = AMG.poisson(100)
    m1 = AMG.smoothed_aggregation(A)
    m2 = AMG.smoothed_aggregation(A, B=np.ones(100))
    m3 = AMG.smoothed_aggregation(A, B=np.ones((100, 1)))
    # the QR method may flip the sign of a column of Q (LAPACK Householder convention); the
    # hierarchy is the same up to that sign, the solutions are equal (checked in the oracle tests)
    for a, b in ((m1, m2), (m1, m3)):
        assert len(a) == len(b)
        for la, lb in zip(a.levels, b.levels):
            assert np.allclose(np.abs(la.P.toarray()), np.abs(lb.P.toarray()), atol=1e-12)
            assert np.allclose(np.abs(la.A.toarray()), np.abs(lb.A.toarray()), atol=1e-12)


def test_jacobi_prolongation():  # sa_tests.jl:382-388
    x = AMG.JacobiProlongation(4 / 3)(AMG.poisson(100), AMG.poisson(100), 1, 1)
    assert ((x.toarray() - load_csc("ref_R").toarray()) ** 2).sum() < 1e-6


def test_issue24_nodes_not_aggregated():  # sa_tests.jl:390-396, test_regression.jl:7-12
    ml = AMG.smoothed_aggregation(load_csc("onetoall"))
    assert ml.levels[1].A.shape == (11, 11) and ml.final_A.shape == (2, 2)


def test_lin_elastic_hierarchy():  # nns_test.jl:213-234 (setup part)
    d = load_npz("lin_elastic_2d")
    A = load_csc("lin_elastic_2d")
    ml = AMG.smoothed_aggregation(A, B=d["B"])
    assert [l.A.m for l in ml.levels] == [208, 39] and ml.final_A.m == 3
    Agg = AMG.StandardAggregation()(A)
    Q, R = AMG.fit_candidates(Agg, d["B"])
    Qd = Q.toarray()
    assert np.allclose(d["B"], Qd @ R) and np.allclose(d["B"], Qd @ (Qd.T @ d["B"]))


def test_poisson_gallery():  # gallery.jl:1-63
    A = AMG.poisson((3, 4)).toarray()
    n = 12
    ref = np.zeros((n, n))
    for j in range(4):
        for i in range(3):
            r = i + 3 * j
            ref[r, r] = 4
            if i > 0: ref[r, r - 1] = -1
            if i < 2: ref[r, r + 1] = -1
            if j > 0: ref[r, r - 3] = -1
            if j < 3: ref[r, r + 3] = -1
    assert np.array_equal(A, ref)
    A3 = AMG.poisson((8, 8, 8))
    assert A3.m == 512 and A3.nnz == 7 * 512 - 6 * 64
    assert AMG.poisson((256, 1)).nnz == 3 * 256 - 2


def test_improve_candidates_level_parallel_sweeps_are_bitwise_the_sequential_ones(monkeypatch):
    """improve_candidates (aggregation.jl:135-136: symmetric Gauss-Seidel on the candidates, the Hermitian fast path of
    smoother.jl:61-90) runs its sweeps level-scheduled on all host threads for big operators: rows of one dependency
    level in parallel, each with the scalar loop's arithmetic — bit for bit the sequential sweeps."""
    import numpy as np
    from amg_amd._libs import setup_lib
    L = setup_lib()
    A = AMG.poisson((60, 50, 40))        # 120 000 rows: above the threshold of the parallel path
    n = A.m
    out = {}
    for seq in ("1", None):
        if seq:
            monkeypatch.setenv("AMGS_SEQUENTIAL_GS", seq)
        else:
            monkeypatch.delenv("AMGS_SEQUENTIAL_GS", raising=False)
        B = np.asfortranarray(np.stack([np.ones(n) + 1e-3 * np.cos(np.arange(n)), np.linspace(-1.0, 1.0, n)], axis=1))
        assert L.amgs_improve_candidates(A._h, B.ctypes.data, 2, 3) == 0
        out[seq] = B.copy()
    assert np.array_equal(out["1"], out[None])
