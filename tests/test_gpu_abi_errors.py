"""Error behaviour of the C ABI on a real device: every misuse returns a negative code, nothing
throws or crashes (include/amghip.h conventions)."""
import ctypes as C

import numpy as np
import pytest

import amg_amd as AMG
from amg_amd._libs import amgh_smoother_t

pytestmark = pytest.mark.gpu

EINVAL, ESTATE, EUNSUPPORTED = -2, -3, -5


def _csr(A):
    rp, ci, va = A.csr_arrays()
    return rp.ctypes.data, ci.ctypes.data, va.ctypes.data


def test_state_machine_and_argument_checks():
    lib = AMG.hip_lib()
    A = AMG.poisson(40)
    ml = AMG.ruge_stuben(A, max_levels=2)
    lev = ml.levels[0]
    h = C.c_void_p()
    assert lib.amgh_create(C.byref(h), 99, 1) == EINVAL                     # no such device
    assert lib.amgh_create(C.byref(h), 0, 1) == 0
    gs = amgh_smoother_t(1, 2, 1, 0, 1.0)
    bad = amgh_smoother_t(7, 0, 1, 0, 1.0)
    a, p, r = _csr(lev.A), (lev.R.colptr.ctypes.data, lev.R.rowval.ctypes.data, lev.R.nzval.ctypes.data), \
        (lev.P.colptr.ctypes.data, lev.P.rowval.ctypes.data, lev.P.nzval.ctypes.data)
    n, nc = lev.A.m, lev.P.n
    x = np.zeros(n); b = np.ones(n); it = C.c_int(0)
    # not finalized yet
    assert lib.amgh_solve(h, b.ctypes.data, x.ctypes.data, 0, 10, 0.0, 1e-8, 1, None, C.byref(it)) == ESTATE
    assert lib.amgh_finalize(h) == ESTATE                                    # no coarse level set
    assert lib.amgh_push_level(h, n, nc, *a, None, None, None, *p, *r, C.byref(bad), C.byref(gs)) == EINVAL
    assert lib.amgh_push_level(h, n, nc, None, None, None, None, None, None, *p, *r, C.byref(gs), C.byref(gs)) == EINVAL
    assert lib.amgh_push_level(h, n, nc, *a, None, None, None, *p, *r, C.byref(gs), C.byref(gs)) == 0
    # next level must have n == previous nc
    assert lib.amgh_push_level(h, n, nc, *a, None, None, None, *p, *r, C.byref(gs), C.byref(gs)) == EINVAL
    fa = _csr(ml.final_A)
    op = np.asfortranarray(ml.coarse_solver.dense_operator())
    assert lib.amgh_set_coarse(h, nc + 1, *fa, op.ctypes.data) == EINVAL     # size mismatch with the last level
    assert lib.amgh_set_coarse(h, nc, *fa, None) == EINVAL
    assert lib.amgh_set_coarse(h, nc, *fa, op.ctypes.data) == 0
    assert lib.amgh_set_coarse(h, nc, *fa, op.ctypes.data) == ESTATE         # already set
    assert lib.amgh_finalize(h) == 0
    assert lib.amgh_finalize(h) == ESTATE
    assert lib.amgh_push_level(h, n, nc, *a, None, None, None, *p, *r, C.byref(gs), C.byref(gs)) == ESTATE
    assert lib.amgh_solve(h, None, x.ctypes.data, 0, 10, 0.0, 1e-8, 1, None, C.byref(it)) == EINVAL
    assert lib.amgh_solve(h, b.ctypes.data, x.ctypes.data, 5, 10, 0.0, 1e-8, 1, None, C.byref(it)) == EINVAL  # bad cycle
    assert lib.amgh_solve(h, b.ctypes.data, x.ctypes.data, 0, -1, 0.0, 1e-8, 1, None, C.byref(it)) == EINVAL
    assert lib.amgh_level_spmv(h, 9, 0, b.ctypes.data, x.ctypes.data) == EINVAL
    assert lib.amgh_solve(h, b.ctypes.data, x.ctypes.data, 0, 50, 0.0, 1e-10, 1, None, C.byref(it)) == 0
    assert np.linalg.norm(A.to_scipy() @ x - b) <= 1e-9 * np.linalg.norm(b) and 0 < it.value <= 50
    assert lib.amgh_num_levels(h) == 1 and lib.amgh_level_size(h, 0) == n and lib.amgh_level_size(h, 1) == nc
    assert lib.amgh_device_bytes(h) > 0
    lib.amgh_destroy(h)
    lib.amgh_destroy(None)                                                   # no-op


def test_push_level_in_two_halves_state_machine():
    """amgh_push_level_begin / _end / _abort: one pending level per handle, everything else refuses meanwhile."""
    lib = AMG.hip_lib()
    A = AMG.poisson(60)
    ml = AMG.ruge_stuben(A, max_levels=2)
    lev = ml.levels[0]
    gs = amgh_smoother_t(1, 2, 1, 0, 1.0)
    a, p, r = _csr(lev.A), (lev.R.colptr.ctypes.data, lev.R.rowval.ctypes.data, lev.R.nzval.ctypes.data), \
        (lev.P.colptr.ctypes.data, lev.P.rowval.ctypes.data, lev.P.nzval.ctypes.data)
    n, nc = lev.A.m, lev.P.n
    fa = _csr(ml.final_A)
    op = np.asfortranarray(ml.coarse_solver.dense_operator())
    h = C.c_void_p()
    assert lib.amgh_create(C.byref(h), 0, 1) == 0
    assert lib.amgh_push_level_end(h, nc, *p, *r) == ESTATE                  # nothing begun
    assert lib.amgh_push_level_abort(h) == ESTATE
    assert lib.amgh_push_level_begin(h, n, None, None, None, None, None, None, C.byref(gs), C.byref(gs)) == EINVAL
    assert lib.amgh_push_level_begin(h, n, *a, None, None, None, C.byref(gs), C.byref(gs)) == 0
    assert lib.amgh_push_level_begin(h, n, *a, None, None, None, C.byref(gs), C.byref(gs)) == ESTATE   # one at a time
    assert lib.amgh_push_level(h, n, nc, *a, None, None, None, *p, *r, C.byref(gs), C.byref(gs)) == ESTATE
    assert lib.amgh_set_coarse(h, nc, *fa, op.ctypes.data) == ESTATE
    assert lib.amgh_finalize(h) == ESTATE
    assert lib.amgh_push_level_abort(h) == 0                                  # "it was the coarsest level after all"
    assert lib.amgh_num_levels(h) == 0
    assert lib.amgh_push_level_begin(h, n, *a, None, None, None, C.byref(gs), C.byref(gs)) == 0
    assert lib.amgh_push_level_end(h, nc, None, None, None, *r) == EINVAL
    assert lib.amgh_push_level_end(h, nc, *p, *r) == 0
    assert lib.amgh_num_levels(h) == 1
    assert lib.amgh_set_coarse(h, nc, *fa, op.ctypes.data) == 0
    assert lib.amgh_finalize(h) == 0
    # the same cycle as the one-call push, bit for bit
    b = np.linspace(0.5, 1.5, n); z1 = np.zeros(n); z2 = np.zeros(n)
    assert lib.amgh_precond_apply(h, b.ctypes.data, z1.ctypes.data, 0) == 0
    ref = ml.device()
    assert lib.amgh_precond_apply(ref.h, b.ctypes.data, z2.ctypes.data, 0) == 0
    assert np.array_equal(z1, z2)
    # a handle destroyed with a level still pending releases it
    h2 = C.c_void_p()
    assert lib.amgh_create(C.byref(h2), 0, 1) == 0
    assert lib.amgh_push_level_begin(h2, n, *a, None, None, None, C.byref(gs), C.byref(gs)) == 0
    lib.amgh_destroy(h2)
    lib.amgh_destroy(h)


def test_levels_prepared_without_a_handle_join_in_order():
    """amgh_level_prepare (several at once, on host threads) / amgh_push_level_prepared / amgh_level_free."""
    import threading
    lib = AMG.hip_lib()
    ml = AMG.ruge_stuben(AMG.poisson((24, 24, 20)), max_levels=3)
    assert len(ml.levels) == 2
    gs = amgh_smoother_t(1, 2, 1, 0, 1.0)
    bad = amgh_smoother_t(9, 2, 1, 0, 1.0)
    preps = [C.c_void_p(), C.c_void_p()]
    rcs = [None, None]

    def prep(i):
        lev = ml.levels[i]
        rcs[i] = lib.amgh_level_prepare(0, lev.A.m, *_csr(lev.A), None, None, None, C.byref(gs), C.byref(gs), C.byref(preps[i]))

    th = [threading.Thread(target=prep, args=(i,)) for i in (1, 0)]      # the coarser one first, both at once
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert rcs == [0, 0] and preps[0].value and preps[1].value
    extra = C.c_void_p()
    a0 = _csr(ml.levels[0].A)
    assert lib.amgh_level_prepare(0, ml.levels[0].A.m, *a0, None, None, None, C.byref(bad), C.byref(gs), C.byref(extra)) == EINVAL
    assert lib.amgh_level_prepare(0, 0, *a0, None, None, None, C.byref(gs), C.byref(gs), C.byref(extra)) == EINVAL
    assert extra.value is None
    assert lib.amgh_level_prepare(0, ml.levels[0].A.m, *a0, None, None, None, C.byref(gs), C.byref(gs), C.byref(extra)) == 0
    lib.amgh_level_free(extra)                                               # never pushed
    lib.amgh_level_free(None)

    def pr(lev):
        return ((lev.R.colptr.ctypes.data, lev.R.rowval.ctypes.data, lev.R.nzval.ctypes.data),
                (lev.P.colptr.ctypes.data, lev.P.rowval.ctypes.data, lev.P.nzval.ctypes.data))

    h = C.c_void_p()
    assert lib.amgh_create(C.byref(h), 0, 1) == 0
    assert lib.amgh_push_level_prepared(h, None) == EINVAL
    assert lib.amgh_push_level_prepared(h, preps[0]) == 0
    assert lib.amgh_push_level_prepared(h, preps[1]) == ESTATE               # one pending level per handle; still the caller's
    p, r = pr(ml.levels[0])
    assert lib.amgh_push_level_end(h, ml.levels[0].P.n, *p, *r) == 0
    assert lib.amgh_push_level_prepared(h, preps[1]) == 0
    p, r = pr(ml.levels[1])
    assert lib.amgh_push_level_end(h, ml.levels[1].P.n, *p, *r) == 0
    op = np.asfortranarray(ml.coarse_solver.dense_operator())
    assert lib.amgh_set_coarse(h, ml.final_A.m, *_csr(ml.final_A), op.ctypes.data) == 0
    assert lib.amgh_finalize(h) == 0
    n = ml.levels[0].A.m
    b = np.linspace(0.5, 1.5, n); z1 = np.zeros(n); z2 = np.zeros(n)
    assert lib.amgh_precond_apply(h, b.ctypes.data, z1.ctypes.data, 0) == 0
    ref = ml.device()
    assert lib.amgh_precond_apply(ref.h, b.ctypes.data, z2.ctypes.data, 0) == 0
    assert np.array_equal(z1, z2)
    # a level of the wrong size for this place in the hierarchy stays the caller's
    wrong = C.c_void_p()
    h2 = C.c_void_p()
    assert lib.amgh_create(C.byref(h2), 0, 1) == 0
    assert lib.amgh_level_prepare(0, ml.levels[0].A.m, *a0, None, None, None, C.byref(gs), C.byref(gs), C.byref(wrong)) == 0
    assert lib.amgh_push_level_prepared(h2, wrong) == 0
    p, r = pr(ml.levels[0])
    assert lib.amgh_push_level_end(h2, ml.levels[0].P.n, *p, *r) == 0
    again = C.c_void_p()
    assert lib.amgh_level_prepare(0, ml.levels[0].A.m, *a0, None, None, None, C.byref(gs), C.byref(gs), C.byref(again)) == 0
    assert lib.amgh_push_level_prepared(h2, again) == EINVAL                 # n != nc of the level above
    lib.amgh_level_free(again)
    lib.amgh_destroy(h2)
    lib.amgh_destroy(h)


def test_hierarchy_without_levels_needs_final_A_and_pcg_rejects_blocks():
    lib = AMG.hip_lib()
    A = AMG.poisson(8)
    op = np.asfortranarray(np.linalg.inv(A.toarray()))
    h = C.c_void_p()
    assert lib.amgh_create(C.byref(h), 0, 1) == 0
    assert lib.amgh_set_coarse(h, 8, None, None, None, op.ctypes.data) == 0
    assert lib.amgh_finalize(h) == ESTATE    # no levels and no final_A: the residual of multilevel.jl:188 has no operator
    lib.amgh_destroy(h)
    ml = AMG.ruge_stuben(AMG.poisson(50))
    dev = ml.device(nrhs=2)
    B = np.ones((50, 2), order="F"); X = np.zeros((50, 2), order="F"); it = C.c_int(0)
    assert lib.amgh_pcg(dev.h, B.ctypes.data, X.ctypes.data, 0, 1, 10, 0.0, 1e-8, None, C.byref(it)) == EUNSUPPORTED


def test_strerror_covers_hip_codes():
    lib = AMG.hip_lib()
    assert b"HIP error" in lib.amgh_strerror(-1001)
    assert b"unknown" in lib.amgh_strerror(-7)


def test_sharded_abi_error_codes():
    """amgh_dist_*: bad arguments and bad states come back as codes, nothing throws or hangs."""
    import ctypes as C
    lib = AMG.hip_lib()
    vp = C.c_void_p
    assert lib.amgh_local_group_create(None, 2) == -2
    g = vp()
    assert lib.amgh_local_group_create(C.byref(g), 0) == -2
    assert lib.amgh_local_group_create(C.byref(g), 1) == 0
    d = vp()
    assert lib.amgh_dist_create_local(C.byref(d), 0, 5, g) == -2            # rank outside the group
    assert lib.amgh_dist_create_local(C.byref(d), 99, 0, g) == -2           # no such device
    assert lib.amgh_dist_create_local(C.byref(d), 0, 0, g) == 0
    # not finalized yet
    assert lib.amgh_dist_precond_apply_d(d, None, None, 0) == -3
    r0, r1 = C.c_int64(0), C.c_int64(0)
    assert lib.amgh_dist_local_range(d, 0, C.byref(r0), C.byref(r1)) == -2
    # push_level: cuts that do not span the level, missing arrays
    A = AMG.poisson((6, 6))
    rp, ci, va = A.csr_arrays()
    pre = AMG.GaussSeidel().c_struct()
    cuts_bad = np.array([0, 5], dtype=np.int64)
    cuts = np.array([0, A.m], dtype=np.int64)
    ccuts = np.array([0, 4], dtype=np.int64)
    args = lambda rc_, cc_: (d, A.m, 4, rc_.ctypes.data, cc_.ctypes.data, rp.ctypes.data, ci.ctypes.data, va.ctypes.data,  # noqa: E731
                             None, None, None, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, rp.ctypes.data,
                             ci.ctypes.data, va.ctypes.data, C.byref(pre), C.byref(pre))
    assert lib.amgh_dist_push_level(*args(cuts_bad, ccuts)) == -2
    assert lib.amgh_dist_push_level(d, A.m, 4, cuts.ctypes.data, ccuts.ctypes.data, None, None, None, None, None, None,
                                    None, None, None, None, None, None, C.byref(pre), C.byref(pre)) == -2
    # a tail that is not finalized / of another block size is refused
    h = vp()
    assert lib.amgh_create(C.byref(h), 0, 1) == 0
    assert lib.amgh_dist_set_tail(d, h) == -2
    lib.amgh_destroy(h)
    # finalize without levels and without a tail: a hierarchy of size 0 is fine, solve on it is a no-op
    # round 5: the modes of the exact sweep, the pipeline query, host tails only on device = -1 handles
    assert lib.amgh_dist_set_gs_mode(d, 3) == -2 and lib.amgh_dist_set_gs_mode(d, -1) == -2
    assert all(lib.amgh_dist_set_gs_mode(d, m) == 0 for m in (0, 2, 1))
    assert lib.amgh_dist_gs_pipelined(d, 0) == -1                            # not finalized
    assert lib.amgh_dist_set_host_tail(d, None, None) == -2                  # a GPU handle executes on the GPU
    assert lib.amgh_dist_set_host_tail(None, None, None) == -2
    assert lib.amgh_dist_finalize(d) == 0
    assert lib.amgh_dist_finalize(d) == -3
    assert lib.amgh_dist_gs_pipelined(d, 0) == -1                            # no such level
    # a tail after finalize: once, on the owner, of the right size — here nothing is collapsed onto this rank (size 0): refused
    h2 = vp()
    assert lib.amgh_create(C.byref(h2), 0, 1) == 0
    assert lib.amgh_dist_set_tail(d, h2) == -2                               # (not finalized itself)
    lib.amgh_destroy(h2)
    assert lib.amgh_dist_set_tail(d, None) == -3
    assert lib.amgh_dist_precond_apply_d(d, None, None, 7) == -2
    assert lib.amgh_dist_spmv_d(d, 0, None, None) == -2                      # no sharded level
    out2 = np.zeros(2, dtype=np.int64)
    assert lib.amgh_dist_stats(d, out2.ctypes.data, 1) == 0 and out2.tolist() == [0, 0]
    assert lib.amgh_strerror(-2005).decode().startswith("RCCL error 5")
    lib.amgh_dist_destroy(d)
    lib.amgh_local_group_destroy(g)


def test_setup_entry_points_reject_bad_arguments():
    """The GPU half of the setup (amgh_setup_*): non-square operators, null outputs and null inputs are AMGH_EINVAL (-2),
    never a crash; a sweep of more right-hand-side columns than a handle was created for cannot be asked for at all
    (amgh_create fixes the block size: multilevel.jl:28-59)."""
    import ctypes as C
    from amg_amd.hierarchy import _DMat
    lib = AMG.hip_lib()
    A = AMG.poisson((12, 10))
    dA = _DMat.upload(A, lib)
    R = AMG.ruge_stuben(A).levels[0].R                       # nc x n: not square
    dR = _DMat.upload(R, lib)
    out = C.c_void_p()
    assert lib.amgh_setup_symmetric_strength(dR.h, 0.0, 0, C.byref(out)) == -2
    assert lib.amgh_setup_symmetric_strength(dA.h, 0.0, 0, None) == -2
    assert lib.amgh_setup_symmetric_strength(None, 0.0, 0, C.byref(out)) == -2
    b = np.ones(A.m)
    bc = np.zeros(A.m)
    assert lib.amgh_setup_fit_candidates_vector(dA.h, None, 1e-10, C.byref(out), bc.ctypes.data) == -2
    assert lib.amgh_setup_fit_candidates_vector(dA.h, b.ctypes.data, 1e-10, None, bc.ctypes.data) == -2
    assert lib.amgh_setup_fit_candidates_vector(None, b.ctypes.data, 1e-10, C.byref(out), bc.ctypes.data) == -2
    assert lib.amgh_debug_set_tunable(b"gs_bw_nc", -1) == 0 and lib.amgh_debug_set_tunable(b"no_such_tunable", 1) == -2
    h = C.c_void_p()
    assert lib.amgh_create(C.byref(h), 0, 65) != 0 and lib.amgh_create(C.byref(h), 0, 0) != 0      # 1 <= nrhs <= 64
