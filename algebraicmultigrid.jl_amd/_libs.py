"""ctypes loading of the two in-tree shared libraries.

libamgsetup.so  host-side setup phase (include/amgsetup.h)
libamghip.so    gfx950 solve phase    (include/amghip.h)

The product path fails loudly when libamghip.so is missing or there is no GPU:
there is no CPU fallback for the solve phase.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SETUP_SO = os.path.join(_HERE, "libamgsetup.so")
HIP_SO = os.path.join(_HERE, "libamghip.so")
HIP_F32_SO = os.path.join(_HERE, "libamghip_f32.so")   # the same source with amgh_real = float (solve phase only)

i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)
i64 = C.c_int64
vp = C.c_void_p


class AMGError(RuntimeError):
    pass


class amgs_options(C.Structure):
    _fields_ = [("theta", C.c_double), ("max_levels", C.c_int32), ("max_coarse", C.c_int32),
                ("hermitian", C.c_int32), ("sa_omega", C.c_double), ("sa_improve_iters", C.c_int32),
                ("sa_B_is_vector", C.c_int32)]


class amgh_smoother_t(C.Structure):
    _fields_ = [("kind", C.c_int32), ("sweep", C.c_int32), ("iter", C.c_int32), ("pad_", C.c_int32),
                ("omega", C.c_double)]


COARSE_FN = C.CFUNCTYPE(C.c_int, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), i64)
COARSE_FN_F32 = C.CFUNCTYPE(C.c_int, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), i64)

_setup = None
_hip = None
_hip_f32 = None


def setup_lib():
    global _setup
    if _setup is not None:
        return _setup
    if not os.path.exists(SETUP_SO):
        raise AMGError(f"{SETUP_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(SETUP_SO)
    L.amgs_last_error.restype = C.c_char_p
    L.amgs_set_threads.argtypes = [C.c_int]
    L.amgs_set_threads_here.argtypes = [C.c_int]
    L.amgs_mat_create.restype = vp
    L.amgs_mat_create.argtypes = [i64, i64, vp, vp, vp]
    L.amgs_mat_alloc.restype = vp
    L.amgs_mat_alloc.argtypes = [i64, i64, i64]
    L.amgs_mat_free.argtypes = [vp]
    for f in ("amgs_mat_rows", "amgs_mat_cols", "amgs_mat_nnz"):
        getattr(L, f).restype = i64
        getattr(L, f).argtypes = [vp]
    for f in ("amgs_mat_colptr", "amgs_mat_rowval", "amgs_mat_nzval"):
        getattr(L, f).restype = vp
        getattr(L, f).argtypes = [vp]
    L.amgs_mat_transpose.restype = vp
    L.amgs_mat_transpose.argtypes = [vp]
    L.amgs_mat_spgemm.restype = vp
    L.amgs_mat_spgemm.argtypes = [vp, vp]
    L.amgs_mat_is_symmetric.argtypes = [vp]
    L.amgs_poisson.restype = vp
    L.amgs_poisson.argtypes = [C.c_int, C.POINTER(i64)]
    L.amgs_classical_strength.argtypes = [vp, C.c_double, C.POINTER(vp), C.POINTER(vp)]
    L.amgs_symmetric_strength.restype = vp
    L.amgs_symmetric_strength.argtypes = [vp, C.c_double, C.c_int]
    L.amgs_rs_splitting.argtypes = [vp, vp]
    L.amgs_rs_cf_splitting_patterns.argtypes = [i64, vp, vp, vp, vp, vp]
    L.amgs_direct_interpolation.restype = vp
    L.amgs_direct_interpolation.argtypes = [vp, vp, vp]
    L.amgs_standard_aggregation.restype = vp
    L.amgs_standard_aggregation.argtypes = [vp]
    L.amgs_fit_candidates.restype = vp
    L.amgs_fit_candidates.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.POINTER(vp), C.POINTER(i64)]
    L.amgs_jacobi_prolongation.restype = vp
    L.amgs_jacobi_prolongation.argtypes = [vp, vp, C.c_double]
    L.amgs_improve_candidates.argtypes = [vp, vp, C.c_int, C.c_int]
    L.amgs_free.argtypes = [vp]
    L.amgs_default_options_rs.argtypes = [C.POINTER(amgs_options)]
    L.amgs_default_options_sa.argtypes = [C.POINTER(amgs_options)]
    L.amgs_ruge_stuben.restype = vp
    L.amgs_ruge_stuben.argtypes = [vp, C.POINTER(amgs_options)]
    L.amgs_smoothed_aggregation.restype = vp
    L.amgs_smoothed_aggregation.argtypes = [vp, vp, C.c_int, C.POINTER(amgs_options)]
    L.amgs_hier_free.argtypes = [vp]
    L.amgs_hier_num_levels.argtypes = [vp]
    L.amgs_hier_get.restype = vp
    L.amgs_hier_get.argtypes = [vp, C.c_int, C.c_int]
    _setup = L
    return L


def _bind_solve_phase(L, creal, coarse_fn):
    """argtypes of the solve-phase handle and the stand-alone operators (both instances of the library)."""
    L.amgh_strerror.restype = C.c_char_p
    L.amgh_strerror.argtypes = [C.c_int]
    L.amgh_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.amgh_destroy.argtypes = [vp]
    L.amgh_destroy.restype = None
    L.amgh_push_level.argtypes = [vp, i64, i64] + [vp] * 12 + [C.POINTER(amgh_smoother_t), C.POINTER(amgh_smoother_t)]
    L.amgh_push_level_begin.argtypes = [vp, i64] + [vp] * 6 + [C.POINTER(amgh_smoother_t), C.POINTER(amgh_smoother_t)]
    L.amgh_push_level_end.argtypes = [vp, i64] + [vp] * 6
    L.amgh_push_level_abort.argtypes = [vp]
    L.amgh_level_prepare.argtypes = [C.c_int, i64] + [vp] * 6 + [C.POINTER(amgh_smoother_t), C.POINTER(amgh_smoother_t), C.POINTER(vp)]
    L.amgh_level_prepare_nrhs.argtypes = [C.c_int, C.c_int, i64] + [vp] * 6 + [C.POINTER(amgh_smoother_t), C.POINTER(amgh_smoother_t), C.POINTER(vp)]
    L.amgh_push_level_prepared.argtypes = [vp, vp]
    L.amgh_level_free.argtypes = [vp]
    L.amgh_level_free.restype = None
    L.amgh_set_coarse.argtypes = [vp, i64, vp, vp, vp, vp]
    L.amgh_set_coarse_host.argtypes = [vp, i64, vp, vp, vp, coarse_fn, vp]
    L.amgh_finalize.argtypes = [vp]
    L.amgh_num_levels.argtypes = [vp]
    L.amgh_tail_dense_build.argtypes = [vp, C.c_int]
    L.amgh_tail_dense_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(C.c_double)]
    L.amgh_level_size.restype = i64
    L.amgh_level_size.argtypes = [vp, C.c_int]
    L.amgh_device_bytes.restype = i64
    L.amgh_device_bytes.argtypes = [vp]
    L.amgh_device_bytes_detail.argtypes = [vp, vp]
    L.amgh_gs_num_dependency_levels.argtypes = [vp, C.c_int]
    L.amgh_gs_num_sweep_steps.argtypes = [vp, C.c_int, C.c_int]
    L.amgh_gs_sweep_stats.argtypes = [vp, C.c_int, C.c_int, vp]
    solve_args = [vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, vp, C.POINTER(C.c_int)]
    L.amgh_solve.argtypes = solve_args
    L.amgh_solve_d.argtypes = solve_args
    L.amgh_precond_apply.argtypes = [vp, vp, vp, C.c_int]
    L.amgh_precond_apply_d.argtypes = [vp, vp, vp, C.c_int]
    pcg_args = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp, C.POINTER(C.c_int)]
    L.amgh_pcg.argtypes = pcg_args
    L.amgh_pcg_d.argtypes = pcg_args
    L.amgh_level_spmv.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.amgh_level_spmv_d.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.amgh_level_residual_d.argtypes = [vp, C.c_int, vp, vp, vp]
    L.amgh_level_smooth.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.amgh_level_smooth_d.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.amgh_csr_create.argtypes = [C.POINTER(vp), C.c_int, i64, i64, vp, vp, vp]
    L.amgh_csr_destroy.argtypes = [vp]
    L.amgh_csr_destroy.restype = None
    L.amgh_csr_prepare.argtypes = [vp, C.c_int, C.c_int]
    L.amgh_csr_spmv_d.argtypes = [vp, vp, vp, vp]
    L.amgh_csr_residual_d.argtypes = [vp, vp, vp, vp, vp]
    L.amgh_csr_spmv_add_d.argtypes = [vp, vp, vp, vp]
    L.amgh_csr_jacobi_d.argtypes = [vp, C.c_double, vp, vp, vp, vp]
    L.amgh_csr_gs_d.argtypes = [vp, C.c_int, C.c_double, C.c_int, vp, vp, vp]
    L.amgh_csr_gs_ex_d.argtypes = [vp, C.c_int, C.c_double, C.c_int, vp, vp, vp, C.c_int]
    L.amgh_gather_d.argtypes = [C.c_int, i64, vp, vp, vp, vp]
    L.amgh_dot_d.argtypes = [C.c_int, i64, vp, vp, vp, C.POINTER(creal), vp]
    L.amgh_cycle_d.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.amgh_set_stream.argtypes = [vp, vp]
    L.amgh_dev_alloc.argtypes = [C.c_int, i64, C.POINTER(vp)]
    L.amgh_dev_free.argtypes = [C.c_int, vp]
    L.amgh_dev_upload.argtypes = [C.c_int, vp, vp, i64]
    L.amgh_dev_download.argtypes = [C.c_int, vp, vp, i64]
    L.amgh_dev_sync.argtypes = [C.c_int]
    L.amgh_stream.restype = vp
    L.amgh_stream.argtypes = [vp]
    L.amgh_timer_begin.argtypes = [vp]
    L.amgh_timer_end.argtypes = [vp, C.POINTER(C.c_double)]
    L.amgh_bench_op.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.amgh_debug_chain_timing.argtypes = [C.c_int, vp]
    L.amgh_debug_merged_sweep_host.argtypes = [i64, i64, vp, vp, vp, C.c_int, C.c_int, C.c_double, vp, vp]
    L.amgh_debug_bw_poll_giveups.argtypes = [vp, C.c_int]
    L.amgh_debug_bw_mode.argtypes = [vp, C.c_int]
    L.amgh_debug_bw_dict.argtypes = [vp, C.c_int]
    L.amgh_debug_bw_late.argtypes = [vp, C.c_int]
    L.amgh_debug_coded_ops.argtypes = [vp, C.c_int]
    L.amgh_debug_bw_sweep_host.argtypes = [i64, vp, vp, vp, C.c_int, C.c_int, C.c_double, vp, vp, vp]
    L.amgh_debug_bw_dict_sweep_host.argtypes = [i64, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.amgh_debug_set_tunable.argtypes = [C.c_char_p, C.c_int]
    L.amgh_profile_enable.argtypes = [vp, C.c_int]
    L.amgh_profile_read.argtypes = [vp, vp, C.c_int]
    L.amgh_set_use_graph.argtypes = [vp, C.c_int]


def _bind_dist(L):
    """argtypes of the row-sharded hierarchy (amgh_dist_*, amgh_local_group_*): both instances of the library."""
    i64p = C.POINTER(i64)
    L.amgh_dist_rccl_available.argtypes = []
    L.amgh_dist_unique_id.argtypes = [vp]
    L.amgh_dist_create_rccl.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp]
    L.amgh_local_group_create.argtypes = [C.POINTER(vp), C.c_int]
    L.amgh_local_group_destroy.argtypes = [vp]
    L.amgh_local_group_destroy.restype = None
    L.amgh_local_group_abort.argtypes = [vp]
    L.amgh_local_group_abort.restype = None
    L.amgh_dist_create_local.argtypes = [C.POINTER(vp), C.c_int, C.c_int, vp]
    L.amgh_dist_create_ipc.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.amgh_dist_destroy.argtypes = [vp]
    L.amgh_dist_destroy.restype = None
    L.amgh_dist_push_level.argtypes = [vp, i64, i64, vp, vp] + [vp] * 12 + [C.POINTER(amgh_smoother_t),
                                                                            C.POINTER(amgh_smoother_t)]
    L.amgh_dist_set_tail.argtypes = [vp, vp]
    L.amgh_dist_set_host_tail.argtypes = [vp, vp, vp]
    L.amgh_dist_finalize.argtypes = [vp]
    L.amgh_dist_set_gs_mode.argtypes = [vp, C.c_int]
    L.amgh_dist_gs_pipelined.argtypes = [vp, C.c_int]
    L.amgh_dist_pipe_serialized.argtypes = [vp]
    L.amgh_dist_pipe_protocol_failed.argtypes = [vp]
    L.amgh_dist_num_sharded_levels.argtypes = [vp]
    L.amgh_dist_local_range.argtypes = [vp, C.c_int, i64p, i64p]
    L.amgh_dist_precond_apply_d.argtypes = [vp, vp, vp, C.c_int]
    L.amgh_dist_solve_d.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, vp, C.POINTER(C.c_int)]
    L.amgh_dist_spmv_d.argtypes = [vp, C.c_int, vp, vp]
    L.amgh_dist_sync.argtypes = [vp]
    L.amgh_dist_barrier.argtypes = [vp]
    L.amgh_dist_allreduce.argtypes = [vp, vp, C.c_int, C.c_int]
    L.amgh_dist_stats.argtypes = [vp, vp, C.c_int]
    L.amgh_dist_device_bytes.restype = i64
    L.amgh_dist_device_bytes.argtypes = [vp]
    L.amgh_dist_stream.restype = vp
    L.amgh_dist_stream.argtypes = [vp]
    L.amgh_dist_plan_info.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.amgh_dist_plan_info2.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]


def hip_lib(dtype=None):
    """Load libamghip.so (dtype None / float64) or its Float32 instance libamghip_f32.so (dtype float32).
    Raises AMGError if it has not been built."""
    global _hip
    if dtype is not None and _np_dtype(dtype).itemsize == 4:
        return hip_lib_f32()
    if _hip is not None:
        return _hip
    if not os.path.exists(HIP_SO):
        raise AMGError(f"{HIP_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the solve phase has no CPU fallback)")
    L = C.CDLL(HIP_SO)
    L.real_dtype = "float64"
    _bind_solve_phase(L, C.c_double, COARSE_FN)
    _bind_dist(L)
    # GPU half of the setup phase (amgh_dmat_*, amgh_setup_*)
    L.amgh_dmat_upload.argtypes = [C.POINTER(vp), C.c_int, i64, i64, vp, vp, vp]
    L.amgh_dmat_download.argtypes = [vp, vp, vp, vp]
    L.amgh_dmat_free.argtypes = [vp]
    L.amgh_dmat_free.restype = None
    for f in ("amgh_dmat_rows", "amgh_dmat_cols", "amgh_dmat_nnz"):
        getattr(L, f).restype = i64
        getattr(L, f).argtypes = [vp]
    L.amgh_setup_transpose.argtypes = [vp, C.POINTER(vp)]
    L.amgh_dmat_equal.argtypes = [vp, vp, C.POINTER(C.c_int)]
    L.amgh_setup_classical_strength.argtypes = [vp, C.c_double, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.amgh_setup_symmetric_strength.restype = C.c_int
    L.amgh_setup_symmetric_strength.argtypes = [vp, C.c_double, C.c_int, C.POINTER(vp)]
    L.amgh_setup_fit_candidates_vector.restype = C.c_int
    L.amgh_setup_fit_candidates_vector.argtypes = [vp, vp, C.c_double, C.POINTER(vp), vp]
    L.amgh_setup_direct_interpolation.argtypes = [vp, vp, vp, C.POINTER(vp), C.POINTER(vp)]
    L.amgh_setup_spgemm.argtypes = [vp, vp, C.POINTER(vp)]
    L.amgh_setup_jacobi_prolongation.argtypes = [vp, vp, C.c_double, C.POINTER(vp)]
    _hip = L
    return L


def _np_dtype(dtype):
    import numpy as np
    return np.dtype(dtype)


def hip_lib_f32():
    """The Float32 instance of the solve phase (same entry points, amgh_real = float)."""
    global _hip_f32
    if _hip_f32 is not None:
        return _hip_f32
    if not os.path.exists(HIP_F32_SO):
        raise AMGError(f"{HIP_F32_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the solve phase has no CPU fallback)")
    L = C.CDLL(HIP_F32_SO)
    L.real_dtype = "float32"
    _bind_solve_phase(L, C.c_float, COARSE_FN_F32)
    _bind_dist(L)
    _hip_f32 = L
    return L


def hip_check(rc, what=""):
    if rc != 0:
        msg = hip_lib().amgh_strerror(rc).decode()
        raise AMGError(f"libamghip: {what}: {msg} (rc={rc})")


def gpu_available():
    try:
        return hip_lib().amgh_device_count() > 0
    except (AMGError, OSError):
        return False
