/*
 * amghip.h — C ABI of libamghip: the MI355X (gfx950) AMG solve phase.
 *
 * Drop-in boundary for ONE hot path of AlgebraicMultigrid.jl: the cycling loop
 * of src/multilevel.jl and the relaxation sweeps of src/smoother.jl.  The
 * reference has no FFI of its own (pure Julia, multiple dispatch); these entry
 * points are what a `ccall` shim binds so that `_solve(ml,b)`,
 * `aspreconditioner(ml)` / `ldiv!`, `smooth!` and `mul!` keep their signatures
 * (see INTEGRATION.md and julia/AMGHip.jl).  Each declaration cites the
 * reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.
 *   - all index arrays 0-based int32, values IEEE f64 (the path computes in f64).
 *   - matrices are CSR.  The reference stores CSC; CSC arrays of M are CSR
 *     arrays of M' — the shim passes whichever view each entry point documents.
 *   - pointers are HOST pointers unless the function name ends in `_d`
 *     (DEVICE pointers on the handle's device; work is enqueued on the handle's
 *     stream and the call returns after the stream has been synchronised unless
 *     documented otherwise).
 *   - return 0 on success, <0 on error: -2 bad argument, -3 bad state,
 *     -4 out of memory, -5 unsupported, -(1000+hipError_t) for HIP failures,
 *     -(2000+ncclResult_t) for RCCL failures (row-sharded entry points).
 *     amgh_strerror(rc) gives text.  No exception crosses the boundary.
 *   - a handle is single-threaded and NOT re-entrant, like the reference's
 *     MultiLevel whose workspace is mutated by every solve (multilevel.jl:23-59).
 */
#ifndef AMGHIP_H
#define AMGHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The element type of matrices and vectors: the reference's hierarchy is generic in eltype(A)
 * (multilevel.jl:14-21, test/runtests.jl:244-259 runs Float64 and Float32).  libamghip.so is the
 * Float64 instance; libamghip_f32.so is the same source compiled with -DAMGH_REAL=float and exports
 * the same entry points with amgh_real = float (the solve-phase handle and the stand-alone operators;
 * scalar parameters — tolerances, relaxation factors, timings — stay double in both).  A caller picks
 * the library by eltype, as the Julia method table would (julia/AMGHip.jl).                          */
#ifndef AMGH_REAL
#define AMGH_REAL double
#endif
typedef AMGH_REAL amgh_real;

typedef struct amgh_handle amgh_t; /* MultiLevel on HBM   (multilevel.jl:14-21) */
typedef struct amgh_csr amgh_csr_t; /* one CSR operator on HBM                   */

#define AMGH_OK 0
#define AMGH_EINVAL (-2)
#define AMGH_ESTATE (-3)
#define AMGH_ENOMEM (-4)
#define AMGH_EUNSUPPORTED (-5)

/* smoother configuration: GaussSeidel{Sweep}(sweep, iter) smoother.jl:17-23,
 * Jacobi(omega; iter) :92-99, SOR(omega, sweep, iter) :173-180                  */
enum { AMGH_SMOOTH_NONE = 0, AMGH_SMOOTH_GS = 1, AMGH_SMOOTH_JACOBI = 2, AMGH_SMOOTH_SOR = 3 };
enum { AMGH_SWEEP_FORWARD = 0, AMGH_SWEEP_BACKWARD = 1, AMGH_SWEEP_SYMMETRIC = 2 };
typedef struct {
  int32_t kind;  /* AMGH_SMOOTH_*                                               */
  int32_t sweep; /* AMGH_SWEEP_* (GS, SOR)                                      */
  int32_t iter;  /* number of sweeps (>= 0)                                     */
  int32_t pad_;
  double omega;  /* Jacobi / SOR damping                                        */
} amgh_smoother_t;

/* cycle types: struct V/W/F <: Cycle, multilevel.jl:116-124                     */
enum { AMGH_CYCLE_V = 0, AMGH_CYCLE_W = 1, AMGH_CYCLE_F = 2 };
/* operator selector for amgh_level_spmv                                         */
enum { AMGH_OP_A = 0, AMGH_OP_P = 1, AMGH_OP_R = 2 };
/* timer labels = the reference's @timeit_debug labels, multilevel.jl:180,216-236 */
enum {
  AMGH_T_PRESMOOTH = 0, AMGH_T_RESIDUAL = 1, AMGH_T_RESTRICT = 2,
  AMGH_T_COARSE = 3, AMGH_T_PROLONG = 4, AMGH_T_POSTSMOOTH = 5, AMGH_T_COUNT = 6
};

const char* amgh_strerror(int rc);
/* number of HIP devices visible (0 when there is no GPU); never fails           */
int amgh_device_count(void);

/* ------------------------------------------------------------------------- */
/* Hierarchy construction  — replaces the in-memory Level/MultiLevel structs    */
/* (multilevel.jl:1-21) and MultiLevelWorkspace (multilevel.jl:23-59).          */
/* ------------------------------------------------------------------------- */
/* nrhs = workspace block size `bs` (multilevel.jl:28-35), 1..64: b and x of the solve entry
 * points are n x nrhs column-major and every operator is applied column by column, as the
 * reference's smoothers do (smoother.jl:77,117).                                          */
int amgh_create(amgh_t** h, int device, int nrhs);
void amgh_destroy(amgh_t* h);

/* push!(levels, Level(A, P, R, pre, post))  (classical.jl:48-52,
 * aggregation.jl:147-151).  Host arrays are copied to HBM; the caller may free
 * them on return.
 *   A : n x n CSR of the true operator — what mul!(res, A, x) applies
 *       (multilevel.jl:188,219).
 *   S : n x n CSR of the matrix the SMOOTHER sweeps row-wise.  With
 *       HermitianSymmetry() the reference's "fast" smoothers read CSC column i
 *       as row i (smoother.jl:81-86,128-134), i.e. S = CSC arrays of A taken as
 *       CSR (= A' ; identical to A when A is symmetric).  With NoSymmetry() the
 *       sweeps act on the true rows, S = A.  Pass S_rowptr = NULL for S == A.
 *   P : n x nc CSR,  R : nc x n CSR  (multilevel.jl:223,233).                  */
int amgh_push_level(amgh_t* h, int64_t n, int64_t nc,
                    const int32_t* A_rowptr, const int32_t* A_col, const amgh_real* A_val,
                    const int32_t* S_rowptr, const int32_t* S_col, const amgh_real* S_val,
                    const int32_t* P_rowptr, const int32_t* P_col, const amgh_real* P_val,
                    const int32_t* R_rowptr, const int32_t* R_col, const amgh_real* R_val,
                    const amgh_smoother_t* pre, const amgh_smoother_t* post);

/* The same push in two halves, for a caller that knows A before P and R — the setup phase itself
 * (classical.jl:36-55: strength and C/F splitting of A come before the interpolation that gives P, R):
 * _begin uploads A (S) and builds the smoother schedules, which needs nothing else; _end adds P and R and
 * appends the level.  _begin may run on another host thread while the caller computes P and R (one
 * pending level per handle; every other call on the handle between the two returns AMGH_ESTATE).        */
int amgh_push_level_begin(amgh_t* h, int64_t n,
                          const int32_t* A_rowptr, const int32_t* A_col, const amgh_real* A_val,
                          const int32_t* S_rowptr, const int32_t* S_col, const amgh_real* S_val,
                          const amgh_smoother_t* pre, const amgh_smoother_t* post);
int amgh_push_level_end(amgh_t* h, int64_t nc,
                        const int32_t* P_rowptr, const int32_t* P_col, const amgh_real* P_val,
                        const int32_t* R_rowptr, const int32_t* R_col, const amgh_real* R_val);
/* Drops the begun level instead (coarsening stopped: size(P, 2) == 0, classical.jl:43).                */
int amgh_push_level_abort(amgh_t* h);

/* _begin without a handle: the level's A (S) goes to HBM and its smoother schedules are built into a free-standing
 * object.  Thread-safe — a setup phase that produces level l+1's A while level l's schedules are still under
 * construction prepares them on different host threads — and amgh_push_level_prepared then makes the object the
 * handle's pending level (exactly the state after _begin; levels join in hierarchy order: _prepared, _end, ...).
 * On AMGH_OK the handle owns the level; otherwise the caller still does (amgh_level_free).                    */
typedef struct amgh_level amgh_level_t;
int amgh_level_prepare(int device, int64_t n,
                       const int32_t* A_rowptr, const int32_t* A_col, const amgh_real* A_val,
                       const int32_t* S_rowptr, const int32_t* S_col, const amgh_real* S_val,
                       const amgh_smoother_t* pre, const amgh_smoother_t* post, amgh_level_t** out);
/* The same with the number of right-hand-side columns the level's sweeps will carry (the nrhs of the handle it is going to
 * join; 0 = unknown = amgh_level_prepare): a single-column hierarchy may get a single-column smoother layout — on
 * stencil-like fine levels the exact Gauss-Seidel sweep as a wavefront of blocks walked by single waves
 * (csrc/hip/gs_blocks.hpp) instead of merged dependency levels.  amgh_push_level_begin passes its handle's nrhs.  */
int amgh_level_prepare_nrhs(int device, int nrhs, int64_t n,
                            const int32_t* A_rowptr, const int32_t* A_col, const amgh_real* A_val,
                            const int32_t* S_rowptr, const int32_t* S_col, const amgh_real* S_val,
                            const amgh_smoother_t* pre, const amgh_smoother_t* post, amgh_level_t** out);
int amgh_push_level_prepared(amgh_t* h, amgh_level_t* level);
void amgh_level_free(amgh_level_t* level);

/* Coarsest level: final_A and the coarse solver (coarse_solver.jl).  The
 * callable `(cs)(x, b)` becomes x = dense_op * b with dense_op (n x n,
 * column-major) computed by the host shim: pinv(Matrix(A)) for Pinv
 * (coarse_solver.jl:9-16) or inv(Matrix(A)) standing in for the QR solve of
 * QRSolver (coarse_solver.jl:66-81).  final_A (CSR) is needed only when the
 * hierarchy has no levels (multilevel.jl:179-180 + residual :188); may be NULL
 * otherwise.                                                                    */
int amgh_set_coarse(amgh_t* h, int64_t n, const int32_t* A_rowptr, const int32_t* A_col,
                    const amgh_real* A_val, const amgh_real* dense_op);

/* Pluggable coarse solver run on the HOST — the counterpart of the reference's
 * coarse-solver protocol `(cs)(x, b)` (coarse_solver.jl:2,35-42,75-81; e.g.
 * LinearSolveWrapper, or QRSolver on a coarsest level too large for a dense
 * operator).  Each coarse solve copies b to the host, calls fn(user, b, x), copies
 * x back (two PCIe hops + a stream sync; meant for large or exotic coarse
 * problems, not for the default path).  fn returns 0 on success.               */
typedef int (*amgh_coarse_fn)(void* user, const amgh_real* b_host, amgh_real* x_host, int64_t n);
int amgh_set_coarse_host(amgh_t* h, int64_t n, const int32_t* A_rowptr, const int32_t* A_col,
                         const amgh_real* A_val, amgh_coarse_fn fn, void* user);

/* Allocates the workspace (res_vecs, coarse_xs, coarse_bs), builds the
 * Gauss-Seidel dependency schedules.  Must be called once before any solve.     */
int amgh_finalize(amgh_t* h);

int amgh_num_levels(const amgh_t* h);            /* length(ml.levels)            */
/* The collapsed coarse tail.  The reference recurses through its small levels for free (__solve!, multilevel.jl:214-239, the
 * recursion at :227-231); on the GPU every level of a few hundred rows still costs ~5 launches of ~5 us.  Every step of that
 * recursion is linear in (x, b) — smooth! (smoother.jl), the residual, R * res, the coarse solver (coarse_solver.jl:16,75-81),
 * x += P * coarse_x — so from the first level with at most `tail_dense_rows` rows (tunable, default 6144; 0 = off; read at
 * amgh_finalize) down, __solve_next! (multilevel.jl:200-212) IS one dense n x n operator per cycle type (V / W / F), and the
 * cycle applies it in ONE launch.  The operator is built from the library's own recursion on the columns of the identity
 * (blocks of `tail_dense_batch` right-hand sides, default 64), either by this call or by the first cycle of that type; a hierarchy whose
 * finest level is that small is one operator altogether (a cycle on a non-zero x: x += M (b - A x), the same iteration).
 * Results: the per-level cycle's up to the rounding of an n-term sum (tests: <= 1e-12 relative).  Not built with a host coarse
 * solver (amgh_set_coarse_host).  Tunable "tail_dense" = 0 (read at every cycle) runs the levels one by one again.
 * amgh_tail_dense_info: level = the level collapsed for that cycle type (-1: none / not built yet), its rows, and the
 * milliseconds all builds of this handle took (any of the three may be NULL).                                              */
int amgh_tail_dense_build(amgh_t* h, int cycle);
int amgh_tail_dense_info(const amgh_t* h, int cycle, int* level, int64_t* rows, double* build_ms);
int64_t amgh_level_size(const amgh_t* h, int l); /* size(levels[l].A,1); l==L: final */
int64_t amgh_device_bytes(const amgh_t* h);      /* HBM held by the handle        */
/* The same by category: out8 = {natural-order A / S / P / R, level-ordered CSR copies (the schedules' A, P, R),
 * un-merged slot arrays, merged groups: CSR part, merged groups: slot arrays, pre-pass triangles,
 * block-inverse data + schedule vectors, workspace + coarse operator}.  Memory-lean mode (AMGH_LEAN=1 or tunable
 * "gs_lean", read at amgh_push_level) keeps only what the default cycle touches: no un-merged slot copy and no CSR
 * copy of slotted composite rows where both sweep directions run merged groups, no natural-order P / R / coarse-level
 * A where the cycle runs level-ordered (amgh_level_spmv on a released operator returns AMGH_EINVAL).                 */
int amgh_device_bytes_detail(const amgh_t* h, int64_t* out8);
/* number of Gauss-Seidel dependency levels of level l (0 if no GS smoother)    */
int amgh_gs_num_dependency_levels(const amgh_t* h, int l);
/* Sequential steps one Gauss-Seidel sweep over level l takes as executed: merged groups of dependency levels
 * (DESIGN.md section 4), 128-row block steps on the block-inverse path, or the dependency levels themselves. */
int amgh_gs_num_sweep_steps(const amgh_t* h, int l, int backward);
/* What one Gauss-Seidel sweep over level l streams, as executed (bench.py's sweep roofline):
 * out6 = {launches, rows, entries of the (composite) rows, entries the slot launches read incl. zero padding,
 *         entries of the pre-pass triangle (merged sweeps), dependency levels per merged group (1 = unmerged)}.   */
int amgh_gs_sweep_stats(const amgh_t* h, int l, int backward, int64_t* out6);

/* ------------------------------------------------------------------------- */
/* Solve phase                                                                  */
/* ------------------------------------------------------------------------- */
/* _solve!(x, ml, b, cycle; maxiter, abstol, reltol, log, calculate_residual)
 * multilevel.jl:158-198.  x holds the initial guess on entry (zeros for
 * `_solve`, :152-157) and the iterate on return.  abstol/reltol as given by the
 * caller (the max(reltol*norm(b), abstol) rule of :170-173 is applied inside).
 * resid_hist: NULL or maxiter+1 doubles; [0] = norm(b), [k] = residual after
 * cycle k (the `log=true` vector, :169,174,191).  *iters = cycles performed.    */
int amgh_solve(amgh_t* h, const amgh_real* b, amgh_real* x, int cycle, int maxiter,
               double abstol, double reltol, int calculate_residual,
               amgh_real* resid_hist, int* iters);
int amgh_solve_d(amgh_t* h, const amgh_real* b_d, amgh_real* x_d, int cycle, int maxiter,
                 double abstol, double reltol, int calculate_residual,
                 amgh_real* resid_hist /*host*/, int* iters);

/* ldiv!(x, p::Preconditioner, b): x .= 0; exactly one cycle, no residual
 * (preconditioner.jl:12-19).  The _d form only enqueues on the handle's stream
 * (no synchronisation) so that a caller-side Krylov loop can stay asynchronous. */
int amgh_precond_apply(amgh_t* h, const amgh_real* r, amgh_real* z, int cycle);
int amgh_precond_apply_d(amgh_t* h, const amgh_real* r_d, amgh_real* z_d, int cycle);

/* One `__solve!(x, ml, cycle, b, lvl)` (multilevel.jl:214-239) starting at level `level` on
 * caller-provided device vectors of that level's size; x is NOT zeroed (the W/F re-entry of
 * multilevel.jl:204-212 continues from the current x).  Enqueue only, no synchronisation.
 * Used by the row-sharded multi-GPU driver for the levels collapsed onto rank 0.        */
int amgh_cycle_d(amgh_t* h, int level, amgh_real* x_d, const amgh_real* b_d, int cycle);
/* Make the handle enqueue on the caller's hipStream_t (e.g. torch's current stream);
 * NULL selects the default (null) stream.                                               */
int amgh_set_stream(amgh_t* h, void* stream);

/* cg(A, b; Pl = aspreconditioner(ml), abstol, reltol, maxiter) — the caller the
 * reference's tests and README pair with ldiv! (cycle_tests.jl:23-27,
 * runtests.jl:186,204, README.md:54-56).  IterativeSolvers.jl's PCG recurrence,
 * entirely on device; x0 = 0.  use_precond = 0 gives plain CG.
 * resid_hist: NULL or maxiter+1 doubles ([0] = norm(b)).                         */
int amgh_pcg(amgh_t* h, const amgh_real* b, amgh_real* x, int cycle, int use_precond,
             int maxiter, double abstol, double reltol, amgh_real* resid_hist, int* iters);
int amgh_pcg_d(amgh_t* h, const amgh_real* b_d, amgh_real* x_d, int cycle, int use_precond,
               int maxiter, double abstol, double reltol, amgh_real* resid_hist, int* iters);

/* ------------------------------------------------------------------------- */
/* Per-level operators — unit-test and roofline hooks                           */
/* ------------------------------------------------------------------------- */
/* mul!(y, M, x) with M = levels[l].A | .P | .R (multilevel.jl:188,219,223,233) */
int amgh_level_spmv(amgh_t* h, int level, int which, const amgh_real* x, amgh_real* y);
int amgh_level_spmv_d(amgh_t* h, int level, int which, const amgh_real* x_d, amgh_real* y_d);
/* res = b - A x  (multilevel.jl:219-220)                                        */
int amgh_level_residual_d(amgh_t* h, int level, const amgh_real* x_d, const amgh_real* b_d, amgh_real* r_d);
/* smooth!(x, levels[l].presmoother | .postsmoother, b)  (smoother.jl:61-90,
 * 113-141, 193-221); post = 0 | 1                                               */
int amgh_level_smooth(amgh_t* h, int level, int post, amgh_real* x, const amgh_real* b);
int amgh_level_smooth_d(amgh_t* h, int level, int post, amgh_real* x_d, const amgh_real* b_d);

/* ------------------------------------------------------------------------- */
/* Stand-alone CSR operators on HBM (used by the row-sharded multi-GPU driver,  */
/* where each rank holds n_local x (n_local + n_halo) blocks, and by tests).     */
/* ------------------------------------------------------------------------- */
int amgh_csr_create(amgh_csr_t** op, int device, int64_t nrows, int64_t ncols,
                    const int32_t* rowptr, const int32_t* col, const amgh_real* val);
void amgh_csr_destroy(amgh_csr_t* op);
/* Build the smoother metadata of the operator now instead of at the first sweep: jacobi != 0 the
 * diagonal table, gs != 0 the Gauss-Seidel dependency schedule (host work + uploads).       */
int amgh_csr_prepare(amgh_csr_t* op, int jacobi, int gs);
/* y = M x ; y = b - M x ; y += M x   (device pointers; stream = hipStream_t or NULL) */
int amgh_csr_spmv_d(amgh_csr_t* op, const amgh_real* x_d, amgh_real* y_d, void* stream);
int amgh_csr_residual_d(amgh_csr_t* op, const amgh_real* x_d, const amgh_real* b_d, amgh_real* r_d, void* stream);
int amgh_csr_spmv_add_d(amgh_csr_t* op, const amgh_real* x_d, amgh_real* y_d, void* stream);
/* one damped-Jacobi sweep on the leading nrows x nrows block's diagonal:
 * xout[i] = (1-w) xin[i] + w (b[i] - sum_{j != i} m_ij xin[j]) / m_ii ; rows with
 * m_ii == 0 keep xin[i] (smoother.jl:113-141).  xin has ncols entries (halo
 * included), xout nrows.                                                        */
int amgh_csr_jacobi_d(amgh_csr_t* op, double omega, const amgh_real* xin_d, const amgh_real* b_d,
                      amgh_real* xout_d, void* stream);
/* Gauss-Seidel / SOR sweep in exact lexicographic order over the leading
 * nrows x nrows block (columns >= nrows are halo entries held fixed);
 * backward != 0 sweeps n..1.  omega = 1 is Gauss-Seidel (smoother.jl:61-90),
 * otherwise SOR (:193-221).                                                     */
int amgh_csr_gs_d(amgh_csr_t* op, int backward, double omega, int is_sor, amgh_real* x_d,
                  const amgh_real* b_d, void* stream);
/* amgh_csr_gs_d with hints.  AMGH_GS_REUSE_B: b_d holds the same values as in the previous sweep on this operator
 * (the sweeps of one smooth! call, the post-smoother after the pre-smoother of a cycle): its dependency-level-ordered
 * copy is still in place and is not gathered again. */
#define AMGH_GS_REUSE_B 1
int amgh_csr_gs_ex_d(amgh_csr_t* op, int backward, double omega, int is_sor, amgh_real* x_d, const amgh_real* b_d,
                     void* stream, int flags);

/* dst[i] = src[idx[i]], i < n  — halo pack / unpack (device pointers).              */
int amgh_gather_d(int device, int64_t n, const int32_t* idx_d, const amgh_real* src_d, amgh_real* dst_d, void* stream);
/* *out (HOST) = sum_i x[i]*y[i] over device vectors; deterministic two-pass wavefront
 * reduction; synchronises `stream`.  scratch_d: >= 1025 doubles of device scratch.  */
int amgh_dot_d(int device, int64_t n, const amgh_real* x_d, const amgh_real* y_d, amgh_real* scratch_d, amgh_real* out,
               void* stream);

/* ------------------------------------------------------------------------- */
/* Row-sharded hierarchy: the cycle of multilevel.jl:214-239 over N GPUs         */
/* (BASELINE.json config C4).  The reference is single-process, so these entry   */
/* points have no reference counterpart; they keep the SHAPES of the single-GPU  */
/* ones above (push_level / finalize / precond_apply / solve) so that a Julia    */
/* shim binds them the same way.  Levels are partitioned by contiguous 1-D row   */
/* ranges (row_cuts: nranks+1 ascending offsets); before every operator the halo */
/* entries of its input vector travel by neighbour send/recv (RCCL over xGMI, or  */
/* peer copies between the handles of one process); levels whose rows all sit on  */
/* one rank ("collapsed") are an ordinary amgh_t on that rank.  Jacobi, residual, */
/* restriction, prolongation are exactly the single-GPU arithmetic; Gauss-Seidel  */
/* / SOR is exact inside a shard with the halo frozen per directional sweep.      */
/* Errors: as above, plus -(2000 + ncclResult_t) for RCCL failures.               */
/* ------------------------------------------------------------------------- */
typedef struct amgh_dist amgh_dist_t;
typedef struct amgh_local_group amgh_local_group_t;
#define AMGH_DIST_ID_BYTES 128
/* RCCL transport (one process per GPU): rank 0 calls amgh_dist_unique_id and hands the 128 bytes to
 * the other ranks by whatever the host has (MPI, a file, torch.distributed's store);
 * amgh_dist_create_rccl is collective (ncclCommInitRank).  librccl is loaded on first use:
 * amgh_dist_rccl_available() == 0 and AMGH_EUNSUPPORTED where it is missing.                       */
int amgh_dist_rccl_available(void);
int amgh_dist_unique_id(void* id128);
int amgh_dist_create_rccl(amgh_dist_t** d, int device, int rank, int nranks, const void* id128);
/* LOCAL transport: N ranks = N handles of ONE process, each driven by its own host thread (every
 * collective call below blocks until all N ranks have made it); halo entries move by device-to-device
 * / peer copies.  Ranks may share a device (virtual ranks on a single-GPU box).  amgh_local_group_abort
 * releases every rank blocked in a collective (they return AMGH_ESTATE) after one of them failed.     */
int amgh_local_group_create(amgh_local_group_t** g, int nranks);
void amgh_local_group_destroy(amgh_local_group_t* g);
void amgh_local_group_abort(amgh_local_group_t* g);
int amgh_dist_create_local(amgh_dist_t** d, int device, int rank, amgh_local_group_t* g);
/* IPC transport: one PROCESS per rank and only shared memory between them — the ranks' packed send
 * buffers are peer-mapped with hipIpc memory handles, the hand-off of an exchange is a sequence flag the
 * producer's stream writes and the consumer's stream waits on (hipStreamWriteValue64 / WaitValue64 on a
 * page of a POSIX shared-memory segment), the consumer pulls its halo entries with one copy kernel per
 * peer; no host thread takes part in an exchange.  Every rank passes the same fresh
 * `shm_name` ("/name", shm_open syntax); rank 0 creates the segment; the call returns when all
 * `nranks` (<= 64) processes have attached (collective).  Ranks may share a device (RCCL refuses that),
 * so a single-GPU box can run the multi-process path.  Host waits are bounded (AMGH_IPC_TIMEOUT_S,
 * default 300 s) and notice peers that died: the survivors release every flag and return AMGH_ESTATE
 * instead of hanging.  device < 0: PLANS ONLY — the collective setup (halo needs, send / receive
 * plans, interior ranges, collapse) in host memory with no GPU call; the solve entry points then
 * return AMGH_EUNSUPPORTED (what the CPU multi-process test drives).                                */
int amgh_dist_create_ipc(amgh_dist_t** d, int device, int rank, int nranks, const char* shm_name);
void amgh_dist_destroy(amgh_dist_t* d);
/* push!(levels, Level(A, P, R, pre, post)) for one SHARDED level: this rank passes its rows only —
 * rows [row_cuts[rank], row_cuts[rank+1]) of A (n x n), S (as in amgh_push_level; NULL for S == A)
 * and P (n x nc), rows [crow_cuts[rank], crow_cuts[rank+1]) of R (nc x n) — as CSR with rowptr
 * starting at 0 and GLOBAL column indices.  crow_cuts is the row partition of the next level and must
 * equal the row_cuts of the next push; the level after the last pushed one must sit on ONE rank
 * (crow_cuts = 0,..,0,nc,..,nc), which passes the collapsed levels to amgh_dist_set_tail.           */
int amgh_dist_push_level(amgh_dist_t* d, int64_t n_global, int64_t nc_global,
                         const int64_t* row_cuts, const int64_t* crow_cuts,
                         const int32_t* A_rowptr, const int32_t* A_col, const amgh_real* A_val,
                         const int32_t* S_rowptr, const int32_t* S_col, const amgh_real* S_val,
                         const int32_t* P_rowptr, const int32_t* P_col, const amgh_real* P_val,
                         const int32_t* R_rowptr, const int32_t* R_col, const amgh_real* R_val,
                         const amgh_smoother_t* pre, const amgh_smoother_t* post);
/* The collapsed levels: a finalized single-GPU handle (nrhs = 1, same device) holding levels lc.. and
 * the coarse solver, on the rank that owns them; NULL elsewhere.  Borrowed, not owned; its stream
 * becomes the sharded handle's.  Before amgh_dist_finalize — or, on the owner, once AFTER it (the tail's
 * smoother schedules and the shards' plans are independent work: build them side by side); the solve
 * entry points return AMGH_ESTATE on an owner that has not passed its tail yet.                          */
int amgh_dist_set_tail(amgh_dist_t* d, amgh_t* tail);
/* device = -1 handles: the collapsed levels as a HOST callback (x = fn(b): one visit of __solve! at level lc, multilevel.jl:214-239;
 * set on every rank before amgh_dist_finalize, called on the owner).  With it the handle EXECUTES the sharded cycle in host
 * memory — the same C code: halo plans, exchange ordering, turns of the exact Gauss-Seidel, collapse, all-reduces — with the
 * operators as plain loops; amgh_dist_precond_apply_d / amgh_dist_solve_d / amgh_dist_spmv_d then take HOST pointers.  What the
 * CPU multi-process test (gloo-launched, no GPU) runs against the oracle.  V cycles (a W / F cycle's second visit of the
 * collapsed levels would need their previous x).                                                                          */
int amgh_dist_set_host_tail(amgh_dist_t* d, amgh_coarse_fn fn, void* user);
/* Collective.  Exchanges the halo needs, builds the send / receive plans, uploads the local blocks
 * (columns renumbered to [local | halo]) and builds the smoother schedules of the shards.            */
int amgh_dist_finalize(amgh_dist_t* d);
/* Gauss-Seidel / SOR on the sharded levels (smoother.jl:61-90, :193-221; collective in effect: the same mode on every rank).
 * mode = 1 (default): the whole level is swept in exact lexicographic order, the cycle is the reference's (1e-10).  Where every
 *   rank holds the dataflow layout of its shard (amgh_dist_gs_pipelined) it is ONE sweep pipelined across the ranks: all
 *   ranks launch at once, a block that reads rows of the neighbouring rank polls their mailboxes in that rank's memory
 *   (peer-mapped) and starts on them as it starts on rows of its own rank's blocks — one dependency chain through the level,
 *   one exchange per directional sweep.  Elsewhere the ranks sweep in turn (mode 2).
 * mode = 2: exact order with the ranks strictly in turn (upward in a forward sweep, downward in a backward one), every
 *   turn's boundary values travelling before the next: the sum of the shards' sweeps plus nranks - 1 exchanges.
 * mode = 0: the processor-block hybrid — every shard sweeps at once, exact inside, halo frozen per directional sweep:
 *   a different (convergent) iteration that scales with the ranks.  Jacobi is exact across shards in every mode.        */
int amgh_dist_set_gs_mode(amgh_dist_t* d, int mode);
/* 1 when level `level`'s Gauss-Seidel / SOR sweeps run as one pipelined sweep under mode 1, 0 when in turns, < 0: no such level */
int amgh_dist_gs_pipelined(const amgh_dist_t* d, int level);
/* 1 when amgh_dist_finalize found that ranks of ONE process sharing ONE device (virtual ranks: a test and measurement
 * arrangement) do not run their sweep streams concurrently — a process's streams are spread over a few hardware queues, two on
 * the same queue run one after the other — and therefore left every level to the turn loop (a pipelined sweep would wait for a
 * neighbour that cannot start); 0 otherwise, -1 before finalize.  Probed with a bounded flag exchange between the streams. */
int amgh_dist_pipe_serialized(const amgh_dist_t* d);
/* 1 when amgh_dist_finalize ran the mailbox protocol itself between neighbouring ranks — kMailProbeRounds lockstep rounds of the
 * sweeps' own write-through store / system-scope poll on memory allocated and mapped as the levels' mailbox arrays are (the
 * neighbour's pointer with peer access enabled, or hipIpcOpenMemHandle) — and it did NOT come back right (mapping, peer access,
 * visibility across devices, the bound): every level is then left to the turn loop; 0 when it passed or was not needed, -1
 * before finalize.  (The reference is one process, multilevel.jl:214-239; this guards the pipelined sweep across devices.)       */
int amgh_dist_pipe_protocol_failed(const amgh_dist_t* d);
int amgh_dist_num_sharded_levels(const amgh_dist_t* d);
int amgh_dist_local_range(const amgh_dist_t* d, int level, int64_t* r0, int64_t* r1);
/* ldiv! / _solve! on this rank's rows of the fine vectors (device pointers; collective).
 * amgh_dist_precond_apply_d only enqueues (amgh_dist_sync waits); amgh_dist_solve_d returns after
 * the last iteration; its stopping test uses the global residual norm (identical on every rank).     */
int amgh_dist_precond_apply_d(amgh_dist_t* d, const amgh_real* r_loc_d, amgh_real* z_loc_d, int cycle);
int amgh_dist_solve_d(amgh_dist_t* d, const amgh_real* b_loc_d, amgh_real* x_loc_d, int cycle, int maxiter,
                      double abstol, double reltol, int calculate_residual,
                      amgh_real* resid_hist /*host*/, int* iters);
/* y_loc = A_level x_loc, halo exchange included (roofline hook of the sharded SpMV).  Enqueue only.
 * x_loc_d = NULL multiplies the level's resident x (as the last cycle left it) without a copy.       */
int amgh_dist_spmv_d(amgh_dist_t* d, int level, const amgh_real* x_loc_d, amgh_real* y_loc_d);
int amgh_dist_sync(amgh_dist_t* d);
int amgh_dist_barrier(amgh_dist_t* d);                                 /* sync + barrier over the ranks */
int amgh_dist_allreduce(amgh_dist_t* d, double* v, int n, int max_op); /* host values, sum or max       */
/* out2 = {halo exchanges, bytes sent by this rank} since the last reset                              */
int amgh_dist_stats(amgh_dist_t* d, int64_t* out2, int reset);
int64_t amgh_dist_device_bytes(const amgh_dist_t* d);
void* amgh_dist_stream(amgh_dist_t* d);
/* Halo plan of x on `level` (tests): out_counts = {local rows, halo entries, entries sent, first and
 * end interior row of A}; optional arrays: halo_globals[nhalo], send_idx[nsend] (local indices),
 * per-peer send / receive counts [nranks].                                                           */
int amgh_dist_plan_info(const amgh_dist_t* d, int level, int64_t* out_counts, int64_t* halo_globals,
                        int32_t* send_idx, int64_t* send_cnt_per_peer, int64_t* recv_cnt_per_peer);
/* the same for which = 0: x of `level` (level = number of sharded levels: the first collapsed level),
 * which = 1: the residual of `level` (read by R); interior rows are those of A / R                   */
int amgh_dist_plan_info2(const amgh_dist_t* d, int level, int which, int64_t* out_counts,
                         int64_t* halo_globals, int32_t* send_idx, int64_t* send_cnt_per_peer,
                         int64_t* recv_cnt_per_peer);

/* ------------------------------------------------------------------------- */
/* Setup phase, data-parallel half on the GPU: what             */
/* extend_hierarchy_rs! (classical.jl:36-55) does per level except the sequential */
/* C/F splitting (splitting.jl:25-159, host: amgs_rs_cf_splitting_patterns).      */
/* Matrices are Julia's CSC on HBM; results are bitwise those of the host library */
/* libamgsetup (same order of every floating-point operation).                    */
/* ------------------------------------------------------------------------- */
typedef struct amgh_dmat amgh_dmat_t; /* SparseMatrixCSC{Float64,Int32} on HBM (0-based) */
int amgh_dmat_upload(amgh_dmat_t** M, int device, int64_t m, int64_t n, const int32_t* colptr,
                     const int32_t* rowval, const double* nzval);
/* any of the three output arrays may be NULL (patterns only: nzval = NULL)      */
int amgh_dmat_download(const amgh_dmat_t* M, int32_t* colptr, int32_t* rowval, double* nzval);
void amgh_dmat_free(amgh_dmat_t* M);
int64_t amgh_dmat_rows(const amgh_dmat_t* M);
int64_t amgh_dmat_cols(const amgh_dmat_t* M);
int64_t amgh_dmat_nnz(const amgh_dmat_t* M);
/* copy(A')                                                                       */
int amgh_setup_transpose(const amgh_dmat_t* A, amgh_dmat_t** At);
/* *same = 1 when A and B hold identical arrays (issymmetric(A): compare with its transpose) */
int amgh_dmat_equal(const amgh_dmat_t* A, const amgh_dmat_t* B, int* same);
/* S, T = Classical(theta)(At)  (strength.jl:7-37): T = thresholded |At| scaled by its column maxima,
 * S = T'.  Sn, Tn (both NULL or both non-NULL): the PATTERNS the C/F splitting consumes — S without its
 * diagonal (remove_diag, splitting.jl:8-18) and its transpose — so the host needs no matrix work of its own. */
int amgh_setup_classical_strength(const amgh_dmat_t* At, double theta, amgh_dmat_t** S, amgh_dmat_t** T,
                                  amgh_dmat_t** Sn, amgh_dmat_t** Tn);
/* S = SymmetricStrength(theta)(A, bsr_flag)  (strength.jl:77-122): off-diagonal entries with a_ij^2 < theta^2 |a_ii| |a_jj|
 * and stored zeros dropped, |.| of the rest scaled by the column maxima (bsr_flag with theta = 0: the pattern of A with
 * ones) — the strength of smoothed_aggregation, bitwise the host library's.                                     */
int amgh_setup_symmetric_strength(const amgh_dmat_t* A, double theta, int bsr_flag, amgh_dmat_t** S);
/* T, Bc = fit_candidates(AggOp, B::Vector; tol)  (aggregation.jl:161-193): AggOp is n_coarse x n_fine (one entry per aggregated
 * fine node), B and Bc HOST vectors of n_fine / n_coarse entries; T = AggOp' with column i = B restricted to aggregate i,
 * normalised (Bc[i] its norm) — the tentative prolongator, bitwise the host library's.  (Blocks of candidates take the QR
 * per aggregate of the host library, amgs_fit_candidates.)                                                        */
int amgh_setup_fit_candidates_vector(const amgh_dmat_t* AggOp, const double* B, double tol, amgh_dmat_t** T, double* Bc);
/* P, R = direct_interpolation(At, T, splitting)  (classical.jl:57-189); splitting is a HOST array
 * (1 = C node, 0 = F node).  R: nc x n, P = R': n x nc.                                                        */
int amgh_setup_direct_interpolation(const amgh_dmat_t* At, const amgh_dmat_t* T, const int32_t* splitting,
                                    amgh_dmat_t** R, amgh_dmat_t** P);
/* C = X * Y with SparseArrays' spmatmul semantics (structural zeros kept, rows sorted): the two products of
 * RAP = R * A * P (classical.jl:44).  AMGH_EUNSUPPORTED when a column of the product has more than ~440
 * entries (the caller then forms this product with the host library).                                          */
int amgh_setup_spgemm(const amgh_dmat_t* X, const amgh_dmat_t* Y, amgh_dmat_t** C);
/* P = JacobiProlongation(omega)(A, T) = T - (omega * D^-1 * A) * T with D_i = sum_j |a_ij| (aggregation.jl:30-59):
 * the prolongation smoothing of smoothed_aggregation, the same sums in the same order as the host library
 * (bitwise the same P).  AMGH_EUNSUPPORTED as amgh_setup_spgemm.                                               */
int amgh_setup_jacobi_prolongation(const amgh_dmat_t* A, const amgh_dmat_t* T, double omega, amgh_dmat_t** P);

/* ------------------------------------------------------------------------- */
/* Device memory + timing helpers for hosts without a HIP binding of their own  */
/* ------------------------------------------------------------------------- */
int amgh_dev_alloc(int device, int64_t bytes, void** ptr_d);
int amgh_dev_free(int device, void* ptr_d);
int amgh_dev_upload(int device, void* dst_d, const void* src, int64_t bytes);
int amgh_dev_download(int device, void* dst, const void* src_d, int64_t bytes);
int amgh_dev_sync(int device);
void* amgh_stream(amgh_t* h); /* the handle's hipStream_t                        */

/* hipEvent timing on the handle's stream: begin records an event, end records
 * a second one, synchronises and returns the elapsed milliseconds.              */
int amgh_timer_begin(amgh_t* h);
int amgh_timer_end(amgh_t* h, double* ms);
/* Time `reps` back-to-back launches of one operator kernel with HIP events on
 * the launching stream; returns the average milliseconds per launch.
 * which: AMGH_OP_A/P/R = SpMV; 3 = fused residual with A; 4 = pre-smoother.     */
int amgh_bench_op(amgh_t* h, int level, int which, int reps, int warmup, double* avg_ms);

/* Per-label, per-level accumulated milliseconds since the last reset, mirroring
 * TimerOutputs' six labels (multilevel.jl:180,216-236).  Enabling inserts HIP
 * events around every step (small overhead).  out: AMGH_T_COUNT x (num_levels+1)
 * doubles, label-major.                                                         */
int amgh_profile_enable(amgh_t* h, int on);
int amgh_profile_read(amgh_t* h, double* out, int reset);

/* Diagnostics: accumulate shader-cycle counts of the phases of gs_chain_kernel's per-level loop
 * (issue, gather+stage, barrier, row sums, store+barrier; levels; launches).  enable != 0 starts
 * accumulating; out8 (8 x uint64, may be NULL) receives and resets the sums.               */
int amgh_debug_chain_timing(int enable, unsigned long long* out8);
/* Host-only emulation of one merged-level Gauss-Seidel sweep (the construction behind the device schedules:
 * dependency levels, groups of m levels made independent by substitution, pre-pass over the other triangle).
 * No device work: CPU tests check it against the scalar lexicographic sweep (smoother.jl:78-88).
 * x: ncols entries in/out (columns >= nrows are frozen halo values), b: nrows.  Returns the number of groups. */
int amgh_debug_merged_sweep_host(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* col,
                                 const amgh_real* val, int m, int backward, double omega /* 1 = Gauss-Seidel, else SOR */,
                                 amgh_real* x, const amgh_real* b);

/* Diagnostics: one Gauss-Seidel (omega = 1) or SOR sweep as a WAVEFRONT OF BLOCKS (csrc/hip/gs_blocks.hpp) computed on the
 * host from the very plan and packed records the device kernel uses — block partition, launches, steps, the reciprocal
 * quotient.  x (in / out), b: nrows entries; stats4 = {blocks, launches, sum of the deepest blocks' steps, external
 * columns} (may be NULL).  AMGH_EUNSUPPORTED when the operator cannot be laid out (rows of more than 18 off-diagonal
 * entries).  CPU tests compare it bit for bit with the scalar lexicographic sweep (smoother.jl:61-90).  Before it
 * sweeps it checks what the chained kernel relies on — every external position of a block lies in a block of its
 * predecessor list (smaller tickets) or of its successor list (larger tickets), split at the descriptor's npre, the two
 * lists transposes of each other — and returns AMGH_ESTATE if the plan violates it.                                    */
int amgh_debug_bw_sweep_host(int64_t nrows, const int32_t* rowptr, const int32_t* col, const amgh_real* val, int target_rows,
                             int backward, double omega, amgh_real* x, const amgh_real* b, int64_t* stats4);

/* Diagnostics: one Gauss-Seidel sweep executed on the host FROM THE DICTIONARY LAYOUT of the dataflow records (csrc/hip/
 * gs_flow.hpp FlowDict: per row its column chunks and a dictionary index, per block its distinct value rows) — what the
 * relayed and the multi-column kernels read where an operator's rows repeat.  x (in / out), b: nrows entries; stats5 = {1 if
 * the operator has the layout (else x is left untouched), dictionary rows in all, rows of the largest dictionary, bytes of
 * column records + dictionaries, bytes of the plain records} (may be NULL).  AMGH_EUNSUPPORTED: no block layout or a
 * pattern that is not structurally symmetric.  CPU tests compare it bit for bit with the scalar loop (smoother.jl:61-90). */
int amgh_debug_bw_dict_sweep_host(int64_t nrows, const int32_t* rowptr, const int32_t* col, const amgh_real* val, int target_rows,
                                  int backward, amgh_real* x, const amgh_real* b, int64_t* stats5);

/* Diagnostics of the single-launch wavefronts of blocks (blocks ordered by flags or by the data they wait for instead of
 * kernel boundaries): the process-wide word a bounded poll raises when it gives up, read behind a device
 * synchronisation — always 0 (a block only waits for blocks holding smaller tickets).  A raised word is what turns the
 * next synchronising entry point (amgh_solve, amgh_pcg, amgh_precond_apply, amgh_level_smooth, amgh_dev_sync; the
 * asynchronous amgh_precond_apply_d on its next call) into AMGH_ESTATE instead of handing out the numbers of a sweep
 * that went on with stale values; this call leaves it as it is.  -1 when level l has no such schedule.              */
int amgh_debug_bw_poll_giveups(const amgh_t* h, int l);

/* Diagnostics: how level l's Gauss-Seidel / SOR sweeps of ONE right-hand side run with the tunables as they stand
 * (smoother.jl:61-90, :193-221 — exact lexicographic order in every mode): 0 = level schedules (no wavefront of blocks),
 * 1 = wavefront of blocks, one launch per depth of the block graph, 2 = one launch, blocks chained by flags, 3 = one
 * launch as a dataflow (rows published into mailboxes as they are computed; structurally symmetric patterns).  -1: no
 * such level.                                                                                                         */
int amgh_debug_bw_mode(const amgh_t* h, int l);
/* ... and whether the single-column dataflow sweep of level l reads the DICTIONARY layout (column records + every block's
 * distinct value rows, held in LDS; tunable "gs_bw_dict"): 1 / 0, -1: no such level.                                */
int amgh_debug_bw_dict(const amgh_t* h, int l);
/* ... and whether its rows are summed with the sweep's FAR side above the hand-over (the relayed walk's dependency-aware row
 * sum: the entries of smoother.jl:81-87's sum whose x cannot change any more are added first, the near half behind the
 * hand-over — the same iterate, one reassociation per row; tunable "gs_bw_inorder" = 1 restores the stored-order sum, the
 * scalar loop's bits): 1 / 0, -1: no such level.  Needs records whose entries lie split around the padding with at most
 * half of the slots on either side (every non-sharded operator whose rows are that balanced).                         */
int amgh_debug_bw_late(const amgh_t* h, int l);
/* Diagnostics: which operators of level l the level-ordered cycle streams as VALUE-CODED columns (one 32-bit word per
 * entry: column | code << 24, the code an index into the operator's table of distinct values — built at amgh_finalize for
 * operators of >= 2^18 rows, fewer than 2^24 columns and at most 256 distinct values; the sums are the plain kernel's, bit
 * for bit; tunable "stream_code"): bit 0 = A (residual, multilevel.jl:219-220), bit 1 = R (restriction, :221), bit 2 = P
 * (prolongation, :233-234).  -1: no such level.                                                                       */
int amgh_debug_coded_ops(const amgh_t* h, int l);

/* Diagnostics: tunables of the Gauss-Seidel execution (process-wide; used by tools/ to pick the defaults and by
 * tests to force every path).  Read at every sweep: "gs_xcd_map", "gs_block_pipe", "gs_flip", "gs_keep_lo",
 * "gs_slots", "gs_block_target", "gs_min_rows", "gs_threads", "gs_nnz_per_wg", "gs_tiny" (operators that fit LDS
 * entirely: 1 = walked by a single wave from their packed record where one was built, else the whole-operator chain
 * kernel; 2 = the whole-operator chain kernel; 0 = the regular chain kernel — all three bitwise the same sweep),
 * "jacobi_zero", "rhs_il", and "gs_merge" <= 1 /
 * "gs_block_inverse" = 0 to bypass already-built merged groups / block-inverse data.  Read when a schedule is BUILT
 * (amgh_push_level, first stand-alone sweep of an operator): "gs_merge" (largest group of dependency levels tried),
 * "gs_bigslot" (0 off, 1 cost model, 2 always), "gs_super" (blocks per superblock), "gs_block_inverse", "gs_bw" (the
 * wavefront-of-blocks layout of single-column hierarchies: 0 off, 1 where its cost model prefers it, 2 always),
 * "gs_bw_rows" (rows per block aimed at, 512), "gs_bw_min_rows" (smallest operator considered in mode 1: 30 000 rows, half of it for operators of at most 7 entries per row — above it the cost model decides);
 * "gs_bw_two_min_rows" (operators with TWO offset classes — 2-D grids — take that layout from this many rows: 200 000;
 * 0 never), "gs_bw_nrhs" (1: hierarchies created for blocks of right-hand sides get it too, on structurally symmetric
 * levels), "gs_bw_flow" (1: the dataflow layout of the wavefront is built where the pattern is structurally symmetric);
 * read at every sweep: "gs_bw_flow" (0: the chained / launched execution where its layout was kept), "gs_bw_chain" (1: the
 * wavefront of blocks as ONE launch per sweep, blocks chained by flags; 0: one launch per depth of the quotient graph —
 * bitwise the same sweep), "gs_bw_nc" (columns of a block of right-hand sides one workgroup of the dataflow sweep carries: -1 = 3 on the dictionary layout, 2 on plain records;
 * 0 = as many as are instantiated), "gs_bw_spin" / "gs_bw_skip_pub" (test hooks: bound of a poll, a block that publishes
 * nothing), "gs_flow_xzero" (1: a dataflow sweep that starts a smooth! call on x = 0 reads no x — bitwise the same),
 * "gs_bw_dict" (1: dataflow schedules carry the dictionary layout where the rows' values repeat — at most 256 distinct
 * value rows per block, a quarter of the rows overall — and the relayed and the multi-column sweeps read it, bitwise the
 * same; read at build and at every sweep),
 * "stream_code" (1: value-coded columns, see amgh_debug_coded_ops; read at amgh_finalize and at every launch),
 * "gs_dup_launch" (measurement hook: every merged-group launch issued 1 + that many times).
 * Returns AMGH_EINVAL for an unknown name.                                                                       */
int amgh_debug_set_tunable(const char* name, int value);

/* Replay whole cycles from captured hipGraphs (default off: measured no gain on MI355X for big hierarchies —
 * the cycle is GPU-latency-bound and the host runs far ahead — nor for small ones, whose kernels take >= 3 us each;
 * AMGH_USE_GRAPH=1 in the environment also enables it, AMGH_GRAPH_AUTO=1 enables it for hierarchies whose widest
 * level has at most 65 536 rows; rocprofv3's kernel tracing aborts on graph replays of this size).              */
int amgh_set_use_graph(amgh_t* h, int on);

#ifdef __cplusplus
}
#endif
#endif
