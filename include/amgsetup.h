/*
 * amgsetup.h — C ABI of libamgsetup: the HOST-side (CPU) AMG setup phase that
 * produces the Multilevel/Level hierarchy consumed by libamghip (amghip.h).
 *
 * The setup phase is outside the GPU hot path: it runs once,
 * is graph-sequential (RS splitting, greedy aggregation) and stays on the CPU.
 * It exists because there is no Julia on either machine: without it no
 * hierarchy could be built for the V-cycle kernels.
 *
 * Matrices cross this boundary exactly as the reference holds them: compressed
 * sparse COLUMN arrays (colptr, rowval, nzval), here 0-based int32 / f64.
 * (For a symmetric matrix the same arrays are also its CSR.)
 *
 * Reference interfaces restated (paths relative to /root/reference):
 *   amgs_poisson                 src/gallery.jl:1-63
 *   amgs_classical_strength      src/strength.jl:7-70
 *   amgs_symmetric_strength      src/strength.jl:77-122
 *   amgs_rs_splitting            src/splitting.jl:8-159
 *   amgs_direct_interpolation    src/classical.jl:57-189
 *   amgs_standard_aggregation    src/aggregate.jl:12-134
 *   amgs_fit_candidates          src/aggregation.jl:161-230
 *   amgs_jacobi_prolongation     src/aggregation.jl:10-59
 *   amgs_ruge_stuben             src/classical.jl:6-55
 *   amgs_smoothed_aggregation    src/aggregation.jl:66-157
 *
 * Error convention: functions returning int give 0 on success, <0 on error
 * (amgs_last_error() returns the message); functions returning a pointer give
 * NULL on error.  Handles are not thread-safe.
 */
#ifndef AMGSETUP_H
#define AMGSETUP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct amgs_mat amgs_mat;   /* sparse matrix, CSC, int32/f64          */
typedef struct amgs_hier amgs_hier; /* hierarchy: levels (A,P,R) + final_A    */

const char* amgs_last_error(void);
int amgs_set_threads(int nthreads); /* OpenMP threads for SpGEMM/transposes   */
/* The same for the CALLING host thread only (OpenMP's setting is per host thread; other host threads keep the
 * library-wide count of amgs_set_threads): two host threads that run library calls side by side split the cores. */
int amgs_set_threads_here(int nthreads);

/* ---- matrices ---------------------------------------------------------- */
amgs_mat* amgs_mat_create(int64_t m, int64_t n, const int32_t* colptr,
                          const int32_t* rowval, const double* nzval);
/* an m x n matrix with room for nnz stored entries; the caller fills the three arrays in place (the GPU half of the
 * setup downloads into them) and answers for their consistency */
amgs_mat* amgs_mat_alloc(int64_t m, int64_t n, int64_t nnz);
void amgs_mat_free(amgs_mat*);
int64_t amgs_mat_rows(const amgs_mat*);
int64_t amgs_mat_cols(const amgs_mat*);
int64_t amgs_mat_nnz(const amgs_mat*);
/* borrowed pointers into the matrix (valid until amgs_mat_free) */
const int32_t* amgs_mat_colptr(const amgs_mat*);
const int32_t* amgs_mat_rowval(const amgs_mat*);
const double* amgs_mat_nzval(const amgs_mat*);
amgs_mat* amgs_mat_transpose(const amgs_mat*);          /* copy(A')          */
amgs_mat* amgs_mat_spgemm(const amgs_mat* X, const amgs_mat* Y); /* X*Y       */
int amgs_mat_is_symmetric(const amgs_mat*);  /* arrays identical to A' ?      */

/* ---- gallery (src/gallery.jl) ------------------------------------------ */
amgs_mat* amgs_poisson(int ndim, const int64_t* dims);

/* ---- setup kernels, exposed for the reference's unit-level goldens ------ */
/* Classical(theta)(At) -> (S = copy(T'), T)                                  */
int amgs_classical_strength(const amgs_mat* At, double theta, amgs_mat** S,
                            amgs_mat** T);
/* SymmetricStrength(theta)(A, bsr_flag) -> S                                 */
amgs_mat* amgs_symmetric_strength(const amgs_mat* A, double theta,
                                  int bsr_flag);
/* RS()(S): removes the diagonal of S IN PLACE, then RS_CF_splitting.         */
/* splitting[n]: 0 = F, 1 = C                                                 */
int amgs_rs_splitting(amgs_mat* S, int32_t* splitting);
/* The sequential sweep alone (splitting.jl:25-159) on two PATTERNS: S without its diagonal and T = S' (column      */
/* pointers n+1, row indices) — what the GPU setup path hands over (amgh_setup_classical_strength, amghip.h).      */
int amgs_rs_cf_splitting_patterns(int64_t n, const int32_t* Sp, const int32_t* Sj, const int32_t* Tp,
                                  const int32_t* Tj, int32_t* splitting);
/* direct_interpolation(At, T, splitting) -> R (n_c x n, CSC); P = R'         */
amgs_mat* amgs_direct_interpolation(const amgs_mat* At, const amgs_mat* T,
                                    const int32_t* splitting);
/* StandardAggregation()(S) -> AggOp (n_agg x n, CSC)                         */
amgs_mat* amgs_standard_aggregation(const amgs_mat* S);
/* fit_candidates(AggOp, B): B is n x nB column-major.  vector_path != 0     */
/* selects the Vector method (nB must be 1); otherwise the per-aggregate QR   */
/* method.  Returns Q (n x n_coarse) and writes the coarse candidates         */
/* (n_coarse x nB, column-major) into *Bc (malloc'ed; free with amgs_free).   */
amgs_mat* amgs_fit_candidates(const amgs_mat* AggOp, const double* B, int nB,
                              int vector_path, double tol, double** Bc,
                              int64_t* n_coarse);
/* JacobiProlongation(omega)(A, T, S, B, degree=1, LocalWeighting())          */
amgs_mat* amgs_jacobi_prolongation(const amgs_mat* A, const amgs_mat* T,
                                   double omega);
/* improve_candidates(A, B, 0): `iters` symmetric Gauss-Seidel sweeps on A x = 0 applied to every column of B
 * (n x nB, column-major, in place) — aggregation.jl:75,135-136 with the Hermitian fast path of smoother.jl:34-38. */
int amgs_improve_candidates(const amgs_mat* A, double* B, int nB, int iters);
void amgs_free(void*);

/* ---- drivers ------------------------------------------------------------ */
typedef struct {
  double theta;         /* strength threshold (RS: 0.25, SA: 0.0)             */
  int32_t max_levels;   /* 10                                                  */
  int32_t max_coarse;   /* 10                                                  */
  int32_t hermitian;    /* 1 = HermitianSymmetry() (default), 0 = NoSymmetry() */
  double sa_omega;      /* JacobiProlongation omega (4/3)                      */
  int32_t sa_improve_iters; /* improve_candidates = GaussSeidel(iter=4); 0=off */
  int32_t sa_B_is_vector;   /* 1: B::Vector path, 0: B::Matrix (QR) path      */
} amgs_options;

void amgs_default_options_rs(amgs_options*);
void amgs_default_options_sa(amgs_options*);

/* ruge_stuben(A; strength=Classical(theta), CF=RS(), max_levels, max_coarse) */
amgs_hier* amgs_ruge_stuben(const amgs_mat* A, const amgs_options*);
/* smoothed_aggregation(A; B, strength=SymmetricStrength(theta), ...)         */
/* B: n x nB column-major, or NULL for ones(n).                               */
amgs_hier* amgs_smoothed_aggregation(const amgs_mat* A, const double* B,
                                     int nB, const amgs_options*);
void amgs_hier_free(amgs_hier*);
int amgs_hier_num_levels(const amgs_hier*);  /* length(ml.levels)             */
/* which: 0 = A (CSC as the reference holds it), 1 = P (n x nc), 2 = R (nc x n)
 * For level == num_levels and which == 0: final_A.  Borrowed pointers.       */
const amgs_mat* amgs_hier_get(const amgs_hier*, int level, int which);

#ifdef __cplusplus
}
#endif
#endif
