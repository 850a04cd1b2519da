/*
 * amg_oracle.c — CPU ORACLE for the AMG solve phase.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the reference's solve-phase
 * algorithm, used (a) by tests/ as the checker for the HIP path, (b) by
 * __graft_entry__.smoke(), (c) by bench.py's `cpu_baseline` leg.  Nothing in the
 * product path (algebraicmultigrid.jl_amd/) may import, link or call it.
 *
 * Parity status: PINNED.  The reference itself cannot run here (pure Julia, no
 * Julia toolchain on either machine), so the oracle is pinned against the
 * known-answer vectors of the reference's own tests (tests/test_oracle_goldens.py;
 * SURVEY.md §8c): Gauss-Seidel hand values (test/sa_tests.jl:316-379), issue #26
 * (test/test_regression.jl:14-23), the five 46-entry V-cycle / PCG solution vectors
 * on thing.jl (test/runtests.jl:143-224), V/W/F convergence (test/cycle_tests.jl),
 * lin_elastic_2d (test/nns_test.jl:213-234), and more.
 *
 * Matrices are CSC (colptr/rowval/nzval, 0-based int32, f64), exactly as the
 * reference holds them; each function cites the reference lines it follows
 * (paths relative to /root/reference).  Compile with -ffp-contract=off so the
 * arithmetic is the un-fused IEEE sequence Julia executes.
 *
 * Third-party arithmetic restated here because its source is not under
 * /root/reference (Project.toml pins only julia>=1.6; no Manifest):
 *   SparseArrays stdlib  mul!(y,A,x) (CSC scatter) and mul!(y,A',x) (per-column dot)
 *   IterativeSolvers.jl  cg(A,b;Pl) — PCGIterable recurrence (src/cg.jl)
 *   LinearAlgebra        norm (2-norm), dense pinv/qr are supplied by the caller
 *                        as a dense operator (numpy/LAPACK), see orc_set_coarse.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* The element type of matrices and vectors.  The reference's code is generic in it; its tests run Float64 and
 * Float32 (test/runtests.jl:244-259).  liboracle.so is the Float64 restatement, liboracle_f32.so the same file
 * compiled with -DORC_REAL=float: every sum, product and division below then rounds to Float32 exactly where the
 * Julia code instantiated for Float32 does (relaxation factors and tolerances stay Float64, as the reference's
 * smoother fields and keyword defaults are).                                                                    */
#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real_t;

typedef struct {
  int64_t m, n;
  const int32_t* colptr;
  const int32_t* rowval;
  const real_t* nzval;
} csc_t;

typedef struct {
  int32_t kind;  /* 0 none, 1 GaussSeidel, 2 Jacobi, 3 SOR */
  int32_t sweep; /* 0 forward, 1 backward, 2 symmetric */
  int32_t iter;
  int32_t pad_;
  double omega;
} orc_smoother_t;

/* ---- SparseArrays mul! --------------------------------------------------- */
/* y = A*x, CSC: y[rowval[j]] += nzval[j]*x[col], columns ascending            */
void orc_spmv(const csc_t* A, const real_t* x, real_t* y) {
  for (int64_t i = 0; i < A->m; ++i) y[i] = 0.0;
  for (int64_t c = 0; c < A->n; ++c) {
    const real_t xc = x[c];
    for (int32_t j = A->colptr[c]; j < A->colptr[c + 1]; ++j) y[A->rowval[j]] += A->nzval[j] * xc;
  }
}
/* y = A'*x, CSC of A: y[col] = sum_j nzval[j]*x[rowval[j]]                     */
void orc_spmv_adj(const csc_t* A, const real_t* x, real_t* y) {
  for (int64_t c = 0; c < A->n; ++c) {
    real_t t = 0.0;
    for (int32_t j = A->colptr[c]; j < A->colptr[c + 1]; ++j) t += A->nzval[j] * x[A->rowval[j]];
    y[c] = t;
  }
}
real_t orc_norm2(const real_t* x, int64_t n) {
  real_t s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += x[i] * x[i];
  return sqrt(s);
}
static real_t dot(const real_t* x, const real_t* y, int64_t n) {
  real_t s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}

/* ---- smoother.jl:61-90  gs!(A, b, x, start, step, stop) -------------------- */
void orc_gs(const csc_t* A, const real_t* b, real_t* x, int backward) {
  const int64_t n = A->m;
  for (int64_t t = 0; t < n; ++t) {
    const int64_t i = backward ? n - 1 - t : t;
    real_t rsum = 0.0, d = 0.0;
    for (int32_t j = A->colptr[i]; j < A->colptr[i + 1]; ++j) {
      const int32_t row = A->rowval[j];
      const real_t val = A->nzval[j];
      if (i == row) d = val; else rsum += val * x[row];
    }
    if (d != 0.0) x[i] = (b[i] - rsum) / d;
  }
}
/* ---- smoother.jl:193-221 sor_step! ---------------------------------------- */
void orc_sor(const csc_t* A, const real_t* b, real_t* x, double omega, int backward) {
  const int64_t n = A->m;
  for (int64_t t = 0; t < n; ++t) {
    const int64_t i = backward ? n - 1 - t : t;
    real_t rsum = 0.0, d = 0.0;
    for (int32_t j = A->colptr[i]; j < A->colptr[i + 1]; ++j) {
      const int32_t row = A->rowval[j];
      const real_t val = A->nzval[j];
      if (i == row) d = val; else rsum += val * x[row];
    }
    if (d != 0.0) x[i] = (1 - omega) * x[i] + (omega / d) * (b[i] - rsum);
  }
}
/* ---- smoother.jl:113-141 FastJacobiSmoother, one sweep --------------------- */
void orc_jacobi(const csc_t* A, const real_t* b, real_t* x, real_t* temp, double omega) {
  const int64_t n = A->m;
  for (int64_t i = 0; i < n; ++i) temp[i] = x[i];
  for (int64_t i = 0; i < n; ++i) {
    real_t rsum = 0.0, diag = 0.0;
    for (int32_t j = A->colptr[i]; j < A->colptr[i + 1]; ++j) {
      const int32_t row = A->rowval[j];
      const real_t val = A->nzval[j];
      if (row == i) diag = val; else rsum += val * temp[row];
    }
    const real_t xcand = (1.0 - omega) * temp[i] + omega * ((b[i] - rsum) / diag);
    if (diag != 0.0) x[i] = xcand;
  }
}

/* ---- NoSymmetry family, smoother.jl:144-171, 226-582 ----------------------- */
/* DiagonalIndices (smoother.jl:231-247): index of the diagonal in each column;
 * returns the 1-based offending column (SingularException) or 0.              */
int64_t orc_diag_indices(const csc_t* A, int32_t* diag) {
  for (int64_t col = 0; col < A->n; ++col) {
    int32_t r1 = A->colptr[col], r2 = A->colptr[col + 1] - 1;
    while (r1 <= r2 && A->rowval[r1] < col) ++r1; /* searchsortedfirst */
    if (r1 > r2 || A->rowval[r1] != col || A->nzval[r1] == 0.0) return col + 1;
    diag[col] = r1;
  }
  return 0;
}
/* z := alpha*U*x + beta*y, U strictly upper (smoother.jl:373-389); z may alias x */
static void gsm_upper(const csc_t* A, const int32_t* diag, double alpha, const real_t* x, double beta,
                      const real_t* y, real_t* z) {
  for (int64_t col = 0; col < A->n; ++col) {
    const real_t ax = alpha * x[col];
    for (int32_t j = A->colptr[col]; j <= diag[col] - 1; ++j) z[A->rowval[j]] += A->nzval[j] * ax;
    z[col] = beta * y[col];
  }
}
/* z := alpha*L*x + beta*y, L strictly lower (smoother.jl:395-408) */
static void gsm_lower(const csc_t* A, const int32_t* diag, double alpha, const real_t* x, double beta,
                      const real_t* y, real_t* z) {
  for (int64_t col = A->n - 1; col >= 0; --col) {
    const real_t ax = alpha * x[col];
    z[col] = beta * y[col];
    for (int32_t j = diag[col] + 1; j <= A->colptr[col + 1] - 1; ++j) z[A->rowval[j]] += A->nzval[j] * ax;
  }
}
/* forward_sub!(F, x) (smoother.jl:282-300) */
static void forward_sub(const csc_t* A, const int32_t* diag, real_t* x) {
  for (int64_t col = 0; col < A->n; ++col) {
    const int32_t idx = diag[col];
    x[col] /= A->nzval[idx];
    for (int32_t i = idx + 1; i <= A->colptr[col + 1] - 1; ++i) x[A->rowval[i]] -= A->nzval[i] * x[col];
  }
}
/* forward_sub!(alpha, F, x, beta, y) (smoother.jl:305-323) */
static void forward_sub_ab(const csc_t* A, const int32_t* diag, double alpha, real_t* x, double beta, const real_t* y) {
  for (int64_t col = 0; col < A->n; ++col) {
    const int32_t idx = diag[col];
    x[col] = alpha * x[col] / A->nzval[idx] + beta * y[col];
    for (int32_t i = idx + 1; i <= A->colptr[col + 1] - 1; ++i) x[A->rowval[i]] -= A->nzval[i] * x[col];
  }
}
/* backward_sub!(F, x) (smoother.jl:329-347) */
static void backward_sub(const csc_t* A, const int32_t* diag, real_t* x) {
  for (int64_t col = A->n - 1; col >= 0; --col) {
    const int32_t idx = diag[col];
    x[col] /= A->nzval[idx];
    for (int32_t i = A->colptr[col]; i <= idx - 1; ++i) x[A->rowval[i]] -= A->nzval[i] * x[col];
  }
}
static void backward_sub_ab(const csc_t* A, const int32_t* diag, double alpha, real_t* x, double beta, const real_t* y) {
  for (int64_t col = A->n - 1; col >= 0; --col) {
    const int32_t idx = diag[col];
    x[col] = alpha * x[col] / A->nzval[idx] + beta * y[col];
    for (int32_t i = A->colptr[col]; i <= idx - 1; ++i) x[A->rowval[i]] -= A->nzval[i] * x[col];
  }
}
/* GS / SOR smooth! of the NoSymmetry family (smoother.jl:410-582); A is the TRUE
 * matrix in CSC.  tmp: n doubles (SOR only).                                   */
static void nosym_gs_sor(const csc_t* A, const int32_t* diag, const orc_smoother_t* s, real_t* x, const real_t* b,
                         real_t* tmp) {
  const int64_t n = A->n;
  for (int it = 0; it < s->iter; ++it) {
    if (s->kind == 1) {
      if (s->sweep == 0 || s->sweep == 2) { gsm_upper(A, diag, -1.0, x, 1.0, b, x); forward_sub(A, diag, x); }
      if (s->sweep == 1 || s->sweep == 2) { gsm_lower(A, diag, -1.0, x, 1.0, b, x); backward_sub(A, diag, x); }
    } else {
      /* tmp is NOT cleared between uses (zeros at setup, then copy! keeps it = x):
       * gauss_seidel_multiply! accumulates into tmp rows before overwriting them,
       * and tmp == x on entry after the first copy!.  At setup tmp = zeros.    */
      if (s->sweep == 0 || s->sweep == 2) {
        gsm_upper(A, diag, -1.0, x, 1.0, b, tmp);
        forward_sub_ab(A, diag, s->omega, tmp, 1.0 - s->omega, x);
        memcpy(x, tmp, sizeof(real_t) * n);
      }
      if (s->sweep == 1 || s->sweep == 2) {
        gsm_lower(A, diag, -1.0, x, 1.0, b, tmp);
        backward_sub_ab(A, diag, s->omega, tmp, 1.0 - s->omega, x);
        memcpy(x, tmp, sizeof(real_t) * n);
      }
    }
  }
}
/* JacobiSmoother (NoSymmetry), smoother.jl:157-171 */
static void nosym_jacobi(const csc_t* A, const orc_smoother_t* s, real_t* x, const real_t* b, real_t* temp) {
  const int64_t n = A->n;
  real_t* dv = (real_t*)calloc(n ? n : 1, sizeof(real_t));
  for (int64_t c = 0; c < n; ++c)
    for (int32_t j = A->colptr[c]; j < A->colptr[c + 1]; ++j)
      if (A->rowval[j] == c) dv[c] += A->nzval[j];
  for (int it = 0; it < s->iter; ++it) {
    orc_spmv(A, x, temp);
    for (int64_t i = 0; i < n; ++i) temp[i] -= b[i];
    for (int64_t i = 0; i < n; ++i)
      if (dv[i] != 0.0) x[i] -= s->omega * temp[i] / dv[i];
  }
  free(dv);
}

/* smooth!(x, setup_smoother(config, A, symmetry), b).  Returns 0, or the 1-based
 * column of a SingularException for the NoSymmetry GS/SOR family.              */
int64_t orc_smooth(const csc_t* A, const orc_smoother_t* s, int hermitian, real_t* x, const real_t* b) {
  const int64_t n = A->m;
  if (s->kind == 0 || s->iter <= 0) return 0;
  if (hermitian) {
    if (s->kind == 2) {
      real_t* temp = (real_t*)malloc(sizeof(real_t) * (n ? n : 1));
      for (int it = 0; it < s->iter; ++it) orc_jacobi(A, b, x, temp, s->omega);
      free(temp);
    } else {
      for (int it = 0; it < s->iter; ++it) {
        if (s->sweep == 0 || s->sweep == 2) { if (s->kind == 1) orc_gs(A, b, x, 0); else orc_sor(A, b, x, s->omega, 0); }
        if (s->sweep == 1 || s->sweep == 2) { if (s->kind == 1) orc_gs(A, b, x, 1); else orc_sor(A, b, x, s->omega, 1); }
      }
    }
    return 0;
  }
  real_t* tmp = (real_t*)calloc(n ? n : 1, sizeof(real_t));
  int64_t rc = 0;
  if (s->kind == 2) {
    nosym_jacobi(A, s, x, b, tmp);
  } else {
    int32_t* diag = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    rc = orc_diag_indices(A, diag);
    if (rc == 0) nosym_gs_sor(A, diag, s, x, b, tmp);
    free(diag);
  }
  free(tmp);
  return rc;
}

/* ---- hierarchy (multilevel.jl:1-59) ---------------------------------------- */
typedef struct {
  csc_t A;       /* n x n */
  csc_t M;       /* RS: R (nc x n) with P = R' lazy (classical.jl:64-65);
                    SA: P (n x nc) with R = P' lazy (aggregation.jl:89,158)    */
  int m_is_R;
  orc_smoother_t pre, post;
  int hermitian;
  real_t *res, *cx, *cb;
} orc_level;

typedef struct {
  int nlev, cap;
  orc_level* lev;
  csc_t finalA;
  int64_t ncoarse;
  const real_t* coarse_op; /* ncoarse x ncoarse column-major, or NULL */
  int (*coarse_fn)(void* user, const real_t* b, real_t* x, int64_t n); /* pluggable (cs)(x,b) */
  void* coarse_user;
  real_t* res_final;
} orc_hier;

orc_hier* orc_create(void) { return (orc_hier*)calloc(1, sizeof(orc_hier)); }

void orc_destroy(orc_hier* h) {
  if (!h) return;
  for (int l = 0; l < h->nlev; ++l) { free(h->lev[l].res); free(h->lev[l].cx); free(h->lev[l].cb); }
  free(h->lev); free(h->res_final); free(h);
}

/* arrays are BORROWED (must outlive the hierarchy) */
int orc_push_level(orc_hier* h, int64_t n, int64_t nc, const int32_t* Ap, const int32_t* Ai, const real_t* Ax,
                   int m_is_R, const int32_t* Mp, const int32_t* Mi, const real_t* Mx, const orc_smoother_t* pre,
                   const orc_smoother_t* post, int hermitian) {
  if (h->nlev == h->cap) {
    h->cap = h->cap ? 2 * h->cap : 8;
    h->lev = (orc_level*)realloc(h->lev, sizeof(orc_level) * h->cap);
  }
  orc_level* L = &h->lev[h->nlev++];
  memset(L, 0, sizeof *L);
  L->A = (csc_t){n, n, Ap, Ai, Ax};
  L->M = m_is_R ? (csc_t){nc, n, Mp, Mi, Mx} : (csc_t){n, nc, Mp, Mi, Mx};
  L->m_is_R = m_is_R;
  L->pre = *pre; L->post = *post; L->hermitian = hermitian;
  L->res = (real_t*)malloc(sizeof(real_t) * (n ? n : 1));
  L->cx = (real_t*)malloc(sizeof(real_t) * (nc ? nc : 1));
  L->cb = (real_t*)malloc(sizeof(real_t) * (nc ? nc : 1));
  return 0;
}
int orc_set_coarse(orc_hier* h, int64_t n, const int32_t* Ap, const int32_t* Ai, const real_t* Ax, const real_t* op) {
  h->finalA = (csc_t){n, n, Ap, Ai, Ax};
  h->ncoarse = n;
  h->coarse_op = op;
  h->res_final = (real_t*)malloc(sizeof(real_t) * (n ? n : 1));
  return 0;
}

int orc_set_coarse_fn(orc_hier* h, int (*fn)(void*, const real_t*, real_t*, int64_t), void* user) {
  h->coarse_fn = fn; h->coarse_user = user;
  return 0;
}

/* (cs)(x, b): Pinv -> mul!(x, pinvA, b) (coarse_solver.jl:16); QRSolver ->
 * factorization \ b (coarse_solver.jl:75-81) — both as x = op*b.               */
static void coarse_solve(const orc_hier* h, real_t* x, const real_t* b) {
  const int64_t n = h->ncoarse;
  if (h->coarse_fn) { h->coarse_fn(h->coarse_user, b, x, n); return; }
  for (int64_t i = 0; i < n; ++i) {
    real_t acc = 0.0;
    for (int64_t j = 0; j < n; ++j) acc += h->coarse_op[i + j * n] * b[j];
    x[i] = acc;
  }
}

static void cycle(orc_hier* h, int l, real_t* x, const real_t* b, int cyc);
/* __solve_next! (multilevel.jl:200-212) */
static void cycle_next(orc_hier* h, int l, real_t* x, const real_t* b, int cyc) {
  if (cyc == 0) { cycle(h, l, x, b, 0); }
  else if (cyc == 1) { cycle(h, l, x, b, 1); cycle(h, l, x, b, 1); }
  else { cycle(h, l, x, b, 2); cycle(h, l, x, b, 0); }
}
/* __solve! (multilevel.jl:214-239) */
static void cycle(orc_hier* h, int l, real_t* x, const real_t* b, int cyc) {
  orc_level* L = &h->lev[l];
  const int64_t n = L->A.m;
  const int64_t nc = L->m_is_R ? L->M.m : L->M.n;
  orc_smooth(&L->A, &L->pre, L->hermitian, x, b);
  orc_spmv(&L->A, x, L->res);
  for (int64_t i = 0; i < n; ++i) L->res[i] = b[i] - L->res[i];
  if (L->m_is_R) orc_spmv(&L->M, L->res, L->cb); else orc_spmv_adj(&L->M, L->res, L->cb);
  for (int64_t i = 0; i < nc; ++i) L->cx[i] = 0.0;
  if (l == h->nlev - 1) coarse_solve(h, L->cx, L->cb);
  else cycle_next(h, l + 1, L->cx, L->cb, cyc);
  if (L->m_is_R) orc_spmv_adj(&L->M, L->cx, L->res); else orc_spmv(&L->M, L->cx, L->res);
  for (int64_t i = 0; i < n; ++i) x[i] += L->res[i];
  orc_smooth(&L->A, &L->post, L->hermitian, x, b);
}

/* _solve!(x, ml, b, cycle; maxiter, abstol, reltol, log, calculate_residual)
 * (multilevel.jl:158-198).  hist: NULL or maxiter+1.                          */
int orc_solve(orc_hier* h, const real_t* b, real_t* x, int cyc, int maxiter, double abstol, double reltol,
              int calc_res, real_t* hist, int* iters) {
  const csc_t* A = h->nlev ? &h->lev[0].A : &h->finalA;
  const int64_t n = A->m;
  real_t normb = orc_norm2(b, n), normres = normb;
  if (normb != 0) abstol = fmax(reltol * normb, abstol);
  if (hist) hist[0] = normb;
  real_t* res = h->nlev ? h->lev[0].res : h->res_final;
  int itr = 1;
  while (itr <= maxiter && (!calc_res || normres > abstol)) {
    if (h->nlev == 0) coarse_solve(h, x, b); else cycle(h, 0, x, b, cyc);
    if (calc_res) {
      orc_spmv(A, x, res);
      for (int64_t i = 0; i < n; ++i) res[i] = b[i] - res[i];
      normres = orc_norm2(res, n);
      if (hist) hist[itr] = normres;
    }
    ++itr;
  }
  if (iters) *iters = itr - 1;
  return 0;
}

/* ldiv!(x, p, b) (preconditioner.jl:12-19) */
void orc_precond(orc_hier* h, const real_t* b, real_t* x, int cyc) {
  const int64_t n = h->nlev ? h->lev[0].A.m : h->finalA.m;
  for (int64_t i = 0; i < n; ++i) x[i] = 0.0;
  orc_solve(h, b, x, cyc, 1, 0.0, 0.0, 0, NULL, NULL);
}

/* IterativeSolvers.jl cg(A, b; Pl = p, abstol, reltol, maxiter): PCGIterable.  x0 = 0. */
int orc_pcg(orc_hier* h, const real_t* b, real_t* x, int cyc, int use_precond, int maxiter, double abstol,
            double reltol, real_t* hist, int* iters) {
  const csc_t* A = h->nlev ? &h->lev[0].A : &h->finalA;
  const int64_t n = A->m;
  real_t* r = (real_t*)malloc(sizeof(real_t) * n);
  real_t* c = (real_t*)malloc(sizeof(real_t) * n);
  real_t* u = (real_t*)calloc(n, sizeof(real_t));
  for (int64_t i = 0; i < n; ++i) { x[i] = 0.0; r[i] = b[i]; }
  real_t residual = orc_norm2(r, n);
  const double tol = fmax(reltol * residual, abstol);
  if (hist) hist[0] = residual;
  real_t rho = 1.0;
  int it = 0;
  while (it < maxiter && residual > tol) {
    if (use_precond) orc_precond(h, r, c, cyc); else memcpy(c, r, sizeof(real_t) * n);
    const real_t rho_prev = rho;
    rho = dot(c, r, n);
    const real_t beta = rho / rho_prev;
    for (int64_t i = 0; i < n; ++i) u[i] = c[i] + beta * u[i];
    orc_spmv(A, u, c);
    const real_t alpha = rho / dot(u, c, n);
    for (int64_t i = 0; i < n; ++i) x[i] += alpha * u[i];
    for (int64_t i = 0; i < n; ++i) r[i] -= alpha * c[i];
    residual = orc_norm2(r, n);
    ++it;
    if (hist) hist[it] = residual;
  }
  if (iters) *iters = it;
  free(r); free(c); free(u);
  return 0;
}

/* helpers for ctypes callers that hold raw arrays */
void orc_spmv_arrays(int64_t m, int64_t n, const int32_t* p, const int32_t* i, const real_t* v, const real_t* x,
                     real_t* y, int adjoint) {
  csc_t A = {m, n, p, i, v};
  if (adjoint) orc_spmv_adj(&A, x, y); else orc_spmv(&A, x, y);
}
int64_t orc_smooth_arrays(int64_t n, const int32_t* p, const int32_t* i, const real_t* v, const orc_smoother_t* s,
                          int hermitian, real_t* x, const real_t* b) {
  csc_t A = {n, n, p, i, v};
  return orc_smooth(&A, s, hermitian, x, b);
}
