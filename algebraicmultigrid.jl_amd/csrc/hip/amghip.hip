// amghip.hip — host logic + C ABI of libamghip (see include/amghip.h).
//
// The hierarchy (A, S, P, R per level, CSR, int32/f64) lives in HBM; one HIP
// stream per handle; the cycle of multilevel.jl:214-239 is a host-driven
// sequence of kernel launches with no host<->device traffic inside it (only the
// residual norm of multilevel.jl:190 comes back, once per outer iteration).
//
// Gauss-Seidel (smoother.jl:61-90) is executed in EXACT lexicographic order by
// dependency-level scheduling: rows are grouped by the length of their longest
// dependency chain in the symmetrised pattern, a level-permuted copy of the
// matrix keeps each group contiguous, wide groups are one coalesced stream-kernel
// launch each, runs of narrow groups are chained inside one workgroup.  The
// backward sweep walks the same groups in reverse.
#include "amghip_internal.hpp"
#include "gs_schedule.hpp"
#include "csr_ops.hpp"

namespace {

constexpr int64_t kGraphAutoRows = 65536;   // hierarchies whose widest level is at most this replay their cycle from a hipGraph

struct Level {
  int64_t n = 0, nc = 0;
  amgh_csr A, S, P, R;
  bool has_S = false;  // S distinct from A
  amgh_smoother_t pre{}, post{};
  real *res = nullptr, *cx = nullptr, *cb = nullptr, *tmp = nullptr;
  real* il = nullptr;   // blocks of right-hand sides: interleaved copy of what R / P gather (max(n, nc) x bs)
  amgh_csr* smat() { return has_S ? &S : &A; }
  // level-ordered cycle (x stays in the smoother's dependency-level order between pre- and post-smoother):
  // P with its rows in that order, R with its columns renumbered to it; A in that order is the schedule's own copy
  amgh_csr Pp, Rp;
  bool lo_ok = false;
  real* lo_val = nullptr;  // S != A with the same pattern: values of A in the order of the schedule's level-ordered copy of S
  CodedCols lo_cc;         // value-coded columns of the level-ordered A (the schedule's copy with lo_val or its own values)
  // blocks of right-hand sides: restriction / prolongation of this level column by column through the stream kernel instead of
  // the interleaved copy (chosen by a timing at amgh_finalize where the operator has value-coded columns: bitwise the same sums)
  bool r_stream = false, p_stream = false;
  bool lo_want = false;    // between amgh_push_level_begin and _end: the level-ordered P / R are to be built
  // the COARSE side in the next level's dependency-level order too (set when the next level is pushed and runs the
  // level-ordered cycle): Rp's rows and Pp's columns are renumbered to it, the restricted residual is written straight
  // into the next level's level-ordered right-hand side and the correction read from its level-ordered x — no gather
  // of b / scatter of x on the next level, and the gathers of R and P stay local (a coarse hyperplane's fine
  // neighbours sit in adjacent fine hyperplanes)
  bool coarse_lo = false;
  bool nat_freed = false;  // memory-lean: the natural-order A (levels >= 1), P and R were released, the cycle runs level-ordered
};

}  // namespace

struct amgh_handle {
  int device = 0;
  int nrhs = 1;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;  // the stream created by amgh_create (when an external one is in use)
  bool ext_stream = false;
  std::vector<Level*> levels;
  Level* pending_level = nullptr;  // between amgh_push_level_begin and amgh_push_level_end
  // coarsest
  int64_t ncoarse = -1;
  amgh_csr finalA;
  bool has_finalA = false;
  real* coarse_op = nullptr;
  amgh_coarse_fn coarse_fn = nullptr;  // host-side pluggable coarse solver
  void* coarse_user = nullptr;
  std::vector<real> coarse_hb, coarse_hx;
  real* res_final = nullptr;  // res_vecs[1] when there are no levels
  bool finalized = false;
  // reductions / scalars
  real* partial = nullptr;   // kRedBlocks
  real* scal = nullptr;      // device scalars: [0] norm/dot out, [1] rho, [2] rho_prev, [3] alpha, [4] beta, [5] tmp
  // internal fine-level buffers for host-pointer entry points and PCG
  real *x0 = nullptr, *b0 = nullptr, *pc_r = nullptr, *pc_c = nullptr, *pc_u = nullptr;
  int64_t ws_bytes = 0;
  // profiling
  bool profile = false;
  std::vector<double> prof_ms;
  struct Ev { hipEvent_t a, b; int label, level; };
  std::vector<Ev> pending;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // Measured on MI355X (profiles/r01_vcycle_profile.log): replaying a 256^3 V-cycle (~11k kernel
  // nodes) from a hipGraph takes 180.1 ms vs 179.6 ms eager — the cycle is bound by the GPU-side
  // latency of each dependency-level kernel, not by host launches — and rocprofv3's kernel tracing
  // aborts on the replay (malformed AQL packet).  So graphs are opt-in (amgh_set_use_graph or
  // AMGH_USE_GRAPH=1).
  bool use_graph = false;
  bool graph_auto = true;   // nobody (environment, amgh_set_use_graph) has decided: amgh_finalize does, by the hierarchy's size
  // hipGraph cache of whole cycles, keyed by the (x, b, cycle) they were captured on
  // exec == nullptr: this key has been run eagerly once (first-use allocations done), capture on the next call
  struct CycleGraph { const real* x; const real* b; int cyc; bool xzero; hipGraphExec_t exec; };
  std::vector<CycleGraph> graphs;
  unsigned long long graph_epoch = 0;  // g_sched_epoch the cached graphs were captured under
  // The collapsed coarse tail: from level tail_level down the recursion of __solve! (multilevel.jl:214-239) is ONE dense operator
  // per cycle type — M[cyc]: row-major n x n, between the vectors the level above hands over (natural order, or the level's
  // schedule order where the level above restricts into / prolongs from it).  level: -1 = not decided, -2 = none.
  struct TailDense { real* M = nullptr; bool built = false; };
  int tail_level = -1;
  int tail_cols = 0;        // columns the workspaces of the tail's levels hold (>= nrhs: the build runs blocks of right-hand sides)
  bool tail_lo = false;     // the operator maps the level-ordered right-hand side to the level-ordered x
  bool tail_building = false;
  TailDense tail[3];
  double tail_build_ms = 0.0;
};

namespace {

int64_t fine_n(const amgh_t* h) { return h->levels.empty() ? h->ncoarse : h->levels[0]->n; }

struct ProfScope {
  amgh_t* h; int label, level; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(amgh_t* h_, int label_, int level_) : h(h_), label(label_), level(level_) {
    if (h->profile) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, h->stream); }
  }
  ~ProfScope() {
    if (h->profile) { hipEventRecord(b, h->stream); h->pending.push_back({a, b, label, level}); }
  }
};

int prof_flush(amgh_t* h) {
  if (h->pending.empty()) return AMGH_OK;
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int L1 = (int)h->levels.size() + 1;
  if ((int)h->prof_ms.size() != AMGH_T_COUNT * L1) h->prof_ms.assign(AMGH_T_COUNT * L1, 0.0);
  for (auto& e : h->pending) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, e.a, e.b);
    h->prof_ms[e.label * L1 + e.level] += ms;
    hipEventDestroy(e.a); hipEventDestroy(e.b);
  }
  h->pending.clear();
  return AMGH_OK;
}

int vec_fill(amgh_t* h, real* x, int64_t n, real v) {
  if (n <= 0) return AMGH_OK;
  if (v == 0.0) { HIP_TRY(hipMemsetAsync(x, 0, sizeof(real) * n, h->stream)); return AMGH_OK; }
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, x, n, v);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
int vec_copy(amgh_t* h, real* dst, const real* src, int64_t n) {
  if (n <= 0 || dst == src) return AMGH_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, sizeof(real) * n, hipMemcpyDeviceToDevice, h->stream));
  return AMGH_OK;
}
// out_d[0] = sum x*y (op 0) or sqrt(sum x*x) (op 1)
int vec_dot(amgh_t* h, const real* x, const real* y, int64_t n, real* out_d, int op) {
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kRedBlocks, (n + kThreads - 1) / kThreads));
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(kThreads), 0, h->stream, x, y, n, h->partial);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, h->stream, h->partial, nb, out_d, op);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
int vec_norm_host(amgh_t* h, const real* x, int64_t n, real* out) {
  RC_TRY(vec_dot(h, x, x, n, h->scal, 1));
  HIP_TRY(hipMemcpyAsync(out, h->scal, sizeof(real), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return AMGH_OK;
}

int coarse_solve(amgh_t* h, real* x, const real* b) {
  const int n = (int)h->ncoarse;
  if (n <= 0) return AMGH_OK;
  if (h->coarse_fn) {
    HIP_TRY(hipMemcpyAsync(h->coarse_hb.data(), b, sizeof(real) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->coarse_fn(h->coarse_user, h->coarse_hb.data(), h->coarse_hx.data(), n) != 0) return AMGH_ESTATE;
    HIP_TRY(hipMemcpyAsync(x, h->coarse_hx.data(), sizeof(real) * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return AMGH_OK;
  }
  hipLaunchKernelGGL(dense_gemv_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->coarse_op, b, x, n);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}

// smooth!(x, smoother, b).  `xc` is the buffer that currently holds x; Jacobi
// sweeps ping-pong between xc and xo (swapped in place).
// x_resident / no_scatter: see csr_gs_sweep (the level-ordered cycle keeps x in the schedule's own vector)
int smooth(amgh_t* h, Level* L, const amgh_smoother_t& s, real*& xc, real*& xo, const real* b, int ncolv = 1,
           bool xzero = false, bool reuse_b = false, bool x_resident = false, bool no_scatter = false) {
  amgh_csr* M = L->smat();
  for (int it = 0; it < s.iter; ++it) {
    switch (s.kind) {
      case AMGH_SMOOTH_NONE: break;
      case AMGH_SMOOTH_JACOBI:
        if (xzero && it == 0 && g_jacobi_zero) RC_TRY(csr_jacobi_zero(M, s.omega, b, xo, h->stream, ncolv));   // x = 0: no matrix pass
        else RC_TRY(csr_jacobi(M, s.omega, xc, b, xo, h->stream, ncolv));
        std::swap(xc, xo);
        break;
      case AMGH_SMOOTH_GS:
      case AMGH_SMOOTH_SOR: {
        const bool sor = s.kind == AMGH_SMOOTH_SOR;
        // x and b enter dependency-level order before the first sweep of this smooth! call and x
        // returns to natural order after the last one
        const bool sym = s.sweep == AMGH_SWEEP_SYMMETRIC;
        const bool first_it = (it == 0), last_it = (it == s.iter - 1);
        if (sym && M->gs && gs_wave_path(M->gs, sor)) {   // a small operator: forward and backward in ONE single-wave launch
          RC_TRY(csr_gs_sweep(M, false, sor, s.omega, xc, b, h->stream, first_it, last_it, ncolv, xzero && first_it,
                              reuse_b || !first_it, x_resident, no_scatter, true));
          break;
        }
        if (s.sweep == AMGH_SWEEP_FORWARD || sym)
          RC_TRY(csr_gs_sweep(M, false, sor, s.omega, xc, b, h->stream, first_it, last_it && !sym, ncolv, xzero && first_it,
                              reuse_b || !first_it, x_resident, no_scatter));
        if (s.sweep == AMGH_SWEEP_BACKWARD || sym)
          RC_TRY(csr_gs_sweep(M, true, sor, s.omega, xc, b, h->stream, first_it && !sym, last_it, ncolv,
                              xzero && first_it && !sym, reuse_b || !first_it, x_resident, no_scatter));
        break;
      }
      default: return AMGH_EINVAL;
    }
  }
  return AMGH_OK;
}

int cycle(amgh_t* h, int l, real* x, const real* b, int cyc, bool xzero, bool lo_io = false);

// ---- the collapsed coarse tail (see amghip_kernels.hpp: dense_rm_gemv_kernel) ----
// is level l entered through its dense operator in a cycle of type cyc?
inline bool tail_here(const amgh_t* h, int l, int cyc) {
  return g_tail_dense && !h->tail_building && h->tail_level == l && cyc >= 0 && cyc < 3 && h->tail[cyc].built;
}
template <int NC, bool ACC>
int tail_launch(const real* M, int64_t n, real* x, int64_t ldx, const real* b, int64_t ldb, int groups, hipStream_t st) {
  // (the thread count is a function of n alone: a column's sum is formed the same way whatever the block of right-hand sides)
  if (n <= 1024) hipLaunchKernelGGL((dense_rm_gemv_kernel<NC, 64, ACC>), dim3((unsigned)n, groups), dim3(64), 0, st, M, b, x, (int)n, ldb, ldx);
  else hipLaunchKernelGGL((dense_rm_gemv_kernel<NC, 256, ACC>), dim3((unsigned)n, groups), dim3(256), 0, st, M, b, x, (int)n, ldb, ldx);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
// x = M b (acc: x += M b) for ncolv columns; the row of M is read once for 8 / 4 / 2 / 1 columns
template <bool ACC>
int tail_apply_t(amgh_t* h, const real* M, int64_t n, real* x, int64_t ldx, const real* b, int64_t ldb, int ncolv) {
  if (ncolv % 8 == 0) return tail_launch<8, ACC>(M, n, x, ldx, b, ldb, ncolv / 8, h->stream);
  if (ncolv % 4 == 0) return tail_launch<4, ACC>(M, n, x, ldx, b, ldb, ncolv / 4, h->stream);
  if (ncolv % 2 == 0) return tail_launch<2, ACC>(M, n, x, ldx, b, ldb, ncolv / 2, h->stream);
  return tail_launch<1, ACC>(M, n, x, ldx, b, ldb, ncolv, h->stream);
}
int tail_apply(amgh_t* h, const real* M, int64_t n, real* x, int64_t ldx, const real* b, int64_t ldb, int ncolv, bool acc) {
  return acc ? tail_apply_t<true>(h, M, n, x, ldx, b, ldb, ncolv) : tail_apply_t<false>(h, M, n, x, ldx, b, ldb, ncolv);
}

// __solve_next! (multilevel.jl:200-212)
// x is the parent's freshly zeroed coarse_x (multilevel.jl:226) on the first visit, not on the second (W, F)
// lo_io: b is already in this level's level-ordered right-hand side (the level above restricted into it) and x is to
// stay in its level-ordered vector (the level above prolongs from there)
int cycle_next(amgh_t* h, int l, real* x, const real* b, int cyc, bool lo_io = false) {
  switch (cyc) {
    case AMGH_CYCLE_V: return cycle(h, l, x, b, AMGH_CYCLE_V, true, lo_io);
    case AMGH_CYCLE_W:
      RC_TRY(cycle(h, l, x, b, AMGH_CYCLE_W, true, lo_io));
      return cycle(h, l, x, b, AMGH_CYCLE_W, false, lo_io);
    case AMGH_CYCLE_F:
      RC_TRY(cycle(h, l, x, b, AMGH_CYCLE_F, true, lo_io));
      return cycle(h, l, x, b, AMGH_CYCLE_V, false, lo_io);
  }
  return AMGH_EINVAL;
}

// __solve! (multilevel.jl:214-239).  x, b: n x bs column-major (bs = workspace block size,
// multilevel.jl:28-59).  The reference loops the columns inside every operator (smoother.jl:77,117); here
// each launch covers all bs columns (one grid row / one workgroup per column): per column the arithmetic and
// its order are those of the single-column path, so the results are bitwise the same.
int cycle(amgh_t* h, int l, real* x, const real* b, int cyc, bool xzero, bool lo_io) {
  Level* L = h->levels[l];
  const int bs = h->nrhs;
  const int64_t n = L->n, nc = L->nc;
  real* xc = x;
  real* xo = L->tmp;
  // Level-ordered variant: between the two smoothers x stays in the smoother's dependency-level order (its own xp),
  // and residual / restriction / prolongation run on level-ordered copies of A, R, P (same entries in the same
  // order inside every row, so the same sums): no scatter after the pre-smoother, no gather before the post-smoother.
  GsSchedule* g = L->smat()->gs;
  const bool lo = L->lo_ok && (g_gs_keep_lo || L->nat_freed || lo_io) && g && g->nblk == 0;
  if (lo_io && !lo) return AMGH_ESTATE;   // (cannot happen: the level above only hands over in level order what was built for it)
  {
    ProfScope p(h, AMGH_T_PRESMOOTH, l);
    RC_TRY(smooth(h, L, L->pre, xc, xo, b, bs, xzero, /*reuse_b*/ lo_io, /*x_resident*/ lo_io && !xzero, /*no_scatter*/ lo));
  }
  if (lo) {
    // the next level takes its vectors in its own level order (Rp's rows / Pp's columns were renumbered when it was pushed)
    Level* C = (L->coarse_lo && l + 1 < (int)h->levels.size()) ? h->levels[l + 1] : nullptr;
    GsSchedule* cg = C ? C->smat()->gs : nullptr;
    if (C && !(cg && C->lo_ok && cg->nblk == 0 && cg->cols_alloc >= bs)) return AMGH_ESTATE;
    const int64_t xs = g->xstride;
    {
      ProfScope p(h, AMGH_T_RESIDUAL, l);  // r = b - A x, everything in level order (bp was gathered by the pre-smoother)
      // (A itself in level order: the schedule's copy, or — when the smoother sweeps S = A' — A's values on S's pattern)
      if (il_block(bs) && g_rhs_il >= 2)   // (2: the block residual reads the matrix once; measured)
        RC_TRY(resid_cols(bs, g->rowptr, g->col, L->lo_val ? L->lo_val : g->val, n, g->xp, xs, g->bp, n, L->res, n, h->stream, &L->lo_cc));
      else
        RC_TRY(raw_apply(M_RESID, g->rowptr, g->col, L->lo_val ? L->lo_val : g->val, n, g->xp, xs, g->bp, n, L->res, n,
                         h->stream, bs, &L->lo_cc));
    }
    {
      ProfScope p(h, AMGH_T_RESTRICT, l);
      if (L->il && il_block(bs) && !(L->r_stream && g_rhs_il == 1))
        RC_TRY(il_apply(bs, false, L->Rp.rowptr, L->Rp.col, L->Rp.val, nc, L->res, n, n, L->il, C ? cg->bp : L->cb, nc, h->stream, &L->Rp.cc));
      else
        RC_TRY(csr_apply(&L->Rp, M_SPMV, L->res, nullptr, C ? cg->bp : L->cb, h->stream, bs));
      if (C) cg->bp_cols = bs;          // the next level's level-ordered right-hand side is in place
    }
    if (!C) RC_TRY(vec_fill(h, L->cx, nc * bs, 0.0));
    if (l == (int)h->levels.size() - 1) {
      ProfScope p(h, AMGH_T_COARSE, l + 1);
      for (int c = 0; c < bs; ++c) RC_TRY(coarse_solve(h, L->cx + c * nc, L->cb + c * nc));
    } else if (tail_here(h, l + 1, cyc) && (C != nullptr) == h->tail_lo) {
      // the levels below as ONE dense operator (built from this very recursion, tail_dense_build): x_c = M b_c
      ProfScope p(h, AMGH_T_COARSE, l + 1);
      RC_TRY(tail_apply(h, h->tail[cyc].M, nc, C ? cg->xp : L->cx, C ? cg->xstride : nc, C ? (const real*)cg->bp : (const real*)L->cb, nc, bs, false));
    } else {
      RC_TRY(cycle_next(h, l + 1, L->cx, L->cb, cyc, C != nullptr));
    }
    {
      ProfScope p(h, AMGH_T_PROLONG, l);  // x += P e on the level-ordered x
      if (L->il && il_block(bs) && !(L->p_stream && g_rhs_il == 1))
        RC_TRY(il_apply(bs, true, L->Pp.rowptr, L->Pp.col, L->Pp.val, n, C ? cg->xp : L->cx, nc, C ? cg->xstride : nc, L->il,
                        g->xp, xs, h->stream, &L->Pp.cc));
      else
        RC_TRY(raw_apply(M_ADD, L->Pp.rowptr, L->Pp.col, L->Pp.val, n, C ? cg->xp : L->cx, C ? cg->xstride : nc, nullptr, 0, g->xp,
                         xs, h->stream, bs, &L->Pp.cc));
    }
    {
      ProfScope p(h, AMGH_T_POSTSMOOTH, l);
      RC_TRY(smooth(h, L, L->post, xc, xo, b, bs, false, true, true, /*no_scatter*/ lo_io));
    }
    return AMGH_OK;
  }
  {
    ProfScope p(h, AMGH_T_RESIDUAL, l);
    RC_TRY(csr_apply(&L->A, M_RESID, xc, b, L->res, h->stream, bs));
  }
  {
    ProfScope p(h, AMGH_T_RESTRICT, l);
    if (L->il && il_block(bs))
      RC_TRY(il_apply(bs, false, L->R.rowptr, L->R.col, L->R.val, nc, L->res, n, n, L->il, L->cb, nc, h->stream));
    else
      RC_TRY(csr_apply(&L->R, M_SPMV, L->res, nullptr, L->cb, h->stream, bs));
  }
  RC_TRY(vec_fill(h, L->cx, nc * bs, 0.0));
  if (l == (int)h->levels.size() - 1) {
    ProfScope p(h, AMGH_T_COARSE, l + 1);
    for (int c = 0; c < bs; ++c) RC_TRY(coarse_solve(h, L->cx + c * nc, L->cb + c * nc));
  } else if (tail_here(h, l + 1, cyc) && !h->tail_lo) {
    ProfScope p(h, AMGH_T_COARSE, l + 1);
    RC_TRY(tail_apply(h, h->tail[cyc].M, nc, L->cx, nc, L->cb, nc, bs, false));
  } else {
    RC_TRY(cycle_next(h, l + 1, L->cx, L->cb, cyc));
  }
  {
    ProfScope p(h, AMGH_T_PROLONG, l);
    if (L->il && il_block(bs))
      RC_TRY(il_apply(bs, true, L->P.rowptr, L->P.col, L->P.val, n, L->cx, nc, nc, L->il, xc, n, h->stream));
    else
      RC_TRY(csr_apply(&L->P, M_ADD, L->cx, nullptr, xc, h->stream, bs));
  }
  {
    ProfScope p(h, AMGH_T_POSTSMOOTH, l);
    // the pre-smoother of this very call left b in level order if it was a level-scheduled GS / SOR (not the block path)
    const bool b_kept = (L->pre.kind == AMGH_SMOOTH_GS || L->pre.kind == AMGH_SMOOTH_SOR) && L->pre.iter > 0;
    RC_TRY(smooth(h, L, L->post, xc, xo, b, bs, false, b_kept));
  }
  if (xc != x) RC_TRY(vec_copy(h, x, xc, n * bs));
  return AMGH_OK;
}

// one application of the hierarchy: a cycle, or the coarse solve if no levels
int apply_once(amgh_t* h, real* x, const real* b, int cyc, bool xzero) {
  if (h->levels.empty()) {
    ProfScope p(h, AMGH_T_COARSE, 0);
    for (int c = 0; c < h->nrhs; ++c) RC_TRY(coarse_solve(h, x + c * h->ncoarse, b + c * h->ncoarse));
    return AMGH_OK;
  }
  if (tail_here(h, 0, cyc) && !h->tail_lo) {
    // the whole hierarchy is the tail: z = M b; on a non-zero x the same stationary iteration, x += M (b - A x)
    Level* L = h->levels[0];
    ProfScope p(h, AMGH_T_COARSE, 0);
    if (xzero) return tail_apply(h, h->tail[cyc].M, L->n, x, L->n, b, L->n, h->nrhs, false);
    RC_TRY(csr_apply(&L->A, M_RESID, x, b, L->res, h->stream, h->nrhs));
    return tail_apply(h, h->tail[cyc].M, L->n, x, L->n, L->res, L->n, h->nrhs, true);
  }
  return cycle(h, 0, x, b, cyc, xzero);
}

// The dense operator of the tail for cycles of type cyc: the library's own recursion (cycle_next / cycle: whatever smoothers,
// cycle type and coarse solver the levels carry) applied to the columns of the identity, tail_cols at a time as a block of
// right-hand sides — every step of it is linear, so the columns are the operator.  Nothing of the result depends on the
// handle's own block size.  Called outside graph capture (amgh_tail_dense_build, or the first cycle of a type: apply_cycle).
int tail_dense_build(amgh_t* h, int cyc) {
  if (h->tail_level < 0 || cyc < 0 || cyc > 2 || h->tail[cyc].built || h->coarse_fn || h->tail_building) return AMGH_OK;
  const int l0 = h->tail_level;
  Level* L0 = h->levels[l0];
  const int64_t n = L0->n;
  const int BB = h->tail_cols;
  if (BB < 1 || BB > 64) return AMGH_ESTATE;
  HIP_TRY(hipEventRecord(h->t0, h->stream));
  real *X = nullptr, *B = nullptr, *Mc = nullptr, *M = nullptr;
  auto cleanup = [&] { hipFree(X); hipFree(B); hipFree(Mc); };
  int rc = dev_alloc(&X, n * BB);
  if (rc == AMGH_OK) rc = dev_alloc(&B, n * BB);
  if (rc == AMGH_OK) rc = dev_alloc(&Mc, n * n);
  if (rc == AMGH_OK) rc = dev_alloc(&M, n * n);
  // levels below the entry that take their vectors in schedule order from the level above: sized for the build's block
  for (size_t l = (size_t)l0; rc == AMGH_OK && l + 1 < h->levels.size(); ++l)
    if (h->levels[l]->coarse_lo) rc = gs_ensure_cols(h->levels[l + 1]->smat(), BB, h->stream);
  const int nrhs0 = h->nrhs;
  const bool prof0 = h->profile;
  h->nrhs = BB; h->profile = false; h->tail_building = true;
  for (int64_t j0 = 0; j0 < n && rc == AMGH_OK; j0 += BB) {
    const int nb = (int)std::min<int64_t>(BB, n - j0);
    if (hipMemsetAsync(B, 0, sizeof(real) * n * BB, h->stream) != hipSuccess || hipMemsetAsync(X, 0, sizeof(real) * n * BB, h->stream) != hipSuccess) { rc = -1001; break; }
    hipLaunchKernelGGL(unit_cols_kernel, dim3(1), dim3(64), 0, h->stream, B, n, (int)j0, nb);
    rc = l0 == 0 ? cycle(h, 0, X, B, cyc, true) : cycle_next(h, l0, X, B, cyc, false);
    if (rc == AMGH_OK && hipMemcpyAsync(Mc + (size_t)j0 * n, X, sizeof(real) * n * nb, hipMemcpyDeviceToDevice, h->stream) != hipSuccess) rc = -1001;
  }
  h->nrhs = nrhs0; h->profile = prof0; h->tail_building = false;
  if (rc == AMGH_OK) {
    const GsSchedule* g = L0->smat()->gs;
    const int32_t* perm = h->tail_lo ? (g ? g->perm : nullptr) : nullptr;
    if (h->tail_lo && !perm) rc = AMGH_ESTATE;
    else {
      const unsigned t = (unsigned)((n + 31) / 32);
      hipLaunchKernelGGL(tail_transpose_kernel, dim3(t, t), dim3(32, 8), 0, h->stream, (const real*)Mc, perm, M, (int)n);
      if (hipGetLastError() != hipSuccess) rc = -1001;
    }
  }
  float ms = 0.f;
  if (rc == AMGH_OK && (hipEventRecord(h->t1, h->stream) != hipSuccess || hipEventSynchronize(h->t1) != hipSuccess ||
                        hipEventElapsedTime(&ms, h->t0, h->t1) != hipSuccess)) rc = -1001;
  if (rc != AMGH_OK) { hipStreamSynchronize(h->stream); cleanup(); hipFree(M); (void)hipGetLastError(); return rc; }
  cleanup();
  h->tail[cyc].M = M;
  h->tail[cyc].built = true;
  h->tail_build_ms += ms;
  h->ws_bytes += kRealB * n * n;
  return AMGH_OK;
}

// apply_once through a captured hipGraph: a cycle is thousands of short,
// launch-bound kernels (one per Gauss-Seidel dependency level), replaying them from a
// graph takes the host launch cost off the critical path.  Falls back to eager
// launches when profiling, with a host coarse solver, or if capture fails.
// xzero: the caller has just zeroed x (ldiv!, the preconditioner inside PCG)
int apply_cycle(amgh_t* h, real* x, const real* b, int cyc, bool xzero = false) {
  // (the first cycle of a type builds the tail's dense operator for it — outside any capture; a build that fails leaves the
  // per-level cycle in charge: amgh_tail_dense_build reports why)
  if (g_tail_dense && h->tail_level >= 0 && cyc >= 0 && cyc < 3 && !h->tail[cyc].built && tail_dense_build(h, cyc) != AMGH_OK) h->tail_level = -2;
  // (an F-cycle's second visit of a level is a V-cycle, multilevel.jl:209-211: the levels above the tail ask for that operator too)
  if (g_tail_dense && h->tail_level >= 0 && cyc == AMGH_CYCLE_F && !h->tail[AMGH_CYCLE_V].built && tail_dense_build(h, AMGH_CYCLE_V) != AMGH_OK) h->tail_level = -2;
  if (!h->use_graph || h->profile || h->coarse_fn || h->levels.empty()) return apply_once(h, x, b, cyc, xzero);
  if (h->graph_epoch != g_sched_epoch) {  // a schedule buffer moved: the captured pointers are stale
    for (auto& g : h->graphs) if (g.exec) hipGraphExecDestroy(g.exec);
    h->graphs.clear();
    h->graph_epoch = g_sched_epoch;
  }
  size_t warm = h->graphs.size();
  for (size_t i = 0; i < h->graphs.size(); ++i) {
    auto& g = h->graphs[i];
    if (g.x == x && g.b == b && g.cyc == cyc && g.xzero == xzero) {
      if (!g.exec) { warm = i; break; }
      HIP_TRY(hipGraphLaunch(g.exec, h->stream));
      return AMGH_OK;
    }
  }
  if (warm == h->graphs.size()) {
    // first cycle with this key: eager, so that every first-use path (scratch growth, SOR children, diagonal
    // tables: hipMalloc / hipFree / synchronisation, all illegal during capture) has run before anything is captured
    RC_TRY(apply_once(h, x, b, cyc, xzero));
    if (h->graph_epoch != g_sched_epoch) {
      for (auto& g : h->graphs) if (g.exec) hipGraphExecDestroy(g.exec);
      h->graphs.clear();
      h->graph_epoch = g_sched_epoch;
    }
    if (h->graphs.size() >= 6) {
      if (h->graphs.front().exec) hipGraphExecDestroy(h->graphs.front().exec);
      h->graphs.erase(h->graphs.begin());
    }
    h->graphs.push_back({x, b, cyc, xzero, nullptr});
    return AMGH_OK;
  }
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return apply_once(h, x, b, cyc, xzero);
  }
  const int rc = apply_once(h, x, b, cyc, xzero);
  const hipError_t e = hipStreamEndCapture(h->stream, &graph);
  if (rc != AMGH_OK || e != hipSuccess || !graph) {
    if (graph) hipGraphDestroy(graph);
    (void)hipGetLastError();
    if (rc != AMGH_OK) return rc;
    h->use_graph = false;  // capture unsupported here: stay eager
    return apply_once(h, x, b, cyc, xzero);
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (ei != hipSuccess) {
    (void)hipGetLastError();
    h->use_graph = false;
    return apply_once(h, x, b, cyc, xzero);
  }
  h->graphs[warm].exec = exec;
  HIP_TRY(hipGraphLaunch(exec, h->stream));
  return AMGH_OK;
}

int fine_residual(amgh_t* h, const real* x, const real* b, real* r) {
  if (h->levels.empty() && !h->has_finalA) return AMGH_ESTATE;
  const amgh_csr* A = h->levels.empty() ? &h->finalA : &h->levels[0]->A;
  RC_TRY(csr_apply(A, M_RESID, x, b, r, h->stream, h->nrhs));
  return AMGH_OK;
}
int fine_spmv(amgh_t* h, const real* x, real* y) {
  if (h->levels.empty()) {
    if (!h->has_finalA) return AMGH_ESTATE;
    return csr_apply(&h->finalA, M_SPMV, x, nullptr, y, h->stream);
  }
  return csr_apply(&h->levels[0]->A, M_SPMV, x, nullptr, y, h->stream);
}

// _solve! (multilevel.jl:158-198) on device pointers
int solve_dev(amgh_t* h, const real* b, real* x, int cyc, int maxiter, double abstol, double reltol,
              int calc_res, real* hist, int* iters) {
  const int64_t n = fine_n(h) * h->nrhs;  // norm(b) of an n x bs matrix is the Frobenius norm
  real normb = 0.0;
  RC_TRY(vec_norm_host(h, b, n, &normb));
  real normres = normb;
  if (normb != 0.0) abstol = std::max(reltol * normb, abstol);
  if (hist) hist[0] = normb;
  real* res = h->levels.empty() ? h->res_final : h->levels[0]->res;
  int itr = 1;
  while (itr <= maxiter && (!calc_res || normres > abstol)) {
    RC_TRY(apply_cycle(h, x, b, cyc));
    if (calc_res) {
      RC_TRY(fine_residual(h, x, b, res));
      RC_TRY(vec_norm_host(h, res, n, &normres));
      if (hist) hist[itr] = normres;
    }
    ++itr;
  }
  if (iters) *iters = itr - 1;
  HIP_TRY(hipStreamSynchronize(h->stream));
  RC_TRY(prof_flush(h));
  return bw_err_check();
}

int scal_div(amgh_t* h, int out, int a, int b) {
  hipLaunchKernelGGL(scalar_kernel, dim3(1), dim3(1), 0, h->stream, h->scal + out, h->scal + a, h->scal + b, 0);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
int scal_copy(amgh_t* h, int out, int a) {
  hipLaunchKernelGGL(scalar_kernel, dim3(1), dim3(1), 0, h->stream, h->scal + out, h->scal + a, h->scal + a, 1);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}

// Preconditioned CG, IterativeSolvers.jl `cg(A, b; Pl)` recurrence (x0 = 0):
//   c = Pl \ r ; rho_prev = rho ; rho = c.r ; beta = rho/rho_prev ; u = c + beta u
//   c = A u ; alpha = rho / u.c ; x += alpha u ; r -= alpha c ; residual = |r|
// stop when residual <= max(reltol*|r0|, abstol) or maxiter reached.
int pcg_dev(amgh_t* h, const real* b, real* x, int cyc, int use_precond, int maxiter, double abstol,
            double reltol, real* hist, int* iters) {
  const int64_t n = fine_n(h);
  real* r = h->pc_r; real* c = h->pc_c; real* u = h->pc_u;
  RC_TRY(vec_fill(h, x, n, 0.0));
  RC_TRY(vec_fill(h, u, n, 0.0));
  RC_TRY(vec_copy(h, r, b, n));
  real residual = 0.0;
  RC_TRY(vec_norm_host(h, r, n, &residual));
  const real tol = std::max(reltol * residual, abstol);
  if (hist) hist[0] = residual;
  const real one = 1.0;
  HIP_TRY(hipMemcpyAsync(h->scal + 1, &one, sizeof(real), hipMemcpyHostToDevice, h->stream));  // rho = 1
  int it = 0;
  // the recurrence between two cycles in 8 launches (pcg_scal_kernel / pcg_update_kernel); the same operations in the same
  // order: bitwise the launch-per-operation loop below (tunable pcg_fused = 0)
  if (g_pcg_fused) {
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kRedBlocks, (n + kThreads - 1) / kThreads));
    if (use_precond) RC_TRY(vec_fill(h, c, n, 0.0));   // (once: every iteration leaves c = 0 behind)
    while (it < maxiter && residual > tol) {
      if (use_precond) RC_TRY(apply_cycle(h, c, r, cyc, true));
      else RC_TRY(vec_copy(h, c, r, n));
      {
        hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(kThreads), 0, h->stream, (const real*)c, (const real*)r, n, h->partial);
        hipLaunchKernelGGL(pcg_scal_kernel, dim3(1), dim3(kThreads), 0, h->stream, (const real*)h->partial, nb, h->scal, 0);
        hipLaunchKernelGGL(xpby_dev_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, u, c, h->scal + 4, n);
        RC_TRY(fine_spmv(h, u, c));
        hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(kThreads), 0, h->stream, (const real*)u, (const real*)c, n, h->partial);
        hipLaunchKernelGGL(pcg_scal_kernel, dim3(1), dim3(kThreads), 0, h->stream, (const real*)h->partial, nb, h->scal, 1);
        hipLaunchKernelGGL(pcg_update_kernel, dim3(nb), dim3(kThreads), 0, h->stream, x, (const real*)u, r, c, (const real*)(h->scal + 3), n, h->partial);
        hipLaunchKernelGGL(pcg_scal_kernel, dim3(1), dim3(kThreads), 0, h->stream, (const real*)h->partial, nb, h->scal, 2);
      }
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(&residual, h->scal, sizeof(real), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      ++it;
      if (hist) hist[it] = residual;
    }
    if (iters) *iters = it;
    HIP_TRY(hipStreamSynchronize(h->stream));
    RC_TRY(prof_flush(h));
    return bw_err_check();
  }
  while (it < maxiter && residual > tol) {
    if (use_precond) {
      RC_TRY(vec_fill(h, c, n, 0.0));
      RC_TRY(apply_cycle(h, c, r, cyc, true));
    } else {
      RC_TRY(vec_copy(h, c, r, n));
    }
    RC_TRY(scal_copy(h, 2, 1));                 // rho_prev = rho
    RC_TRY(vec_dot(h, c, r, n, h->scal + 1, 0)); // rho = c.r
    RC_TRY(scal_div(h, 4, 1, 2));               // beta = rho / rho_prev
    hipLaunchKernelGGL(xpby_dev_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, u, c, h->scal + 4, n);
    RC_TRY(fine_spmv(h, u, c));                 // c = A u
    RC_TRY(vec_dot(h, u, c, n, h->scal + 5, 0));
    RC_TRY(scal_div(h, 3, 1, 5));               // alpha = rho / u.c
    hipLaunchKernelGGL(axpy_dev_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, x, u, h->scal + 3, 1.0, n);
    hipLaunchKernelGGL(axpy_dev_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, r, c, h->scal + 3, -1.0, n);
    HIP_TRY(hipGetLastError());
    RC_TRY(vec_norm_host(h, r, n, &residual));
    ++it;
    if (hist) hist[it] = residual;
  }
  if (iters) *iters = it;
  HIP_TRY(hipStreamSynchronize(h->stream));
  RC_TRY(prof_flush(h));
  return bw_err_check();
}

int check_ready(const amgh_t* h) {
  if (!h) return AMGH_EINVAL;
  if (!h->finalized) return AMGH_ESTATE;
  return AMGH_OK;
}

bool smoother_valid(const amgh_smoother_t* s) {
  if (!s) return false;
  if (s->kind < AMGH_SMOOTH_NONE || s->kind > AMGH_SMOOTH_SOR) return false;
  if (s->iter < 0) return false;
  if ((s->kind == AMGH_SMOOTH_GS || s->kind == AMGH_SMOOTH_SOR) &&
      (s->sweep < AMGH_SWEEP_FORWARD || s->sweep > AMGH_SWEEP_SYMMETRIC)) return false;
  return true;
}

}  // namespace

const char* amgh_rccl_error_string(int code);  // amghip_dist.hpp

extern "C" {

const char* amgh_strerror(int rc) {
  static thread_local char buf[160];
  switch (rc) {
    case AMGH_OK: return "ok";
    case AMGH_EINVAL: return "invalid argument";
    case AMGH_ESTATE: return "invalid state (not finalized, already finalized, or operator missing)";
    case AMGH_ENOMEM: return "out of device memory";
    case AMGH_EUNSUPPORTED: return "unsupported configuration";
  }
  if (rc <= -2000) {
    snprintf(buf, sizeof buf, "RCCL error %d: %s", -rc - 2000, amgh_rccl_error_string(-rc - 2000));
    return buf;
  }
  if (rc <= -1000) {
    snprintf(buf, sizeof buf, "HIP error %d: %s", -rc - 1000, hipGetErrorString((hipError_t)(-rc - 1000)));
    return buf;
  }
  snprintf(buf, sizeof buf, "unknown error %d", rc);
  return buf;
}

int amgh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int amgh_create(amgh_t** hp, int device, int nrhs) {
  if (!hp) return AMGH_EINVAL;
  *hp = nullptr;
  if (nrhs < 1 || nrhs > 64) return AMGH_EUNSUPPORTED;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(device));
  amgh_t* h = new amgh_t;
  h->device = device;
  h->nrhs = nrhs;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; return -(1000 + (int)e); }
  hipEventCreate(&h->t0);
  hipEventCreate(&h->t1);
  if (const char* e = getenv("AMGH_USE_GRAPH")) { h->use_graph = (e[0] == '1'); h->graph_auto = false; }
  *hp = h;
  return AMGH_OK;
}

static void level_discard(Level* L);

void amgh_destroy(amgh_t* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->pending_level) level_discard(h->pending_level);
  for (Level* L : h->levels) {
    csr_free(&L->A); csr_free(&L->S); csr_free(&L->P); csr_free(&L->R); csr_free(&L->Pp); csr_free(&L->Rp);
    hipFree(L->res); hipFree(L->cx); hipFree(L->cb); hipFree(L->tmp); hipFree(L->lo_val); hipFree(L->il);
    L->lo_cc.free_dev();
    delete L;
  }
  csr_free(&h->finalA);
  hipFree(h->coarse_op); hipFree(h->res_final); hipFree(h->partial); hipFree(h->scal);
  hipFree(h->x0); hipFree(h->b0); hipFree(h->pc_r); hipFree(h->pc_c); hipFree(h->pc_u);
  for (auto& t : h->tail) hipFree(t.M);
  for (auto& e : h->pending) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  for (auto& g : h->graphs) if (g.exec) hipGraphExecDestroy(g.exec);
  if (h->t0) hipEventDestroy(h->t0);
  if (h->t1) hipEventDestroy(h->t1);
  if (h->own_stream) hipStreamDestroy(h->own_stream);
  else if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

static void level_discard(Level* L) {
  csr_free(&L->A); csr_free(&L->S); csr_free(&L->P); csr_free(&L->R); csr_free(&L->Pp); csr_free(&L->Rp);
  hipFree(L->lo_val);
  L->lo_cc.free_dev();
  delete L;
}

namespace {
__global__ void perm_rowlen_kernel(const int32_t* rowptr, const int32_t* perm, int64_t n, int32_t* len) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) len[p] = rowptr[perm[p] + 1] - rowptr[perm[p]];
}
__global__ void perm_rows_kernel(const int32_t* rowptr, const int32_t* col, const real* val, const int32_t* perm,
                                 const int32_t* new_rowptr, int64_t n, int32_t* ncol, real* nval) {
  // one wavefront per row (rows of R have a handful of entries), grid-stride over the rows (a launch of n * 64 threads
  // passes 2^32 threads at 67 M rows: "invalid configuration argument"); entries keep their order inside the row
  const int ln = threadIdx.x % kWave;
  const int64_t waves = (int64_t)gridDim.x * (blockDim.x / kWave);
  for (int64_t p = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; p < n; p += waves) {
    const int32_t src = rowptr[perm[p]], len = rowptr[perm[p] + 1] - src, dst = new_rowptr[p];
    for (int e = ln; e < len; e += kWave) { ncol[dst + e] = col[src + e]; nval[dst + e] = val[src + e]; }
  }
}
__global__ void renumber_cols_kernel(int32_t* col, int64_t nnz, const int32_t* inv) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) col[k] = inv[col[k]];
}
__global__ void invert_perm_kernel(const int32_t* perm, int64_t n, int32_t* inv) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) inv[perm[p]] = (int32_t)p;
}

// Renumber the coarse side of level `prev` (rows of Rp, columns of Pp) to the dependency-level order of the next
// level, whose schedule `g` has just been built (g->perm: position -> natural row, on the device).
int coarse_side_to_level_order(Level* prev, const GsSchedule* g) {
  const int64_t nc = prev->nc;
  if ((int64_t)g->n != nc || !prev->Rp.rowptr || !prev->Pp.rowptr) return AMGH_OK;
  // Transactional: the renumbered Rp rows AND Pp columns are built beside the old ones and swapped in together only
  // when everything has succeeded — a failure leaves `prev` exactly as it was (natural-order coarse side).
  int32_t *inv = nullptr, *len = nullptr, *nrp = nullptr, *ncol = nullptr, *npcol = nullptr;
  real* nval = nullptr;
  int rc = dev_alloc(&inv, nc);
  if (rc == AMGH_OK) rc = dev_alloc(&npcol, prev->Pp.nnz);
  if (rc == AMGH_OK) rc = dev_alloc(&len, nc + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&nrp, nc + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&ncol, prev->Rp.nnz);
  if (rc == AMGH_OK) rc = dev_alloc(&nval, prev->Rp.nnz);
  if (rc == AMGH_OK) {
    const unsigned grid = (unsigned)((nc + 255) / 256);
    hipLaunchKernelGGL(invert_perm_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)g->perm, nc, inv);
    hipLaunchKernelGGL(perm_rowlen_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)prev->Rp.rowptr, (const int32_t*)g->perm, nc, len);
    int64_t total = 0;
    rc = dev_exclusive_scan(len, nrp, nc, &total, nullptr);
    if (rc == AMGH_OK && total != prev->Rp.nnz) rc = AMGH_EINVAL;
  }
  if (rc == AMGH_OK) {
    hipLaunchKernelGGL(perm_rows_kernel, dim3((unsigned)std::min<int64_t>((nc * kWave + 255) / 256, 1 << 20)), dim3(256), 0, nullptr,
                       (const int32_t*)prev->Rp.rowptr, (const int32_t*)prev->Rp.col, (const real*)prev->Rp.val,
                       (const int32_t*)g->perm, (const int32_t*)nrp, nc, ncol, nval);
    if (prev->Pp.nnz > 0 &&
        hipMemcpyAsync(npcol, prev->Pp.col, sizeof(int32_t) * (size_t)prev->Pp.nnz, hipMemcpyDeviceToDevice, nullptr) != hipSuccess)
      rc = -1001;
    hipLaunchKernelGGL(renumber_cols_kernel, dim3((unsigned)grid_for(prev->Pp.nnz)), dim3(256), 0, nullptr, npcol,
                       prev->Pp.nnz, (const int32_t*)inv);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = -1001;
  }
  if (rc == AMGH_OK) {
    hipFree(prev->Rp.rowptr); hipFree(prev->Rp.col); hipFree(prev->Rp.val); hipFree(prev->Pp.col);
    prev->Rp.rowptr = nrp; prev->Rp.col = ncol; prev->Rp.val = nval; prev->Pp.col = npcol;
    nrp = ncol = npcol = nullptr; nval = nullptr;
    prev->coarse_lo = true;
  }
  hipFree(inv); hipFree(len); hipFree(nrp); hipFree(ncol); hipFree(nval); hipFree(npcol);
  return rc == AMGH_EUNSUPPORTED ? AMGH_OK : rc;   // (scan size limits: the natural-order coarse side stays)
}
}  // namespace

// First half of push!(levels, Level(A, P, R, pre, post)): everything that needs A (and S) only — the uploads and the
// smoother schedule.  A caller that produces P and R later (the setup phase: C/F splitting on the host, then
// interpolation) can run this half on another host thread meanwhile.
// A level's own half — A (S) in HBM and the smoother schedules — touches no handle: any number of them may be under
// construction on different host threads.
struct amgh_level { Level* L = nullptr; int device = 0; };

static int level_prepare(int device, int64_t n, const int32_t* A_rowptr, const int32_t* A_col, const real* A_val,
                         const int32_t* S_rowptr, const int32_t* S_col, const real* S_val,
                         const amgh_smoother_t* pre, const amgh_smoother_t* post, Level** out, int nrhs_hint = 0) {
  HIP_TRY(hipSetDevice(device));
  (void)hipGetLastError();   // (whatever an earlier, unrelated call left behind is not this level's)
  Level* L = new Level;
  L->n = n; L->pre = *pre; L->post = *post;
  int rc = csr_upload(&L->A, device, n, n, A_rowptr, A_col, A_val);
  if (rc == AMGH_OK && S_rowptr) {
    L->has_S = true;
    rc = csr_upload(&L->S, device, n, n, S_rowptr, S_col, S_val);
  }
  const bool need_gs = pre->kind == AMGH_SMOOTH_GS || pre->kind == AMGH_SMOOTH_SOR ||
                       post->kind == AMGH_SMOOTH_GS || post->kind == AMGH_SMOOTH_SOR;
  if (rc == AMGH_OK && need_gs) {
    // schedule built from the host arrays while we still have them
    amgh_csr* M = L->smat();
    GsSchedule* g = new GsSchedule;
    rc = L->has_S ? gs_build(g, n, n, S_rowptr, S_col, S_val, nrhs_hint) : gs_build(g, n, n, A_rowptr, A_col, A_val, nrhs_hint);
    if (rc == AMGH_OK) { M->gs = g; M->bytes += g->bytes; }
    else { g->free_dev(); delete g; g = nullptr; }   // (nothing below may touch g on this path)
    if (rc == AMGH_OK && g->nblk == 0) {  // SOR: its merged groups depend on the relaxation factor, build them now too
      const int64_t before = g->bytes;
      if (pre->kind == AMGH_SMOOTH_SOR && pre->iter > 0) (void)sor_children(g, pre->omega);
      if (post->kind == AMGH_SMOOTH_SOR && post->iter > 0) (void)sor_children(g, post->omega);
      bool any_child = false;
      for (const GsSchedule::SorSet& ss : g->sor) any_child = any_child || ss.f || ss.b;
      if (any_child) rc = gs_grow_xp_for_merged(g, nullptr);  // now, while no sweep is in flight (never inside one)
      M->bytes += g->bytes - before;
    }
    // level-ordered copies of P and R when both smoothers are level-scheduled sweeps over A itself
    const bool both = (pre->kind == AMGH_SMOOTH_GS || pre->kind == AMGH_SMOOTH_SOR) && pre->iter > 0 &&
                      (post->kind == AMGH_SMOOTH_GS || post->kind == AMGH_SMOOTH_SOR) && post->iter > 0;
    // S distinct from A (Hermitian convention on a matrix that is symmetric only up to rounding, e.g. every Galerkin
    // coarse operator): the level-ordered cycle needs A's VALUES on the schedule's copy of S — possible when the two
    // have the same pattern (then row perm[p] of A lists the same columns in the same order as row p of the copy)
    bool same_pattern = !L->has_S;
    if (L->has_S && rc == AMGH_OK && both && g->nblk == 0 && g_gs_keep_lo && (int64_t)g->h_perm.size() == n &&
        A_rowptr[n] == S_rowptr[n] && std::equal(A_rowptr, A_rowptr + n + 1, S_rowptr) &&
        std::equal(A_col, A_col + A_rowptr[n], S_col)) {
      const std::vector<int32_t>& perm = g->h_perm;
      std::vector<real> lv((size_t)A_rowptr[n]);
      std::vector<int32_t> off(n + 1, 0);
      for (int64_t p2 = 0; p2 < n; ++p2) off[p2 + 1] = off[p2] + (A_rowptr[perm[p2] + 1] - A_rowptr[perm[p2]]);
      const int T = std::max(1, std::min<int>(merge_threads(), 16));
      run_threads(T, [&](int t) {
        for (int64_t p2 = n * t / T; p2 < n * (t + 1) / T; ++p2) {
          const int32_t src = A_rowptr[perm[p2]], len = A_rowptr[perm[p2] + 1] - src;
          std::copy(A_val + src, A_val + src + len, lv.begin() + off[p2]);
        }
      });
      rc = dev_upload(&L->lo_val, lv.data(), (int64_t)lv.size());
      same_pattern = rc == AMGH_OK;
      if (same_pattern) M->bytes += kRealB * (int64_t)lv.size();
    }
    // what the second half needs to know: the level-ordered P / R are wanted (g->h_perm is kept until then)
    L->lo_want = rc == AMGH_OK && both && same_pattern && g->nblk == 0 && g_gs_keep_lo && (int64_t)g->h_perm.size() == n;
    if (g && !L->lo_want) std::vector<int32_t>().swap(g->h_perm);
  }
  if (rc != AMGH_OK) {
    level_discard(L);
    return rc;
  }
  // a kernel launched with an invalid configuration fails no later call: it only shows in hipGetLastError — and its
  // output is then garbage (seen once: 2^32 threads at 67 M rows).  Never let a level through with one pending.
  {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      if (getenv("AMGH_VERBOSE")) fprintf(stderr, "[amghip] pending HIP error at the end of amgh_push_level_begin: %s\n", hipGetErrorString(e));
      level_discard(L);
      return -(1000 + (int)e);
    }
  }
  *out = L;
  return AMGH_OK;
}

int amgh_push_level_begin(amgh_t* h, int64_t n, const int32_t* A_rowptr, const int32_t* A_col, const real* A_val,
                          const int32_t* S_rowptr, const int32_t* S_col, const real* S_val,
                          const amgh_smoother_t* pre, const amgh_smoother_t* post) {
  if (!h || n <= 0 || !A_rowptr) return AMGH_EINVAL;
  if (!smoother_valid(pre) || !smoother_valid(post)) return AMGH_EINVAL;
  if (h->finalized || h->pending_level) return AMGH_ESTATE;
  if (!h->levels.empty() && h->levels.back()->nc != n) return AMGH_EINVAL;
  Level* L = nullptr;
  RC_TRY(level_prepare(h->device, n, A_rowptr, A_col, A_val, S_rowptr, S_col, S_val, pre, post, &L, h->nrhs));
  h->pending_level = L;
  return AMGH_OK;
}

// The first half without a handle: several levels' schedules can be under construction at once (one call per host
// thread); amgh_push_level_prepared then makes one of them the handle's pending level, in hierarchy order.
int amgh_level_prepare(int device, int64_t n, const int32_t* A_rowptr, const int32_t* A_col, const real* A_val,
                       const int32_t* S_rowptr, const int32_t* S_col, const real* S_val,
                       const amgh_smoother_t* pre, const amgh_smoother_t* post, amgh_level_t** out) {
  return amgh_level_prepare_nrhs(device, 0, n, A_rowptr, A_col, A_val, S_rowptr, S_col, S_val, pre, post, out);
}
int amgh_level_prepare_nrhs(int device, int nrhs, int64_t n, const int32_t* A_rowptr, const int32_t* A_col, const real* A_val,
                            const int32_t* S_rowptr, const int32_t* S_col, const real* S_val,
                            const amgh_smoother_t* pre, const amgh_smoother_t* post, amgh_level_t** out) {
  if (!out || n <= 0 || !A_rowptr || nrhs < 0) return AMGH_EINVAL;
  *out = nullptr;
  if (!smoother_valid(pre) || !smoother_valid(post)) return AMGH_EINVAL;
  Level* L = nullptr;
  RC_TRY(level_prepare(device, n, A_rowptr, A_col, A_val, S_rowptr, S_col, S_val, pre, post, &L, nrhs));
  amgh_level_t* p = new amgh_level_t;
  p->L = L; p->device = device;
  *out = p;
  return AMGH_OK;
}

void amgh_level_free(amgh_level_t* p) {
  if (!p) return;
  if (p->L) { hipSetDevice(p->device); level_discard(p->L); }
  delete p;
}

// On AMGH_OK the handle owns the level and `p` is gone; on an error `p` is untouched (amgh_level_free it).
int amgh_push_level_prepared(amgh_t* h, amgh_level_t* p) {
  if (!h || !p || !p->L) return AMGH_EINVAL;
  if (h->finalized || h->pending_level) return AMGH_ESTATE;
  if (p->device != h->device) return AMGH_EINVAL;
  if (!h->levels.empty() && h->levels.back()->nc != p->L->n) return AMGH_EINVAL;
  h->pending_level = p->L;
  p->L = nullptr;
  delete p;
  return AMGH_OK;
}

// Second half: P and R (natural order and, for the level-ordered cycle, permuted), then the level joins the hierarchy.
int amgh_push_level_end(amgh_t* h, int64_t nc, const int32_t* P_rowptr, const int32_t* P_col, const real* P_val,
                        const int32_t* R_rowptr, const int32_t* R_col, const real* R_val) {
  if (!h || nc < 0 || !P_rowptr || !R_rowptr) return AMGH_EINVAL;
  if (h->finalized || !h->pending_level) return AMGH_ESTATE;
  Level* L = h->pending_level;
  const int64_t n = L->n;
  if (P_rowptr[n] != R_rowptr[nc]) return AMGH_EINVAL;   // the pending level stays: the caller may retry or destroy
  HIP_TRY(hipSetDevice(h->device));
  (void)hipGetLastError();
  L->nc = nc;
  int rc = csr_upload(&L->P, h->device, n, nc, P_rowptr, P_col, P_val);
  if (rc == AMGH_OK) rc = csr_upload(&L->R, h->device, nc, n, R_rowptr, R_col, R_val);
  GsSchedule* g = L->smat()->gs;
  if (rc == AMGH_OK && L->lo_want && g) {
    const std::vector<int32_t>& perm = g->h_perm;
    std::vector<int32_t> inv(n);
    for (int64_t p2 = 0; p2 < n; ++p2) inv[perm[p2]] = (int32_t)p2;
    std::vector<int32_t> prp(n + 1, 0), pcl(P_rowptr[n]);
    std::vector<real> pvl(P_rowptr[n]);
    for (int64_t p2 = 0; p2 < n; ++p2) prp[p2 + 1] = prp[p2] + (P_rowptr[perm[p2] + 1] - P_rowptr[perm[p2]]);
    const int T2 = std::max(1, std::min<int>(merge_threads(), 16));
    std::vector<int32_t> rcl(R_rowptr[nc]);
    const int64_t rnnz = R_rowptr[nc];
    run_threads(T2, [&](int t) {
      for (int64_t p2 = n * t / T2; p2 < n * (t + 1) / T2; ++p2) {
        const int32_t src = P_rowptr[perm[p2]], len = P_rowptr[perm[p2] + 1] - src;
        std::copy(P_col + src, P_col + src + len, pcl.begin() + prp[p2]);
        std::copy(P_val + src, P_val + src + len, pvl.begin() + prp[p2]);
      }
      for (int64_t k = rnnz * t / T2; k < rnnz * (t + 1) / T2; ++k) rcl[k] = inv[R_col[k]];  // entries keep their order: same sums
    });
    rc = csr_upload(&L->Pp, h->device, n, nc, prp.data(), pcl.data(), pvl.data());
    if (rc == AMGH_OK) rc = csr_upload(&L->Rp, h->device, nc, n, R_rowptr, rcl.data(), R_val);
    L->lo_ok = rc == AMGH_OK;
  }
  if (g) std::vector<int32_t>().swap(g->h_perm);
  if (rc == AMGH_OK && L->lo_ok && gs_trim()) {
    // trimmed footprint (the default): between the smoothers the cycle only touches the level-ordered copies (the schedule's own A, Pp,
    // Rp); the natural-order P, R and — below the fine level, whose A the outer residual of _solve! needs — A go
    auto drop = [](amgh_csr* op) {
      hipFree(op->rowptr); hipFree(op->col); hipFree(op->val);
      op->rowptr = op->col = nullptr; op->val = nullptr;
      op->bytes -= (op->nrows + 1) * 4 + op->nnz * kEntB;
    };
    drop(&L->P); drop(&L->R);
    if (!h->levels.empty()) { drop(&L->A); if (L->has_S) drop(&L->S); }
    L->nat_freed = true;
  }
  h->pending_level = nullptr;
  if (rc != AMGH_OK) {
    level_discard(L);
    return rc;
  }
  {  // whatever this level's own construction left pending is reported BEFORE the previous level is touched
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      if (getenv("AMGH_VERBOSE")) fprintf(stderr, "[amghip] pending HIP error at the end of amgh_push_level_end: %s\n", hipGetErrorString(e));
      level_discard(L);
      return -(1000 + (int)e);
    }
  }
  // this level runs the level-ordered cycle: the level above hands its coarse vectors over in that order (the last step:
  // it either renumbers the previous level's coarse side completely or leaves it untouched, and nothing after it can fail)
  if (L->lo_ok && g && g->nblk == 0 && g_gs_keep_lo && g_gs_coarse_lo && !h->levels.empty() && h->levels.back()->lo_ok && h->nrhs >= 1) {
    rc = coarse_side_to_level_order(h->levels.back(), g);
    if (rc != AMGH_OK) { level_discard(L); return rc; }
  }
  h->levels.push_back(L);
  return AMGH_OK;
}

// The caller found out that the begun level is the coarsest one after all (size(P, 2) == 0, classical.jl:43).
int amgh_push_level_abort(amgh_t* h) {
  if (!h) return AMGH_EINVAL;
  if (!h->pending_level) return AMGH_ESTATE;
  hipSetDevice(h->device);
  level_discard(h->pending_level);
  h->pending_level = nullptr;
  return AMGH_OK;
}

int amgh_push_level(amgh_t* h, int64_t n, int64_t nc, const int32_t* A_rowptr, const int32_t* A_col,
                    const real* A_val, const int32_t* S_rowptr, const int32_t* S_col, const real* S_val,
                    const int32_t* P_rowptr, const int32_t* P_col, const real* P_val, const int32_t* R_rowptr,
                    const int32_t* R_col, const real* R_val, const amgh_smoother_t* pre,
                    const amgh_smoother_t* post) {
  if (!h || n <= 0 || nc < 0 || !A_rowptr || !P_rowptr || !R_rowptr) return AMGH_EINVAL;
  if (!smoother_valid(pre) || !smoother_valid(post)) return AMGH_EINVAL;
  if (h->finalized || h->pending_level) return AMGH_ESTATE;
  if (!h->levels.empty() && h->levels.back()->nc != n) return AMGH_EINVAL;
  if (P_rowptr[n] != R_rowptr[nc]) return AMGH_EINVAL;
  RC_TRY(amgh_push_level_begin(h, n, A_rowptr, A_col, A_val, S_rowptr, S_col, S_val, pre, post));
  int rc = amgh_push_level_end(h, nc, P_rowptr, P_col, P_val, R_rowptr, R_col, R_val);
  if (rc != AMGH_OK && h->pending_level) { level_discard(h->pending_level); h->pending_level = nullptr; }
  return rc;
}

int amgh_set_coarse(amgh_t* h, int64_t n, const int32_t* A_rowptr, const int32_t* A_col, const real* A_val,
                    const real* dense_op) {
  if (!h || n < 0 || !dense_op) return AMGH_EINVAL;
  if (h->finalized || h->ncoarse >= 0 || h->pending_level) return AMGH_ESTATE;
  if (!h->levels.empty() && h->levels.back()->nc != n) return AMGH_EINVAL;
  if (n > 46000) return AMGH_EUNSUPPORTED;  // dense n*n operator
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(dev_upload(&h->coarse_op, dense_op, n * n));
  if (A_rowptr) {
    RC_TRY(csr_upload(&h->finalA, h->device, n, n, A_rowptr, A_col, A_val));
    h->has_finalA = true;
  }
  h->ncoarse = n;
  return AMGH_OK;
}

int amgh_set_coarse_host(amgh_t* h, int64_t n, const int32_t* A_rowptr, const int32_t* A_col, const real* A_val,
                         amgh_coarse_fn fn, void* user) {
  if (!h || n < 0 || !fn) return AMGH_EINVAL;
  if (h->finalized || h->ncoarse >= 0 || h->pending_level) return AMGH_ESTATE;
  if (!h->levels.empty() && h->levels.back()->nc != n) return AMGH_EINVAL;
  if (n >= INT32_MAX) return AMGH_EUNSUPPORTED;
  HIP_TRY(hipSetDevice(h->device));
  if (A_rowptr) {
    RC_TRY(csr_upload(&h->finalA, h->device, n, n, A_rowptr, A_col, A_val));
    h->has_finalA = true;
  }
  h->coarse_fn = fn;
  h->coarse_user = user;
  h->coarse_hb.assign((size_t)std::max<int64_t>(n, 1), 0.0);
  h->coarse_hx.assign((size_t)std::max<int64_t>(n, 1), 0.0);
  h->ncoarse = n;
  return AMGH_OK;
}

int amgh_finalize(amgh_t* h) {
  if (!h) return AMGH_EINVAL;
  if (h->finalized || h->ncoarse < 0 || h->pending_level) return AMGH_ESTATE;
  if (h->levels.empty() && !h->has_finalA) return AMGH_ESTATE;
  HIP_TRY(hipSetDevice(h->device));
  int64_t ws = 0;
  // the collapsed coarse tail: the first level of at most tail_dense_rows rows (with a device coarse solver below it); its
  // operator is built later (tail_dense_build), from blocks of tail_cols right-hand sides — the workspaces of the tail's
  // levels are sized for them here (a few thousand rows: nothing)
  h->tail_level = -2;
  if (g_tail_dense_rows > 0 && !h->coarse_fn)
    for (size_t l = 0; l < h->levels.size(); ++l)
      if (h->levels[l]->n <= g_tail_dense_rows) { h->tail_level = (int)l; break; }
  if (h->tail_level >= 0) {
    int bb = std::max(1, std::min(g_tail_dense_batch, 64));
    if (bb == 2 || bb == 4 || bb == 8 || bb == 16) bb += 1;   // (not a block size of the interleaved kernels: their buffers belong to the handle's own block)
    h->tail_cols = bb;
    h->tail_lo = h->tail_level >= 1 && h->levels[h->tail_level - 1]->coarse_lo;
  }
  for (size_t l = 0; l < h->levels.size(); ++l) {
    Level* L = h->levels[l];
    const int64_t wc = (h->tail_level >= 0 && (int)l >= h->tail_level) ? std::max(h->nrhs, h->tail_cols) : h->nrhs;   // workspace columns
    RC_TRY(dev_alloc(&L->res, L->n * wc));
    RC_TRY(dev_alloc(&L->cx, L->nc * wc));
    RC_TRY(dev_alloc(&L->cb, L->nc * wc));
    ws += kRealB * (L->n + 2 * L->nc) * wc;
    if (h->nrhs == 2 || h->nrhs == 4 || h->nrhs == 8 || h->nrhs == 16) {
      RC_TRY(dev_alloc(&L->il, std::max(L->n, L->nc) * h->nrhs));
      ws += kRealB * std::max(L->n, L->nc) * h->nrhs;
    }
    if (L->pre.kind == AMGH_SMOOTH_JACOBI || L->post.kind == AMGH_SMOOTH_JACOBI) {
      RC_TRY(dev_alloc(&L->tmp, L->n * wc));
      ws += kRealB * L->n * wc;
      RC_TRY(csr_ensure_diag(L->smat(), h->stream));
    }
  }
  // value-coded columns of the big operators the level-ordered cycle streams (its A, Rp, Pp: everything is in its final order
  // by now), where they hold few distinct values
  for (Level* L : h->levels) {
    GsSchedule* g = L->smat()->gs;
    if (!L->lo_ok || !g || g->nblk != 0 || !g->rowptr || g->compacted) continue;
    RC_TRY(code_values(g->col, L->lo_val ? L->lo_val : g->val, L->n, g->ncols, g->nnz, &L->lo_cc, h->stream));
    RC_TRY(code_values(L->Rp.col, L->Rp.val, L->Rp.nrows, L->Rp.ncols, L->Rp.nnz, &L->Rp.cc, h->stream));
    RC_TRY(code_values(L->Pp.col, L->Pp.val, L->Pp.nrows, L->Pp.ncols, L->Pp.nnz, &L->Pp.cc, h->stream));
    ws += L->lo_cc.bytes + L->Rp.cc.bytes + L->Pp.cc.bytes;
    // blocks of right-hand sides: with 4 bytes per entry the stream kernel, column after column out of L2, can beat the interleaved
    // gather + its transposition pass (256^3, bs = 8: restriction 1.26 -> 0.95 ms, prolongation 1.68 -> 1.35 on the fine level —
    // and 0.53 -> 1.14 on the next one): measured here, per operator, on the level's own buffers (their contents do not matter yet)
    if (L->il && il_block(h->nrhs) && g_rhs_il == 1) {
      const int bs = h->nrhs;
      const int64_t n = L->n, nc = L->nc;
      auto timed = [&](auto&& fn, double* ms) -> int {
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess) return -1001;
        if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return -1001; }
        int rc = fn();
        if (rc == AMGH_OK && hipEventRecord(e0, h->stream) != hipSuccess) rc = -1001;
        for (int r = 0; r < 2 && rc == AMGH_OK; ++r) rc = fn();
        float t = 0.f;
        if (rc == AMGH_OK && (hipEventRecord(e1, h->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                              hipEventElapsedTime(&t, e0, e1) != hipSuccess)) rc = -1001;
        hipEventDestroy(e0); hipEventDestroy(e1);
        *ms = t;
        return rc;
      };
      if (L->Rp.cc.ccol) {
        double t_il = 0, t_st = 0;
        RC_TRY(hipMemsetAsync(L->res, 0, sizeof(real) * n * bs, h->stream) == hipSuccess ? AMGH_OK : -1001);
        RC_TRY(timed([&] { return il_apply(bs, false, L->Rp.rowptr, L->Rp.col, L->Rp.val, nc, L->res, n, n, L->il, L->cb, nc, h->stream, &L->Rp.cc); }, &t_il));
        RC_TRY(timed([&] { return csr_apply(&L->Rp, M_SPMV, L->res, nullptr, L->cb, h->stream, bs); }, &t_st));
        L->r_stream = t_st < 0.95 * t_il;
        if (getenv("AMGH_VERBOSE")) fprintf(stderr, "[amghip] n=%lld restriction of %d columns: interleaved %.3f ms, column by column %.3f ms\n", (long long)n, bs, t_il / 2, t_st / 2);
      }
      if (L->Pp.cc.ccol) {
        double t_il = 0, t_st = 0;
        RC_TRY(hipMemsetAsync(L->cx, 0, sizeof(real) * nc * bs, h->stream) == hipSuccess ? AMGH_OK : -1001);
        RC_TRY(hipMemsetAsync(L->res, 0, sizeof(real) * n * bs, h->stream) == hipSuccess ? AMGH_OK : -1001);
        RC_TRY(timed([&] { return il_apply(bs, true, L->Pp.rowptr, L->Pp.col, L->Pp.val, n, L->cx, nc, nc, L->il, L->res, n, h->stream, &L->Pp.cc); }, &t_il));
        RC_TRY(timed([&] { return raw_apply(M_ADD, L->Pp.rowptr, L->Pp.col, L->Pp.val, n, L->cx, nc, nullptr, 0, L->res, n, h->stream, bs, &L->Pp.cc); }, &t_st));
        L->p_stream = t_st < 0.95 * t_il;
        if (getenv("AMGH_VERBOSE")) fprintf(stderr, "[amghip] n=%lld prolongation of %d columns: interleaved %.3f ms, column by column %.3f ms\n", (long long)n, bs, t_il / 2, t_st / 2);
      }
    }
  }
  // The restriction of a big level with the XCD-contiguous mapping of its workgroups where that is faster (a coarse row gathers from a
  // fine vector several times its own size: which L2 holds the lines matters) — timed here on the level's own buffers, single column.
  if (h->nrhs == 1 && g_stream_xcd)
    for (Level* L : h->levels) {
      if (!L->lo_ok || !L->Rp.rowptr || L->Rp.nrows < (1 << 18)) continue;
      if (hipMemsetAsync(L->res, 0, sizeof(real) * L->n, h->stream) != hipSuccess) return -1001;
      double ms[2] = {0, 0};
      int rc = AMGH_OK;
      for (int v = 0; v < 2 && rc == AMGH_OK; ++v) {
        L->Rp.xcd_map = v == 1;
        rc = csr_apply(&L->Rp, M_SPMV, L->res, nullptr, L->cb, h->stream, 1);
        if (rc == AMGH_OK && hipEventRecord(h->t0, h->stream) != hipSuccess) rc = -1001;
        for (int r = 0; r < 4 && rc == AMGH_OK; ++r) rc = csr_apply(&L->Rp, M_SPMV, L->res, nullptr, L->cb, h->stream, 1);
        float t = 0.f;
        if (rc == AMGH_OK && (hipEventRecord(h->t1, h->stream) != hipSuccess || hipEventSynchronize(h->t1) != hipSuccess ||
                              hipEventElapsedTime(&t, h->t0, h->t1) != hipSuccess)) rc = -1001;
        ms[v] = t;
      }
      L->Rp.xcd_map = false;
      RC_TRY(rc);
      L->Rp.xcd_map = ms[1] < 0.93 * ms[0];
      if (getenv("AMGH_VERBOSE")) fprintf(stderr, "[amghip] n=%lld restriction: %.3f ms, XCD-contiguous %.3f ms -> %s\n", (long long)L->n, ms[0] / 4, ms[1] / 4, L->Rp.xcd_map ? "XCD-contiguous" : "plain");
    }
  // Trimmed footprint (the default): where an operator the level-ordered cycle streams has value-coded columns, the cycle reads
  // THEM — 4 bytes per entry — and the 12 bytes of columns and values beside them are a copy nothing in it touches: released
  // (the restriction and prolongation of the level; its A where the level sweeps the block layout as a dataflow alone, so that
  // no kernel of the sweep reads the level-ordered CSR either).  256^3: 20.4 -> 15.2 GB.  The stand-alone hooks
  // (amgh_level_spmv, amgh_bench_op) read the coded columns too (level_lo_op); the run-time tunable stream_code = 0 then has no
  // plain copy to switch to on these operators (AMGH_LEAN=0 / gs_lean = 0 keeps both).
  if (gs_trim() && g_stream_code && g_trim_coded) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (Level* L : h->levels) {
      for (amgh_csr* M : {&L->Rp, &L->Pp})
        if (M->cc.ccol && M->col && M->val) {
          hipFree(M->col); hipFree(M->val); M->col = nullptr; M->val = nullptr;
          M->bytes -= M->nnz * kEntB;
        }
      amgh_csr* S = L->smat();
      GsSchedule* g = S->gs;
      if (L->lo_cc.ccol && g && g->bw.on && g->bw.flow.on && !g->bw.rec && g->segs.empty() && !g->mf && !g->mb && g->col && g->val) {
        hipFree(g->col); hipFree(g->val); g->col = nullptr; g->val = nullptr;
        int64_t freed = g->nnz * kEntB;
        if (L->lo_val) { hipFree(L->lo_val); L->lo_val = nullptr; }
        g->csr_bytes -= freed; g->bytes -= freed; S->bytes -= freed;
      }
    }
  }
  // a level that receives its right-hand side in level order gets it written into its schedule's own vector by the
  // level above: sized for the block of right-hand sides before the first cycle
  for (size_t l = 0; l + 1 < h->levels.size(); ++l)
    if (h->levels[l]->coarse_lo) RC_TRY(gs_ensure_cols(h->levels[l + 1]->smat(), h->nrhs, h->stream));
  const int64_t n = fine_n(h);
  if (h->levels.empty()) { RC_TRY(dev_alloc(&h->res_final, n * h->nrhs)); ws += kRealB * n * h->nrhs; }
  RC_TRY(dev_alloc(&h->partial, kRedBlocks));
  RC_TRY(dev_alloc(&h->scal, 8));
  RC_TRY(dev_alloc(&h->x0, n * h->nrhs));
  RC_TRY(dev_alloc(&h->b0, n * h->nrhs));
  ws += 2 * kRealB * n * h->nrhs;
  h->ws_bytes = ws;
  // Graph replay stays opt-in (AMGH_USE_GRAPH=1 / amgh_set_use_graph), for small hierarchies too: measured on C1 / C2 / C5
  // (profiles/r03_small_configs.log) a replayed cycle is as fast as the eager one (0.243 vs 0.248, 0.241 vs 0.232,
  // 0.341 vs 0.337 ms) — their kernels take >= 3 us each, the host keeps ahead of the device — and rocprofv3's kernel
  // tracing aborts on replays of this size.  AMGH_GRAPH_AUTO=1 turns it on for hierarchies whose widest level has at
  // most kGraphAutoRows rows.
  if (h->graph_auto && !h->levels.empty() && getenv("AMGH_GRAPH_AUTO") && getenv("AMGH_GRAPH_AUTO")[0] == '1') {
    int64_t widest = 0;
    for (Level* L : h->levels) widest = std::max(widest, L->n);
    h->use_graph = widest <= kGraphAutoRows;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->finalized = true;
  return AMGH_OK;
}

int amgh_num_levels(const amgh_t* h) { return h ? (int)h->levels.size() : 0; }
int amgh_tail_dense_build(amgh_t* h, int cycle_) {
  RC_TRY(check_ready(h));
  if (cycle_ < 0 || cycle_ > 2) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  if (!g_tail_dense || h->tail_level < 0) return AMGH_OK;
  RC_TRY(tail_dense_build(h, cycle_));
  if (cycle_ == AMGH_CYCLE_F) RC_TRY(tail_dense_build(h, AMGH_CYCLE_V));   // (its second visits are V-cycles)
  return AMGH_OK;
}
int amgh_tail_dense_info(const amgh_t* h, int cycle_, int* level, int64_t* rows, double* build_ms) {
  if (!h || cycle_ < 0 || cycle_ > 2) return AMGH_EINVAL;
  const bool on = h->finalized && h->tail_level >= 0 && h->tail[cycle_].built;
  if (level) *level = on ? h->tail_level : -1;
  if (rows) *rows = on ? h->levels[h->tail_level]->n : 0;
  if (build_ms) *build_ms = h->tail_build_ms;
  return AMGH_OK;
}
int64_t amgh_level_size(const amgh_t* h, int l) {
  if (!h || l < 0 || l > (int)h->levels.size()) return -1;
  return l == (int)h->levels.size() ? h->ncoarse : h->levels[l]->n;
}
int64_t amgh_device_bytes(const amgh_t* h) {
  if (!h) return 0;
  int64_t b = h->ws_bytes + ((h->ncoarse > 0 && h->coarse_op) ? h->ncoarse * h->ncoarse * kRealB : 0) + h->finalA.bytes;
  for (Level* L : h->levels) b += L->A.bytes + L->S.bytes + L->P.bytes + L->R.bytes + L->Pp.bytes + L->Rp.bytes;
  return b;
}
// out8 = {natural-order A/P/R (+ S), level-ordered CSR copies (schedules' A, Pp, Rp), un-merged slot arrays, merged groups:
//         CSR part, merged groups: slot arrays, pre-pass triangles, block-inverse data + vectors of the schedules, workspace}
int amgh_device_bytes_detail(const amgh_t* h, int64_t* out8) {
  if (!h || !out8) return AMGH_EINVAL;
  for (int q = 0; q < 8; ++q) out8[q] = 0;
  for (Level* L : h->levels) {
    for (const amgh_csr* op : {&L->A, &L->S, &L->P, &L->R})
      if (op->rowptr) out8[0] += (op->nrows + 1) * 4 + op->nnz * kEntB;
    out8[1] += L->Pp.bytes + L->Rp.bytes;
    const GsSchedule* g = L->smat()->gs;
    if (!g) continue;
    out8[1] += g->csr_bytes;
    out8[2] += g->slot_bytes;
    int64_t known = g->csr_bytes + g->slot_bytes;
    for (const GsSchedule* c : {g->mf, g->mb})
      if (c) { out8[3] += c->csr_bytes; out8[4] += c->bytes - c->csr_bytes; known += c->bytes; }   // [4]: slots, SELL-like copy, diagonals
    const int64_t tri = (g->tri_nnz + g->tri_nnz_b) * kEntB + ((g->tri_nnz ? 1 : 0) + (g->tri_nnz_b ? 1 : 0)) * (g->n + 1) * 4;
    out8[5] += tri;
    known += tri;
    out8[6] += std::max<int64_t>(0, g->bytes - known);
  }
  out8[0] += h->finalA.bytes;
  out8[7] = h->ws_bytes + ((h->ncoarse > 0 && h->coarse_op) ? h->ncoarse * h->ncoarse * kRealB : 0);
  return AMGH_OK;
}
// Diagnostics of the chained / dataflow wavefront of blocks: the poll give-up word (always 0: a block only waits for
// blocks with smaller tickets).  The bound turns a protocol error into an ERROR instead of a hang: a give-up raises this
// word and the next synchronising entry point (amgh_solve, amgh_precond_apply, amgh_pcg, amgh_dev_sync, amgh_dist_*)
// returns AMGH_ESTATE — the values of that sweep are not to be used.  -1 when level l has no such schedule.
// Synchronises the device; reads the word without clearing it.
int amgh_debug_bw_poll_giveups(const amgh_t* h, int l) {
  if (!h || l < 0 || l >= (int)h->levels.size()) return -1;
  amgh_csr* M = h->levels[l]->smat();
  const GsSchedule* g = M->gs;
  if (!g || !g->bw.on || !g->bw.err) return -1;
  if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
  return (int)*(volatile int32_t*)g->bw.err;   // (the process-wide word, left as it is: the next synchronising entry point reports it)
}
int amgh_debug_bw_mode(const amgh_t* h, int l) {
  if (!h || l < 0 || l >= (int)h->levels.size()) return -1;
  amgh_csr* M = h->levels[l]->smat();
  const GsSchedule* g = M->gs;
  if (!g || !g->bw.on) return 0;
  if (g->bw.flow.on && (g_gs_bw_flow || !g->bw.rec)) return 3;
  return (g_gs_bw_chain && g->bw.flags) ? 2 : 1;
}
int amgh_debug_bw_dict(const amgh_t* h, int l) {
  if (!h || l < 0 || l >= (int)h->levels.size()) return -1;
  const GsSchedule* g = h->levels[l]->smat()->gs;
  return g && g->bw.on && g->bw.flow.on && g->bw.flow.dict_on && (g_gs_bw_dict || !g->bw.flow.srec) && (g_gs_bw_flow || !g->bw.rec) ? 1 : 0;
}
int amgh_debug_bw_late(const amgh_t* h, int l) {
  if (!h || l < 0 || l >= (int)h->levels.size()) return -1;
  const GsSchedule* g = h->levels[l]->smat()->gs;
  return g && g->bw.on && g->bw.flow.on && g->bw.flow.late_ok && !g_gs_bw_inorder && g_gs_bw_relay > 0 && h->nrhs == 1 && (g_gs_bw_flow || !g->bw.rec) ? 1 : 0;
}
int amgh_debug_coded_ops(const amgh_t* h, int l) {
  if (!h || l < 0 || l >= (int)h->levels.size()) return -1;
  const Level* L = h->levels[l];
  // (what the launches take: the coded columns where they exist and either the tunable says so or no plain copy is left)
  const GsSchedule* g = (L->has_S ? L->S : L->A).gs;
  const bool a_on = L->lo_cc.ccol && (g_stream_code || (g && !g->col));
  const bool r_on = L->Rp.cc.ccol && (g_stream_code || !L->Rp.col), p_on = L->Pp.cc.ccol && (g_stream_code || !L->Pp.col);
  return (a_on ? 1 : 0) | (r_on ? 2 : 0) | (p_on ? 4 : 0);
}
int amgh_gs_num_dependency_levels(const amgh_t* h, int l) {
  if (!h || l < 0 || l >= (int)h->levels.size()) return -1;
  amgh_csr* M = h->levels[l]->smat();
  return M->gs ? M->gs->nlev : 0;
}
int amgh_gs_num_sweep_steps(const amgh_t* h, int l, int backward) {
  if (!h || l < 0 || l >= (int)h->levels.size()) return -1;
  amgh_csr* M = h->levels[l]->smat();
  const GsSchedule* g = M->gs;
  if (!g) return 0;
  if (g->dti_f && g_gs_dense_tri && g_gs_block_inverse) return (int)g->dti_off.size() - 1;   // dense triangular sweep: blocks
  if (g->nblk > 0 && g_gs_block_inverse) return g->nblk;             // block-inverse sweep: sequential block steps
  if (g->bw.on) return (int)g->bw.launch_ptr.size() - 1;             // wavefront of blocks: depths of the quotient DAG
  const GsSchedule* c = backward ? g->mb : g->mf;
  return (c && g_gs_merge > 1) ? c->nlev : g->nlev;                  // merged groups, or dependency levels
}

int amgh_gs_sweep_stats(const amgh_t* h, int l, int backward, int64_t* out6) {
  if (!h || l < 0 || l >= (int)h->levels.size() || !out6) return AMGH_EINVAL;
  for (int q = 0; q < 6; ++q) out6[q] = 0;
  amgh_csr* M = h->levels[l]->smat();
  const GsSchedule* g = M->gs;
  if (!g) return AMGH_OK;
  if (g->dti_f && g_gs_dense_tri && g_gs_block_inverse) {  // dense triangular sweep: pre-pass + one triangular GEMV
    out6[0] = 2 * ((int64_t)g->dti_off.size() - 1);
    out6[1] = g->n;
    out6[2] = out6[3] = g->nnz + g->dti_off.back() / 2;
    return AMGH_OK;
  }
  if (g->nblk > 0 && g_gs_block_inverse) {  // block-inverse sweep: one launch per superblock (+ its pre-pass), dense inverses streamed
    const int S = g->super > 0 ? g->super : g->nblk;
    out6[0] = 2 * ((g->nblk + S - 1) / S);
    out6[1] = g->n;
    out6[2] = out6[3] = g->nnz + (int64_t)g->nblk * kBlk * kBlk;
    out6[5] = 0;
    return AMGH_OK;
  }
  if (g->bw.on) {   // wavefront of blocks: the operator's own entries (padded to the record's row width), no pre-pass
    const int64_t depths = (int64_t)g->bw.launch_ptr.size() - 1;
    const bool one = (g->bw.flow.on && (g_gs_bw_flow || !g->bw.rec)) || (g_gs_bw_chain && g->bw.flags);
    out6[0] = one ? 1 : depths;   // dataflow / chained by flags: one launch per sweep
    out6[1] = g->n;
    out6[2] = g->nnz - g->n;
    out6[3] = g->bw.rec_entries;
    out6[4] = 0;
    out6[5] = depths > 0 ? (g->nlev + depths - 1) / depths : 0;
    return AMGH_OK;
  }
  const GsSchedule* c = backward ? g->mb : g->mf;
  const bool merged = c && g_gs_merge > 1;
  const GsSchedule* lay = merged ? c : g;
  out6[0] = (int64_t)lay->segs.size();
  out6[1] = g->n;
  int64_t slotted = 0;  // entries of the rows that run from the slot arrays are counted with their padding
  for (const GsSchedule::Seg& sg : lay->segs)
    if (!sg.chain && sg.nslots > 0) slotted += (int64_t)sg.nslots * lay->slot_entries;
  out6[2] = lay->nnz;
  out6[3] = slotted;
  out6[4] = merged ? (backward ? g->tri_nnz_b : g->tri_nnz) : 0;
  out6[5] = merged ? (backward ? g->merge_b : g->merge_f) : 1;
  return AMGH_OK;
}

static int ensure_pcg_bufs(amgh_t* h) {
  if (h->pc_r) return AMGH_OK;
  const int64_t n = fine_n(h);
  RC_TRY(dev_alloc(&h->pc_r, n));
  RC_TRY(dev_alloc(&h->pc_c, n));
  RC_TRY(dev_alloc(&h->pc_u, n));
  h->ws_bytes += 3 * kRealB * n;
  return AMGH_OK;
}

int amgh_solve_d(amgh_t* h, const real* b_d, real* x_d, int cycle_, int maxiter, double abstol, double reltol,
                 int calculate_residual, real* resid_hist, int* iters) {
  RC_TRY(check_ready(h));
  if (!b_d || !x_d || cycle_ < 0 || cycle_ > 2 || maxiter < 0) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  return solve_dev(h, b_d, x_d, cycle_, maxiter, abstol, reltol, calculate_residual, resid_hist, iters);
}

int amgh_solve(amgh_t* h, const real* b, real* x, int cycle_, int maxiter, double abstol, double reltol,
               int calculate_residual, real* resid_hist, int* iters) {
  RC_TRY(check_ready(h));
  if (!b || !x || cycle_ < 0 || cycle_ > 2 || maxiter < 0) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  const int64_t n = fine_n(h) * h->nrhs;
  HIP_TRY(hipMemcpyAsync(h->b0, b, sizeof(real) * n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->x0, x, sizeof(real) * n, hipMemcpyHostToDevice, h->stream));
  RC_TRY(solve_dev(h, h->b0, h->x0, cycle_, maxiter, abstol, reltol, calculate_residual, resid_hist, iters));
  HIP_TRY(hipMemcpy(x, h->x0, sizeof(real) * n, hipMemcpyDeviceToHost));
  return AMGH_OK;
}

int amgh_precond_apply_d(amgh_t* h, const real* r_d, real* z_d, int cycle_) {
  RC_TRY(check_ready(h));
  if (!r_d || !z_d || cycle_ < 0 || cycle_ > 2) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(bw_err_check());   // (asynchronous: a sweep of an earlier call that gave up a poll is reported here, or by amgh_dev_sync)
  RC_TRY(vec_fill(h, z_d, fine_n(h) * h->nrhs, 0.0));
  return apply_cycle(h, z_d, r_d, cycle_, true);
}

int amgh_precond_apply(amgh_t* h, const real* r, real* z, int cycle_) {
  RC_TRY(check_ready(h));
  if (!r || !z) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  const int64_t n = fine_n(h) * h->nrhs;
  HIP_TRY(hipMemcpyAsync(h->b0, r, sizeof(real) * n, hipMemcpyHostToDevice, h->stream));
  RC_TRY(amgh_precond_apply_d(h, h->b0, h->x0, cycle_));
  HIP_TRY(hipStreamSynchronize(h->stream));
  RC_TRY(prof_flush(h));
  RC_TRY(bw_err_check());
  HIP_TRY(hipMemcpy(z, h->x0, sizeof(real) * n, hipMemcpyDeviceToHost));
  return AMGH_OK;
}

int amgh_cycle_d(amgh_t* h, int level, real* x_d, const real* b_d, int cycle_) {
  RC_TRY(check_ready(h));
  if (!x_d || !b_d || cycle_ < 0 || cycle_ > 2 || level < 0 || level > (int)h->levels.size()) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  if (level == (int)h->levels.size()) return coarse_solve(h, x_d, b_d);
  return cycle(h, level, x_d, b_d, cycle_, false);
}

int amgh_set_stream(amgh_t* h, void* stream) {
  if (!h) return AMGH_EINVAL;
  h->ext_stream = true;
  h->own_stream = h->own_stream ? h->own_stream : h->stream;
  h->stream = (hipStream_t)stream;  // NULL = the default (null) stream, e.g. torch's current stream
  return AMGH_OK;
}

int amgh_pcg_d(amgh_t* h, const real* b_d, real* x_d, int cycle_, int use_precond, int maxiter, double abstol,
               double reltol, real* resid_hist, int* iters) {
  RC_TRY(check_ready(h));
  if (!b_d || !x_d || cycle_ < 0 || cycle_ > 2 || maxiter < 0) return AMGH_EINVAL;
  if (h->nrhs != 1) return AMGH_EUNSUPPORTED;  // IterativeSolvers' cg takes vectors
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(ensure_pcg_bufs(h));
  return pcg_dev(h, b_d, x_d, cycle_, use_precond, maxiter, abstol, reltol, resid_hist, iters);
}

int amgh_pcg(amgh_t* h, const real* b, real* x, int cycle_, int use_precond, int maxiter, double abstol,
             double reltol, real* resid_hist, int* iters) {
  RC_TRY(check_ready(h));
  if (!b || !x) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  const int64_t n = fine_n(h);
  HIP_TRY(hipMemcpyAsync(h->b0, b, sizeof(real) * n, hipMemcpyHostToDevice, h->stream));
  RC_TRY(amgh_pcg_d(h, h->b0, h->x0, cycle_, use_precond, maxiter, abstol, reltol, resid_hist, iters));
  HIP_TRY(hipMemcpy(x, h->x0, sizeof(real) * n, hipMemcpyDeviceToHost));
  return AMGH_OK;
}

static amgh_csr* level_op(amgh_t* h, int level, int which) {
  const int L = (int)h->levels.size();
  if (level == L && which == AMGH_OP_A && h->has_finalA) return &h->finalA;
  if (level < 0 || level >= L) return nullptr;
  amgh_csr* op = nullptr;
  switch (which) {
    case AMGH_OP_A: op = &h->levels[level]->A; break;
    case AMGH_OP_P: op = &h->levels[level]->P; break;
    case AMGH_OP_R: op = &h->levels[level]->R; break;
  }
  return (op && op->rowptr) ? op : nullptr;  // (memory-lean hierarchies have released some natural-order operators)
}

// The level's operator in LEVEL ORDER (what the cycle itself multiplies with): rows / columns / values and the two
// permutations (position -> natural index; nullptr = natural order on that side) of its row side and its column side.
struct LoOp {
  const int32_t *rowptr = nullptr, *col = nullptr;
  const real* val = nullptr;
  int64_t nrows = 0, ncols = 0;
  const int32_t *row_perm = nullptr, *col_perm = nullptr;
  const CodedCols* cc = nullptr;   // value-coded columns of the same operator (the trimmed footprint keeps only them: col / val nullptr)
};
static bool level_lo_op(amgh_t* h, int level, int which, LoOp* o) {
  if (level < 0 || level >= (int)h->levels.size()) return false;
  Level* L = h->levels[level];
  GsSchedule* g = L->smat()->gs;
  if (!L->lo_ok || !g || !g->perm) return false;
  const int32_t* coarse_perm = nullptr;
  if (L->coarse_lo) {
    if (level + 1 >= (int)h->levels.size()) return false;
    GsSchedule* cg = h->levels[level + 1]->smat()->gs;
    if (!cg || !cg->perm) return false;
    coarse_perm = cg->perm;
  }
  switch (which) {
    case AMGH_OP_A:
      if ((int64_t)g->ncols != L->n) return false;
      *o = LoOp{g->rowptr, g->col, L->lo_val ? L->lo_val : g->val, L->n, L->n, g->perm, g->perm, &L->lo_cc};
      return g->rowptr != nullptr && !g->compacted;
    case AMGH_OP_P:
      *o = LoOp{L->Pp.rowptr, L->Pp.col, L->Pp.val, L->n, L->nc, g->perm, coarse_perm, &L->Pp.cc};
      return L->Pp.rowptr != nullptr;
    case AMGH_OP_R:
      *o = LoOp{L->Rp.rowptr, L->Rp.col, L->Rp.val, L->nc, L->n, coarse_perm, g->perm, &L->Rp.cc};
      return L->Rp.rowptr != nullptr;
  }
  return false;
}

// y = op x (M_SPMV) or y = b - op x (M_RESID) for the stand-alone hooks, natural order in and out.  Where the
// natural-order copy of the operator was released (the default footprint keeps only what the cycle multiplies with),
// the level-ordered copy does the product between a gather of x and a scatter of y: same entries in the same order
// inside every row, hence the same sums.
static int level_apply(amgh_t* h, int level, int which, int mode, const real* x_d, const real* b_d, real* y_d) {
  if (amgh_csr* op = level_op(h, level, which)) return csr_apply(op, mode, x_d, b_d, y_d, h->stream);
  LoOp o;
  if (!level_lo_op(h, level, which, &o)) return AMGH_EINVAL;
  real *xi = nullptr, *yo = nullptr, *bi = nullptr;
  int rc = AMGH_OK;
  const real* xin = x_d;
  const real* bin = b_d;
  if (o.col_perm) {
    rc = dev_alloc(&xi, o.ncols);
    if (rc == AMGH_OK)
      hipLaunchKernelGGL(gather_perm_kernel, dim3(grid_for(o.ncols)), dim3(256), 0, h->stream, x_d, o.col_perm, xi, (int)o.ncols,
                         (int64_t)0, (int64_t)0);
    xin = xi;
  }
  if (rc == AMGH_OK && o.row_perm) {
    rc = dev_alloc(&yo, o.nrows);
    if (rc == AMGH_OK && mode == M_RESID) {
      rc = dev_alloc(&bi, o.nrows);
      if (rc == AMGH_OK)
        hipLaunchKernelGGL(gather_perm_kernel, dim3(grid_for(o.nrows)), dim3(256), 0, h->stream, b_d, o.row_perm, bi, (int)o.nrows,
                           (int64_t)0, (int64_t)0);
      bin = bi;
    }
  }
  if (rc == AMGH_OK)
    rc = raw_apply(mode, o.rowptr, o.col, o.val, o.nrows, xin, o.ncols, bin, o.nrows, o.row_perm ? yo : y_d, o.nrows, h->stream, 1, o.cc);
  if (rc == AMGH_OK && o.row_perm)
    hipLaunchKernelGGL(scatter_perm_kernel, dim3(grid_for(o.nrows)), dim3(256), 0, h->stream, (const real*)yo, o.row_perm, y_d,
                       (int)o.nrows, (int64_t)0, (int64_t)0);
  if (rc == AMGH_OK && hipGetLastError() != hipSuccess) rc = -1001;
  if (hipStreamSynchronize(h->stream) != hipSuccess && rc == AMGH_OK) rc = -1001;   // the temporaries go now
  hipFree(xi); hipFree(yo); hipFree(bi);
  return rc;
}
// rows x columns of the level's operator (whichever copy exists)
static bool level_op_shape(amgh_t* h, int level, int which, int64_t* nrows, int64_t* ncols) {
  if (amgh_csr* op = level_op(h, level, which)) { *nrows = op->nrows; *ncols = op->ncols; return true; }
  LoOp o;
  if (!level_lo_op(h, level, which, &o)) return false;
  *nrows = o.nrows; *ncols = o.ncols;
  return true;
}

int amgh_level_spmv_d(amgh_t* h, int level, int which, const real* x_d, real* y_d) {
  RC_TRY(check_ready(h));
  if (!x_d || !y_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(level_apply(h, level, which, M_SPMV, x_d, nullptr, y_d));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return AMGH_OK;
}

int amgh_level_spmv(amgh_t* h, int level, int which, const real* x, real* y) {
  RC_TRY(check_ready(h));
  int64_t nr = 0, ncl = 0;
  if (!x || !y || !level_op_shape(h, level, which, &nr, &ncl)) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  real *xd = nullptr, *yd = nullptr;
  RC_TRY(dev_upload(&xd, x, ncl));
  int rc = dev_alloc(&yd, nr);
  if (rc == AMGH_OK) rc = level_apply(h, level, which, M_SPMV, xd, nullptr, yd);
  if (rc == AMGH_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = -1000 - (int)hipGetLastError();
  if (rc == AMGH_OK && hipMemcpy(y, yd, sizeof(real) * nr, hipMemcpyDeviceToHost) != hipSuccess) rc = -1001;
  hipFree(xd); hipFree(yd);
  return rc;
}

int amgh_level_residual_d(amgh_t* h, int level, const real* x_d, const real* b_d, real* r_d) {
  RC_TRY(check_ready(h));
  if (!x_d || !b_d || !r_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(level_apply(h, level, AMGH_OP_A, M_RESID, x_d, b_d, r_d));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return AMGH_OK;
}

static int level_smooth_enqueue(amgh_t* h, int level, int post, real* x_d, const real* b_d) {
  Level* L = h->levels[level];
  const amgh_smoother_t& s = post ? L->post : L->pre;
  real* xc = x_d;
  real* xo = L->tmp;
  if (s.kind == AMGH_SMOOTH_JACOBI && !xo) return AMGH_ESTATE;
  RC_TRY(smooth(h, L, s, xc, xo, b_d));
  if (xc != x_d) RC_TRY(vec_copy(h, x_d, xc, L->n));
  return AMGH_OK;
}

int amgh_level_smooth_d(amgh_t* h, int level, int post, real* x_d, const real* b_d) {
  RC_TRY(check_ready(h));
  if (level < 0 || level >= (int)h->levels.size() || !x_d || !b_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(level_smooth_enqueue(h, level, post, x_d, b_d));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return bw_err_check();
}

int amgh_level_smooth(amgh_t* h, int level, int post, real* x, const real* b) {
  RC_TRY(check_ready(h));
  if (level < 0 || level >= (int)h->levels.size() || !x || !b) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  const int64_t n = h->levels[level]->n;
  real *xd = nullptr, *bd = nullptr;
  RC_TRY(dev_upload(&xd, x, n));
  int rc = dev_upload(&bd, b, n);
  if (rc == AMGH_OK) rc = level_smooth_enqueue(h, level, post, xd, bd);
  if (rc == AMGH_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = -1001;
  if (rc == AMGH_OK) rc = bw_err_check();
  if (rc == AMGH_OK && hipMemcpy(x, xd, sizeof(real) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = -1001;
  hipFree(xd); hipFree(bd);
  return rc;
}

// ---- stand-alone operators ------------------------------------------------
int amgh_csr_create(amgh_csr_t** opp, int device, int64_t nrows, int64_t ncols, const int32_t* rowptr,
                    const int32_t* col, const real* val) {
  if (!opp) return AMGH_EINVAL;
  *opp = nullptr;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(device));
  amgh_csr* op = new amgh_csr;
  int rc = csr_upload(op, device, nrows, ncols, rowptr, col, val);
  if (rc != AMGH_OK) { csr_free(op); delete op; return rc; }
  *opp = op;
  return AMGH_OK;
}
void amgh_csr_destroy(amgh_csr_t* op) {
  if (!op) return;
  hipSetDevice(op->device);
  csr_free(op);
  delete op;
}
int amgh_csr_prepare(amgh_csr_t* op, int jacobi, int gs) {
  if (!op) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  if (jacobi) RC_TRY(csr_ensure_diag(op, nullptr));
  if (gs) RC_TRY(csr_ensure_gs(op));
  HIP_TRY(hipDeviceSynchronize());
  return AMGH_OK;
}
int amgh_csr_spmv_d(amgh_csr_t* op, const real* x_d, real* y_d, void* stream) {
  if (!op || !x_d || !y_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_apply(op, M_SPMV, x_d, nullptr, y_d, (hipStream_t)stream);
}
int amgh_csr_residual_d(amgh_csr_t* op, const real* x_d, const real* b_d, real* r_d, void* stream) {
  if (!op || !x_d || !b_d || !r_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_apply(op, M_RESID, x_d, b_d, r_d, (hipStream_t)stream);
}
int amgh_csr_spmv_add_d(amgh_csr_t* op, const real* x_d, real* y_d, void* stream) {
  if (!op || !x_d || !y_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_apply(op, M_ADD, x_d, nullptr, y_d, (hipStream_t)stream);
}
int amgh_csr_jacobi_d(amgh_csr_t* op, double omega, const real* xin_d, const real* b_d, real* xout_d,
                      void* stream) {
  if (!op || !xin_d || !b_d || !xout_d || xin_d == xout_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_jacobi(op, omega, xin_d, b_d, xout_d, (hipStream_t)stream);
}
int amgh_csr_gs_d(amgh_csr_t* op, int backward, double omega, int is_sor, real* x_d, const real* b_d,
                  void* stream) {
  if (!op || !x_d || !b_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_gs_sweep(op, backward != 0, is_sor != 0, omega, x_d, b_d, (hipStream_t)stream);
}
int amgh_csr_gs_ex_d(amgh_csr_t* op, int backward, double omega, int is_sor, real* x_d, const real* b_d,
                     void* stream, int flags) {
  if (!op || !x_d || !b_d || (flags & ~AMGH_GS_REUSE_B)) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_gs_sweep(op, backward != 0, is_sor != 0, omega, x_d, b_d, (hipStream_t)stream, true, true, 1, false,
                      (flags & AMGH_GS_REUSE_B) != 0);
}

int amgh_gather_d(int device, int64_t n, const int32_t* idx_d, const real* src_d, real* dst_d, void* stream) {
  if (n < 0 || (n > 0 && (!idx_d || !src_d || !dst_d)) || n >= INT32_MAX) return AMGH_EINVAL;
  if (n == 0) return AMGH_OK;
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(gather_perm_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src_d, idx_d, dst_d,
                     (int)n, (int64_t)0, (int64_t)0);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}

int amgh_dot_d(int device, int64_t n, const real* x_d, const real* y_d, real* scratch_d, real* out,
               void* stream) {
  if (n < 0 || !scratch_d || !out || (n > 0 && (!x_d || !y_d))) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kRedBlocks, (n + kThreads - 1) / kThreads));
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(kThreads), 0, st, x_d, y_d, n, scratch_d + 1);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, st, scratch_d + 1, nb, scratch_d, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, scratch_d, sizeof(real), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return AMGH_OK;
}

// ---- device memory + timing helpers ----------------------------------------
int amgh_dev_alloc(int device, int64_t bytes, void** ptr_d) {
  if (!ptr_d || bytes < 0) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(device));
  char* p = nullptr;
  RC_TRY(dev_alloc(&p, bytes));
  *ptr_d = p;
  return AMGH_OK;
}
int amgh_dev_free(int device, void* ptr_d) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(ptr_d));
  return AMGH_OK;
}
int amgh_dev_upload(int device, void* dst_d, const void* src, int64_t bytes) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst_d, src, (size_t)bytes, hipMemcpyHostToDevice));
  return AMGH_OK;
}
int amgh_dev_download(int device, void* dst, const void* src_d, int64_t bytes) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, src_d, (size_t)bytes, hipMemcpyDeviceToHost));
  return AMGH_OK;
}
int amgh_dev_sync(int device) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipDeviceSynchronize());
  return bw_err_check();
}
void* amgh_stream(amgh_t* h) { return h ? (void*)h->stream : nullptr; }

int amgh_timer_begin(amgh_t* h) {
  if (!h) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipEventRecord(h->t0, h->stream));
  return AMGH_OK;
}
int amgh_timer_end(amgh_t* h, double* ms) {
  if (!h || !ms) return AMGH_EINVAL;
  HIP_TRY(hipEventRecord(h->t1, h->stream));
  HIP_TRY(hipEventSynchronize(h->t1));
  float f = 0.f;
  HIP_TRY(hipEventElapsedTime(&f, h->t0, h->t1));
  *ms = f;
  return AMGH_OK;
}

int amgh_bench_op(amgh_t* h, int level, int which, int reps, int warmup, double* avg_ms) {
  RC_TRY(check_ready(h));
  if (!avg_ms || reps <= 0 || warmup < 0) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  // which: 0 / 1 / 2 = A / P / R products, 3 = fused residual, 4 = pre-smoother; 5 / 6 / 7 = residual / P / R on the
  // LEVEL-ORDERED copies, exactly as the cycle launches them.  0-3 take the natural-order operator where the hierarchy
  // still holds it and the level-ordered one otherwise (vectors in that order: no gather / scatter in the timing).
  const int base_which = which <= AMGH_OP_R ? which : which == 6 ? AMGH_OP_P : which == 7 ? AMGH_OP_R : AMGH_OP_A;
  amgh_csr* op = which >= 5 ? nullptr : level_op(h, level, base_which);
  LoOp lo;
  if (!op && which != 4 && !level_lo_op(h, level, base_which, &lo)) return AMGH_EINVAL;
  if (!op && which == 4 && (level < 0 || level >= (int)h->levels.size())) return AMGH_EINVAL;
  real *x = nullptr, *y = nullptr, *b = nullptr;
  const int64_t nx = op ? std::max(op->ncols, op->nrows) : which == 4 ? h->levels[level]->n : std::max(lo.ncols, lo.nrows);
  RC_TRY(dev_alloc(&x, nx));
  RC_TRY(dev_alloc(&y, nx));
  RC_TRY(dev_alloc(&b, nx));
  // deterministic non-trivial contents (a zero fill would flatter DVFS)
  {
    std::vector<real> hx(nx);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int64_t i = 0; i < nx; ++i) {
      s += 0x9E3779B97F4A7C15ull;
      uint64_t z = s;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      hx[i] = (real)(z >> 11) * (1.0 / 9007199254740992.0);
    }
    hipMemcpy(x, hx.data(), sizeof(real) * nx, hipMemcpyHostToDevice);
    hipMemcpy(b, hx.data(), sizeof(real) * nx, hipMemcpyHostToDevice);
    hipMemcpy(y, hx.data(), sizeof(real) * nx, hipMemcpyHostToDevice);
  }
  int rc = AMGH_OK;
  auto run = [&]() -> int {
    if (which == 4) {
      if (level >= (int)h->levels.size()) return AMGH_EINVAL;
      return level_smooth_enqueue(h, level, 0, y, b);
    }
    const int mode = (which == 3 || which == 5) ? M_RESID : M_SPMV;
    if (op) return csr_apply(op, mode, x, b, y, h->stream);
    return raw_apply(mode, lo.rowptr, lo.col, lo.val, lo.nrows, x, lo.ncols, b, lo.nrows, y, lo.nrows, h->stream, 1, lo.cc);
    if (which == 4) {
      if (level >= (int)h->levels.size()) return AMGH_EINVAL;
      return level_smooth_enqueue(h, level, 0, y, b);
    }
    return AMGH_EINVAL;
  };
  for (int i = 0; i < warmup && rc == AMGH_OK; ++i) rc = run();
  if (rc == AMGH_OK) {
    hipEventRecord(h->t0, h->stream);
    for (int i = 0; i < reps && rc == AMGH_OK; ++i) rc = run();
    hipEventRecord(h->t1, h->stream);
    hipEventSynchronize(h->t1);
    float f = 0.f;
    hipEventElapsedTime(&f, h->t0, h->t1);
    *avg_ms = (double)f / reps;
  }
  hipFree(x); hipFree(y); hipFree(b);
  return rc;
}

// The wavefront-of-blocks sweep (gs_blocks.hpp) on the HOST, from the very records the device kernel reads: plan (block
// partition from monotone potentials, launches, packed rows), then launch by launch, block by block, step by step what
// gs_bw_packed_kernel does — external x entries snapshot at the block's start, rows of a step from the block's x, products
// and sums in entry order, the quotient as reciprocal + one fma correction (the division itself outside the normal
// range).  x, b: nrows entries in natural order; omega = 1: Gauss-Seidel, else SOR.  stats4 = {blocks, launches, sum over
// launches of the deepest block's steps, external columns}.  Returns AMGH_EUNSUPPORTED when the operator cannot be laid
// out (rows longer than the kernels take).  CPU tests compare it bit for bit with the scalar lexicographic sweep.
int amgh_debug_bw_sweep_host(int64_t nrows, const int32_t* rowptr, const int32_t* col, const real* val, int target_rows,
                             int backward, double omega, real* x, const real* b, int64_t* stats4) {
  if (nrows <= 0 || !rowptr || !x || !b || omega == 0.0 || target_rows < 1) return AMGH_EINVAL;
  const int64_t n = nrows;
  for (int32_t j = 0; j < rowptr[n]; ++j)
    if (col[j] < 0 || col[j] >= n) return AMGH_EINVAL;   // (square operators only: no halo columns here)
  bw::Params prm;
  prm.target_rows = target_rows;
  prm.threads = 2;
  bw::Plan P;
  try {
    if (!bw::plan<real>(n, rowptr, col, val, prm, &P)) return AMGH_EUNSUPPORTED;
  } catch (const std::exception&) {
    return AMGH_ENOMEM;
  }
  if (stats4) { stats4[0] = (int64_t)P.blocks.size(); stats4[1] = (int64_t)P.launch_ptr.size() - 1; stats4[2] = P.sum_depth; stats4[3] = P.ext_total; }
  // every row exactly once
  {
    std::vector<char> seen(n, 0);
    for (int64_t p = 0; p < n; ++p) { if (P.perm[p] < 0 || P.perm[p] >= n || seen[P.perm[p]]) return AMGH_ESTATE; seen[P.perm[p]] = 1; }
  }
  // what the chained kernel (one launch per sweep, gs_bw_chain_kernel) relies on: a block's external positions split at npre into
  // those BEFORE the block in block order — every one of them inside a block of its predecessor list, all with smaller tickets —
  // and those behind it — inside blocks of its successor list, all with larger tickets; the two lists are each other's transpose
  {
    const int32_t B = (int32_t)P.blocks.size();
    if ((int32_t)P.dep_ptr.size() != B + 1 || (int32_t)P.sdep_ptr.size() != B + 1) return AMGH_ESTATE;
    std::vector<int32_t> first(B);
    for (int32_t ob = 0; ob < B; ++ob) {
      first[ob] = P.blocks[ob].row0;
      if (ob && first[ob] != P.blocks[ob - 1].row0 + P.blocks[ob - 1].nrows) return AMGH_ESTATE;
    }
    auto block_of = [&](int32_t q) { return (int32_t)(std::upper_bound(first.begin(), first.end(), q) - first.begin()) - 1; };
    int64_t transposed = 0;
    for (int32_t ob = 0; ob < B; ++ob) {
      const bw::Desc& d = P.blocks[ob];
      if (d.npre < 0 || d.npre > d.next) return AMGH_ESTATE;
      for (int32_t e = 0; e < d.next; ++e) {
        const int32_t q = P.ext_col[d.ext0 + e];
        const bool near = e < d.npre;
        if (near ? !(q < d.row0) : !(q >= d.row0 + d.nrows)) return AMGH_ESTATE;
        if (e && !(P.ext_col[d.ext0 + e - 1] < q)) return AMGH_ESTATE;                       // sorted, each position once
        const int32_t bq = block_of(q);
        const std::vector<int32_t>& lst = near ? P.dep : P.sdep;
        const std::vector<int32_t>& ptr = near ? P.dep_ptr : P.sdep_ptr;
        if (!std::binary_search(lst.begin() + ptr[ob], lst.begin() + ptr[ob + 1], bq)) return AMGH_ESTATE;
      }
      for (int32_t e = P.dep_ptr[ob]; e < P.dep_ptr[ob + 1]; ++e) {
        const int32_t pb = P.dep[e];
        if (pb < 0 || pb >= ob) return AMGH_ESTATE;                                          // smaller tickets only
        if (std::binary_search(P.sdep.begin() + P.sdep_ptr[pb], P.sdep.begin() + P.sdep_ptr[pb + 1], ob)) ++transposed;
      }
      for (int32_t e = P.sdep_ptr[ob]; e < P.sdep_ptr[ob + 1]; ++e)
        if (P.sdep[e] <= ob || P.sdep[e] >= B) return AMGH_ESTATE;
    }
    if (transposed != (int64_t)P.dep.size() || P.dep.size() != P.sdep.size()) return AMGH_ESTATE;
  }
  std::vector<real> xp(n), bp(n);
  for (int64_t p = 0; p < n; ++p) { xp[p] = x[P.perm[p]]; bp[p] = b[P.perm[p]]; }
  const int nl = (int)P.launch_ptr.size() - 1;
  const real om = (real)omega;
  for (int s = 0; s < nl; ++s) {
    const int l = backward ? nl - 1 - s : s;
    // the blocks of a launch are independent: each works on a snapshot of what the launch started from
    const std::vector<real> x0 = xp;
    for (int32_t ob = P.launch_ptr[l]; ob < P.launch_ptr[l + 1]; ++ob) {
      const bw::Desc& d = P.blocks[ob];
      const size_t rs = bw::Packed<real>::row_bytes(d.maxk);
      const int nvc = bw::Packed<real>::nvc(d.maxk);
      const unsigned char* rec = P.rec.data() + (size_t)d.rec * 16;
      const uint16_t* stp = (const uint16_t*)(rec + (size_t)d.nrows * rs);
      std::vector<real> xl(d.nrows + d.next + 1);
      for (int32_t p = 0; p < d.nrows; ++p) xl[p] = x0[d.row0 + p];
      for (int32_t e = 0; e < d.next; ++e) xl[d.nrows + e] = x0[P.ext_col[d.ext0 + e]];
      xl[d.nrows + d.next] = 0.0;
      for (int k = 0; k < d.nlev; ++k) {
        const int st = backward ? d.nlev - 1 - k : k;
        std::vector<real> xn;   // rows of a step are computed from the same state
        for (int32_t p = stp[st]; p < stp[st + 1]; ++p) {
          const real* v = (const real*)(rec + (size_t)p * rs);
          const uint16_t* cc = (const uint16_t*)(rec + (size_t)p * rs + (size_t)16 * nvc);
          real acc = 0.0;
          for (int e = 0; e < d.maxk; ++e) acc += v[e] * xl[cc[e] / sizeof(real)];
          const real dg = v[d.maxk], rc = v[d.maxk + 1];
          real q = xl[p];
          if (dg != 0.0) {
            const real nn = bp[d.row0 + p] - acc;
            if (om != (real)1) q = ((real)1 - om) * xl[p] + (om / dg) * nn;
            else {
              q = nn * rc;
              const real rem = std::fma(-dg, q, nn);
              q = std::fma(rem, rc, q);
              const real an = nn < 0 ? -nn : nn;
              const bool safe = sizeof(real) == 8 ? (an > (real)1e-200 && an < (real)1e200) : (an > (real)1e-25 && an < (real)1e25);
              if (!(rc != 0.0 && safe)) q = nn / dg;
            }
          }
          xn.push_back(q);
        }
        for (int32_t p = stp[st]; p < stp[st + 1]; ++p) xl[p] = xn[p - stp[st]];
      }
      for (int32_t p = 0; p < d.nrows; ++p) xp[d.row0 + p] = xl[p];
    }
  }
  for (int64_t p = 0; p < n; ++p) x[P.perm[p]] = xp[p];
  return AMGH_OK;
}

// Host-only: one Gauss-Seidel sweep executed FROM THE DICTIONARY LAYOUT of the dataflow records (gs_flow.hpp FlowDict: column
// records chunk-major per step, the dictionary index in the publish word, every block's distinct value rows) — the blocks in
// ticket order, a block's steps in walking order, a row's values out of its block's dictionary.  CPU tests compare it bit for
// bit with the scalar lexicographic sweep.  stats5 = {1 if the operator has the layout, dictionary rows in all, largest
// dictionary, bytes of column records + dictionaries, bytes of the plain records}.  AMGH_EUNSUPPORTED: no block layout /
// no dataflow layout (pattern not structurally symmetric); without the dictionary layout x is left untouched (stats5[0] = 0).
int amgh_debug_bw_dict_sweep_host(int64_t nrows, const int32_t* rowptr, const int32_t* col, const real* val, int target_rows,
                                  int backward, real* x, const real* b, int64_t* stats5) {
  if (nrows <= 0 || !rowptr || !x || !b || target_rows < 1) return AMGH_EINVAL;
  const int64_t n = nrows;
  for (int32_t j = 0; j < rowptr[n]; ++j)
    if (col[j] < 0 || col[j] >= n) return AMGH_EINVAL;
  bw::Params prm;
  prm.target_rows = target_rows;
  prm.threads = 2;
  bw::Plan P;
  bw::Flow F;
  try {
    if (!bw::plan<real>(n, rowptr, col, val, prm, &P)) return AMGH_EUNSUPPORTED;
    if (!bw::structurally_symmetric(n, rowptr, col, 2) || !bw::flow_build<real>(P, 2, &F)) return AMGH_EUNSUPPORTED;
  } catch (const std::exception&) {
    return AMGH_ENOMEM;
  }
  const int32_t B = (int32_t)P.blocks.size();
  if (stats5) {
    int64_t rows = 0, mx = 0;
    if (F.dc.on) for (int32_t e : F.dc.ent) { const int64_t r = (int64_t)((uint32_t)e >> 24) + 1; rows += r; mx = std::max(mx, r); }
    stats5[0] = F.dc.on ? 1 : 0; stats5[1] = rows; stats5[2] = mx;
    stats5[3] = (int64_t)(F.dc.crec.size() + F.dc.dict.size()); stats5[4] = (int64_t)F.srec.size();
  }
  if (!F.dc.on) return AMGH_OK;
  std::vector<real> xp(n), bp(n);
  for (int64_t p = 0; p < n; ++p) { xp[p] = x[P.perm[p]]; bp[p] = b[P.perm[p]]; }
  for (int32_t t = 0; t < B; ++t) {
    const int32_t ob = backward ? B - 1 - t : t;
    const bw::Desc& d = P.blocks[ob];
    const bw::FlowDesc& f = F.fd[ob];
    const int nvc = bw::Packed<real>::nvc(d.maxk), ncc = bw::Packed<real>::ncc(d.maxk), ns = d.nlev;
    const int cdw = ((d.maxk + 1) / 2 - 1) % 4 + 1;
    const unsigned char* crec = F.dc.crec.data() + (size_t)d.row0 * (size_t)ncc * 16;
    const int32_t de = F.dc.ent[(size_t)ob];
    const unsigned char* dict = F.dc.dict.data() + (size_t)(de & 0xffffff) * 16;
    const int drows = (int)((uint32_t)de >> 24) + 1;
    const uint32_t* ax = F.aux.data() + f.aux + (backward ? ns + 1 : 0);
    std::vector<real> xl((size_t)d.nrows + d.next + 1);
    for (int32_t p = 0; p < d.nrows; ++p) xl[p] = xp[d.row0 + p];
    for (int32_t e = 0; e < d.next; ++e) xl[d.nrows + e] = xp[P.ext_col[d.ext0 + e]];
    xl[(size_t)d.nrows + d.next] = 0.0;
    for (int k = 0; k < ns; ++k) {
      const uint32_t w = ax[k];
      const int r0 = (int)(w & ((1u << bw::kStepRowBits) - 1)), nr = (int)((w >> bw::kStepRowBits) & ((1u << bw::kStepCntBits) - 1));
      std::vector<real> xn((size_t)nr);
      for (int q = 0; q < nr; ++q) {
        uint16_t cc[64];
        int32_t pw = 0;
        for (int c = 0; c < ncc; ++c) {
          const unsigned char* ch = crec + ((size_t)ncc * r0 + (size_t)c * nr + q) * 16;
          std::memcpy(cc + 8 * c, ch, 16);
          if (c == ncc - 1) std::memcpy(&pw, ch + 4 * cdw, 4);
        }
        const int idx = (int)(((uint32_t)pw >> bw::kDictIdxShift) & 0xffu);
        if (idx >= drows) return AMGH_ESTATE;
        if ((pw & ~(0xff << bw::kDictIdxShift)) != F.pub[(size_t)d.row0 + r0 + q]) return AMGH_ESTATE;   // the publish word rides unchanged
        const real* v = (const real*)(dict + (size_t)idx * nvc * 16);
        real acc = 0.0;
        for (int e = 0; e < d.maxk; ++e) acc += v[e] * xl[cc[e] / sizeof(real)];
        const real dg = v[d.maxk], rc = v[d.maxk + 1];
        real qv = xl[(size_t)r0 + q];
        if (dg != 0.0) {
          const real nn = bp[d.row0 + r0 + q] - acc;
          qv = nn * rc;
          const real rem = std::fma(-dg, qv, nn);
          qv = std::fma(rem, rc, qv);
          const real an = nn < 0 ? -nn : nn;
          const bool safe = sizeof(real) == 8 ? (an > (real)1e-200 && an < (real)1e200) : (an > (real)1e-25 && an < (real)1e25);
          if (!(rc != 0.0 && safe)) qv = nn / dg;
        }
        xn[(size_t)q] = qv;
      }
      for (int q = 0; q < nr; ++q) xl[(size_t)r0 + q] = xn[(size_t)q];
    }
    for (int32_t p = 0; p < d.nrows; ++p) xp[d.row0 + p] = xl[p];
  }
  for (int64_t p = 0; p < n; ++p) x[P.perm[p]] = xp[p];
  return AMGH_OK;
}

// Host-only emulation of ONE merged-level Gauss-Seidel sweep (no device work: usable without a GPU).  Runs the same
// construction the device schedules use — dependency levels, level order, groups of m levels made independent by
// substitution, pre-pass over the other triangle — and then applies the composite rows group by group on the host.
// x: ncols entries (in/out, columns >= nrows are frozen halo values), b: nrows entries.  CPU tests compare it with
// the scalar lexicographic sweep.  Returns the number of groups, or a negative error code.
int amgh_debug_merged_sweep_host(int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* col,
                                 const real* val, int m, int backward, double omega, real* x, const real* b) {
  if (nrows < 0 || ncols < nrows || !rowptr || m < 1 || !x || !b || omega == 0.0) return AMGH_EINVAL;
  const int64_t n = nrows;
  HostLevelCsr base;
  std::vector<int32_t> perm;
  level_order(n, ncols, rowptr, col, val, base, perm);
  // SOR with factor omega = the triangular solve with diagonal D / omega and the diagonal term
  // ((1 - omega) / omega) D x_old on the right-hand side; omega = 1 is Gauss-Seidel
  HostLevelCsr scaled = base;
  for (real& dd : scaled.pdiag) dd /= omega;
  const real shift = (1.0 - omega) / omega;
  MergeResult R = merge_build(scaled, ncols, m, backward != 0);
  if (R.max_row > kBigSlot) return AMGH_EUNSUPPORTED;
  const HostLevelCsr& S = R.sys;
  std::vector<real> ext(ncols + n);
  for (int64_t c = 0; c < ncols; ++c) ext[c] = c < n ? x[perm[c]] : x[c];
  // pre-pass: s = b - T x over the triangle this direction does not substitute over (+ halo columns, + diagonal term)
  std::vector<int32_t> lev_of(n);
  for (int l = 0; l < base.nlev; ++l)
    for (int32_t p = base.lvl_ptr[l]; p < base.lvl_ptr[l + 1]; ++p) lev_of[p] = l;
  for (int64_t p = 0; p < n; ++p) {
    real s = b[perm[p]];
    for (int32_t j = base.prow[p]; j < base.prow[p + 1]; ++j) {
      const int32_t c = base.pcol[j];
      if (c == p) continue;
      const bool other = c >= n || (backward ? lev_of[c] < lev_of[p] : lev_of[c] > lev_of[p]);
      if (other) s -= base.pval[j] * ext[c];
    }
    if (base.pdiag[p] != 0.0) s += shift * base.pdiag[p] * ext[p];
    ext[ncols + p] = s;
  }
  // groups in sweep order; rows of a group only read earlier groups and s, so their order does not matter
  for (int k = 0; k < S.nlev; ++k) {
    const int q = backward ? S.nlev - 1 - k : k;
    std::vector<real> xn(S.lvl_ptr[q + 1] - S.lvl_ptr[q]);
    for (int32_t p = S.lvl_ptr[q]; p < S.lvl_ptr[q + 1]; ++p) {
      real acc = 0.0;
      for (int32_t j = S.prow[p]; j < S.prow[p + 1]; ++j) acc += S.pval[j] * ext[S.pcol[j]];
      xn[p - S.lvl_ptr[q]] = S.pdiag[p] != 0.0 ? (ext[ncols + p] - acc) / S.pdiag[p] : ext[p];
    }
    for (int32_t p = S.lvl_ptr[q]; p < S.lvl_ptr[q + 1]; ++p) ext[p] = xn[p - S.lvl_ptr[q]];  // written after all reads
  }
  for (int64_t p = 0; p < n; ++p) x[perm[p]] = ext[p];
  return S.nlev;
}

int amgh_debug_set_tunable(const char* name, int value) {
  if (!name) return AMGH_EINVAL;
  if (!strcmp(name, "gs_block_target")) g_gs_block_target = value;
  else if (!strcmp(name, "gs_min_rows")) g_gs_min_rows = value;
  else if (!strcmp(name, "gs_nnz_per_wg")) g_gs_nnz_per_wg = value;
  else if (!strcmp(name, "gs_threads")) g_gs_threads = value;
  else if (!strcmp(name, "gs_block_inverse")) g_gs_block_inverse = value;
  else if (!strcmp(name, "gs_slots")) g_gs_slots = value;
  else if (!strcmp(name, "gs_xcd_map")) g_gs_xcd_map = value;
  else if (!strcmp(name, "gs_lpr")) g_gs_lpr = value;
  else if (!strcmp(name, "gs_il")) g_gs_il = value;
  else if (!strcmp(name, "trim_coded")) g_trim_coded = value;
  else if (!strcmp(name, "gs_wave_quad")) g_gs_wave_quad = value;
  else if (!strcmp(name, "pcg_fused")) g_pcg_fused = value;
  else if (!strcmp(name, "gs_tri_rb")) g_gs_tri_rb = value;
  else if (!strcmp(name, "gs_tri_rb1")) g_gs_tri_rb1 = value;
  else if (!strcmp(name, "gs_dti_pre")) g_gs_dti_pre = value;
  else if (!strcmp(name, "stream_xcd")) g_stream_xcd = value;
  else if (!strcmp(name, "tail_dense_rows")) g_tail_dense_rows = value < 0 ? 0 : value;
  else if (!strcmp(name, "tail_dense")) { g_tail_dense = value; g_sched_epoch++; }   // (captured cycles hold the path they were captured on)
  else if (!strcmp(name, "tail_dense_batch")) g_tail_dense_batch = value;
  else if (!strcmp(name, "gs_lean")) g_gs_lean = value;
  else if (!strcmp(name, "gs_sell")) g_gs_sell = value;
  else if (!strcmp(name, "gs_sample")) g_gs_sample = value;
  else if (!strcmp(name, "gs_tiny")) g_gs_tiny = value;
  else if (!strcmp(name, "gs_bw")) g_gs_bw = value;
  else if (!strcmp(name, "gs_bw_rows")) g_gs_bw_rows = value;
  else if (!strcmp(name, "gs_bw_chain")) g_gs_bw_chain = value;
  else if (!strcmp(name, "gs_bw_flow")) g_gs_bw_flow = value;
  else if (!strcmp(name, "gs_bw_spin")) g_gs_bw_spin = value;
  else if (!strcmp(name, "gs_bw_nc")) g_gs_bw_nc = value;
  else if (!strcmp(name, "gs_bw_nrhs")) g_gs_bw_nrhs = value;
  else if (!strcmp(name, "gs_bw_skip_pub")) g_gs_bw_skip_pub = value;
  else if (!strcmp(name, "gs_bw_min_rows")) g_gs_bw_min_rows = value;
  else if (!strcmp(name, "gs_bw_two_min_rows")) g_gs_bw_two_min_rows = value;
  else if (!strcmp(name, "gs_dup_launch")) g_gs_dup_launch = value < 0 ? 0 : value > 16 ? 16 : value;   // (a repeat count: never negative)
  else if (!strcmp(name, "gs_flow_xzero")) g_gs_flow_xzero = value;
  else if (!strcmp(name, "gs_bw_grid")) g_gs_bw_grid = value < 0 ? 0 : value;
  else if (!strcmp(name, "gs_bw_grid_long")) g_gs_bw_grid_long = value < 0 ? 0 : value;
  else if (!strcmp(name, "gs_bw_dict")) g_gs_bw_dict = value != 0;
  else if (!strcmp(name, "gs_bw_inorder")) g_gs_bw_inorder = value != 0;
  else if (!strcmp(name, "stream_code")) g_stream_code = value != 0;
  else if (!strcmp(name, "gs_bw_relay")) g_gs_bw_relay = value == 0 ? 0 : BW_RELAY_W;   // (one count is instantiated)
  else if (!strcmp(name, "rhs_il")) g_rhs_il = value;
  else if (!strcmp(name, "jacobi_zero")) g_jacobi_zero = value;
  else if (!strcmp(name, "gs_ept")) g_gs_ept = value;
  else if (!strcmp(name, "gs_merge")) g_gs_merge = value;
  else if (!strcmp(name, "gs_merge_force")) g_gs_merge_force = value;
  else if (!strcmp(name, "gs_merge_force_maxn")) g_gs_merge_force_maxn = value;
  else if (!strcmp(name, "gs_zone")) g_gs_zone = value;
  else if (!strcmp(name, "gs_coarse_lo")) g_gs_coarse_lo = value;
  else if (!strcmp(name, "gs_dense_tri")) g_gs_dense_tri = value;
  else if (!strcmp(name, "gs_dense_blk")) g_gs_dense_blk = value;
  else if (!strcmp(name, "gs_zone_t0_ns")) g_gs_zone_t0_ns = value;
  else if (!strcmp(name, "gs_zone_floor_ns")) g_gs_zone_floor_ns = value;
  else if (!strcmp(name, "gs_bigslot")) g_gs_bigslot = value;
  else if (!strcmp(name, "gs_flip")) g_gs_flip = value;
  else if (!strcmp(name, "gs_keep_lo")) g_gs_keep_lo = value;
  else if (!strcmp(name, "gs_super")) g_gs_super = value;
  else if (!strcmp(name, "gs_block_pipe")) g_gs_block_pipe = value;
  else return AMGH_EINVAL;
  ++g_sched_epoch;   // captured cycles bake the execution path in: every handle captures again after a change
  return AMGH_OK;
}

int amgh_debug_chain_timing(int enable, unsigned long long* out8) {
  if (enable && !g_chain_tim) {
    HIP_TRY(hipMalloc((void**)&g_chain_tim, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(g_chain_tim, 0, 8 * sizeof(unsigned long long)));
  }
  if (out8 && g_chain_tim) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out8, g_chain_tim, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(g_chain_tim, 0, 8 * sizeof(unsigned long long)));
  }
  if (!enable && g_chain_tim) { hipFree(g_chain_tim); g_chain_tim = nullptr; }
  return AMGH_OK;
}

int amgh_profile_enable(amgh_t* h, int on) {
  if (!h) return AMGH_EINVAL;
  h->profile = on != 0;
  return AMGH_OK;
}
int amgh_profile_read(amgh_t* h, double* out, int reset) {
  if (!h || !out) return AMGH_EINVAL;
  RC_TRY(prof_flush(h));
  const int L1 = (int)h->levels.size() + 1;
  if ((int)h->prof_ms.size() != AMGH_T_COUNT * L1) h->prof_ms.assign(AMGH_T_COUNT * L1, 0.0);
  std::copy(h->prof_ms.begin(), h->prof_ms.end(), out);
  if (reset) std::fill(h->prof_ms.begin(), h->prof_ms.end(), 0.0);
  return AMGH_OK;
}
int amgh_set_use_graph(amgh_t* h, int on) {
  if (!h) return AMGH_EINVAL;
  h->use_graph = on != 0;
  h->graph_auto = false;
  return AMGH_OK;
}

}  // extern "C"

// the row-sharded cycle is written in amgh::real like the solve phase: both instances of the library carry it.  The GPU
// half of the setup is Float64 only (a Float32 hierarchy is built in Float64 and rounded once): the
// Float32 instance (-DAMGH_REAL=float -DAMGH_NO_SETUP) leaves it out.
#include "amghip_dist.hpp"
#ifndef AMGH_NO_SETUP
#include "amghip_setup.hpp"
#endif
