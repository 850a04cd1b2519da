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
#include "amghip_kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/amghip.h"

using namespace amgh;

#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return -(1000 + (int)e_);     \
  } while (0)
#define RC_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != AMGH_OK) return rc_; \
  } while (0)

namespace {

template <class T>
int dev_alloc(T** p, int64_t count) {
  *p = nullptr;
  if (count <= 0) count = 1;
  hipError_t e = hipMalloc((void**)p, sizeof(T) * (size_t)count);
  if (e == hipErrorOutOfMemory) return AMGH_ENOMEM;
  if (e != hipSuccess) return -(1000 + (int)e);
  return AMGH_OK;
}
template <class T>
int dev_upload(T** p, const T* src, int64_t count) {
  RC_TRY(dev_alloc(p, count));
  if (count > 0) HIP_TRY(hipMemcpy(*p, src, sizeof(T) * (size_t)count, hipMemcpyHostToDevice));
  return AMGH_OK;
}
inline int grid_for(int64_t n, int threads = 256) {
  int64_t g = (n + threads - 1) / threads;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, 256 * 8));
}

// Gauss-Seidel dependency schedule of one square operator.
struct GsSchedule {
  int nlev = 0;
  std::vector<int32_t> lvl_ptr;  // host, nlev+1
  int32_t* d_lvl_ptr = nullptr;
  // level-permuted copy of the smoother's matrix
  int32_t* rowptr = nullptr;
  int32_t* col = nullptr;
  double* val = nullptr;
  int32_t* perm = nullptr;
  int32_t* dpos = nullptr;
  double* diag = nullptr;
  i4_t* rowmeta = nullptr;  // per permuted row {start, end, diagonal position, original row}
  i4_t* desc = nullptr;     // per dependency level {first row, end row, first nnz, end nnz}
  double* bp = nullptr;     // right-hand side in dependency-level order (scratch)
  double* xp = nullptr;     // x in dependency-level order (scratch, ncols entries per right-hand-side column)
  int cols_alloc = 1;       // right-hand-side columns bp / xp currently hold
  int32_t* permx = nullptr; // perm extended by the identity over halo columns
  int64_t n = 0, ncols = 0;
  int64_t bytes = 0;
  struct Seg { int l0, l1; bool chain; int rows; int slot0, nslots; };  // dependency levels [l0, l1); launch shape
  std::vector<Seg> segs;
  // slot layout of the wide levels (gs_slot_kernel)
  int32_t* wcol = nullptr; double* wval = nullptr; int32_t* slot_row = nullptr; i4_t* wmeta = nullptr;
  // block-inverse path (small, densely coupled operators; see gs_block_kernel)
  struct Outer {
    int32_t* rowptr = nullptr; int32_t* col = nullptr; double* val = nullptr; double* tinv = nullptr;
    int32_t* near_ptr = nullptr; i2_t* near_pi = nullptr; double* near_val = nullptr;  // see gs_block_pipe_kernel
    // entries that reference blocks swept LATER (and the in-block other triangle): they read old x only, so
    // b - O_next x is one full-chip residual launch before the sequential sweep
    int32_t* nx_rowptr = nullptr; int32_t* nx_col = nullptr; double* nx_val = nullptr;
    // entries that reference EARLIER superblocks: final once that superblock is done, applied to the rows of a
    // superblock by one parallel launch before its sequential sweep
    int32_t* sp_rowptr = nullptr; int32_t* sp_col = nullptr; double* sp_val = nullptr;
  };
  Outer blk_f, blk_b;
  double* blk_diag = nullptr;
  double* blk_s = nullptr;  // b - O_next x (n entries per right-hand-side column)
  int nblk = 0;  // 0 = block path not used for this operator
  int super = 0; // blocks per superblock (0: the whole operator is one superblock)
  double blk_cond = 0.0;  // largest inf-norm condition estimate of an in-block triangle
  void free_dev() {
    for (Outer* o : {&blk_f, &blk_b}) {
      hipFree(o->rowptr); hipFree(o->col); hipFree(o->val); hipFree(o->tinv);
      hipFree(o->near_ptr); hipFree(o->near_pi); hipFree(o->near_val);
      hipFree(o->nx_rowptr); hipFree(o->nx_col); hipFree(o->nx_val);
      hipFree(o->sp_rowptr); hipFree(o->sp_col); hipFree(o->sp_val);
      *o = Outer();
    }
    hipFree(blk_diag); blk_diag = nullptr;
    hipFree(blk_s); blk_s = nullptr;
    hipFree(wcol); hipFree(wval); hipFree(slot_row); hipFree(wmeta); wcol = slot_row = nullptr; wval = nullptr; wmeta = nullptr;
    hipFree(d_lvl_ptr); hipFree(rowptr); hipFree(col); hipFree(val);
    hipFree(perm); hipFree(dpos); hipFree(diag); hipFree(rowmeta); hipFree(desc); hipFree(bp); hipFree(xp); hipFree(permx);
    d_lvl_ptr = rowptr = col = perm = dpos = nullptr; val = diag = bp = xp = nullptr; rowmeta = desc = nullptr; permx = nullptr;
  }
};

}  // namespace

struct amgh_csr {
  int device = 0;
  int64_t nrows = 0, ncols = 0, nnz = 0;
  int32_t* rowptr = nullptr;
  int32_t* col = nullptr;
  double* val = nullptr;
  // smoother metadata in natural row order (Jacobi), built on demand
  int32_t* dpos = nullptr;
  double* diag = nullptr;
  GsSchedule* gs = nullptr;
  int64_t bytes = 0;
};

namespace {

// launch-shape tunables of the per-level Gauss-Seidel launches (amgh_debug_set_tunable)
int g_gs_block_target = 256;   // aim for at least this many workgroups per wide level
int g_gs_min_rows = 4;         // but never fewer rows per workgroup than this
int g_gs_nnz_per_wg = 256;     // and about this many nonzeros per workgroup (one per thread)
int g_gs_threads = 256;
int g_gs_block_pipe = 1;        // software-pipelined block sweep (gs_block_pipe_kernel)
int g_gs_super = 8;             // block-inverse sweeps: blocks per superblock (0 = one launch for the whole operator); read at schedule build
int g_gs_xcd_map = 1;           // XCD-contiguous slot -> workgroup mapping in gs_slot_kernel
int g_gs_slots = 1;             // wide levels from the slot layout (0 = CSR stream kernel)
int g_gs_block_inverse = 1;     // block-inverse sweeps for small densely coupled operators (0 = exact order everywhere)        // workgroup size of the per-level launches (64 or 256)

unsigned long long* g_chain_tim = nullptr;  // diagnostics buffer (amgh_debug_chain_timing)

constexpr int kChainWidth = 1024;  // dependency levels at most this wide are chained

__global__ void find_diag_kernel(const int32_t* rowptr, const int32_t* col, const double* val, int n,
                                 int32_t* dpos, double* diag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int dp = -1;
  double d = 0.0;
  // the reference keeps the LAST matching entry (d = ifelse(i == row, val, d))
  for (int j = rowptr[i]; j < rowptr[i + 1]; ++j)
    if (col[j] == i) { dp = j; d = val[j]; }
  dpos[i] = dp;
  diag[i] = d;
}

int csr_upload(amgh_csr* op, int device, int64_t nrows, int64_t ncols, const int32_t* rowptr,
               const int32_t* col, const double* val) {
  if (nrows < 0 || ncols < 0 || !rowptr) return AMGH_EINVAL;
  if (nrows >= INT32_MAX || ncols >= INT32_MAX) return AMGH_EUNSUPPORTED;
  const int64_t nnz = rowptr[nrows];
  if (rowptr[0] != 0 || nnz < 0) return AMGH_EINVAL;
  if (nnz > 0 && (!col || !val)) return AMGH_EINVAL;
  op->device = device;
  op->nrows = nrows;
  op->ncols = ncols;
  op->nnz = nnz;
  RC_TRY(dev_upload(&op->rowptr, rowptr, nrows + 1));
  RC_TRY(dev_upload(&op->col, col, nnz));
  RC_TRY(dev_upload(&op->val, val, nnz));
  op->bytes = (nrows + 1) * 4 + nnz * 12;
  return AMGH_OK;
}

void csr_free(amgh_csr* op) {
  if (!op) return;
  hipFree(op->rowptr); hipFree(op->col); hipFree(op->val);
  hipFree(op->dpos); hipFree(op->diag);
  if (op->gs) { op->gs->free_dev(); delete op->gs; }
  op->rowptr = op->col = op->dpos = nullptr; op->val = op->diag = nullptr; op->gs = nullptr;
}

int csr_ensure_diag(amgh_csr* op, hipStream_t st) {
  if (op->dpos) return AMGH_OK;
  const int64_t n = std::min(op->nrows, op->ncols);
  RC_TRY(dev_alloc(&op->dpos, op->nrows));
  RC_TRY(dev_alloc(&op->diag, op->nrows));
  if (n < op->nrows) {
    HIP_TRY(hipMemsetAsync(op->dpos, 0xff, sizeof(int32_t) * op->nrows, st));
    HIP_TRY(hipMemsetAsync(op->diag, 0, sizeof(double) * op->nrows, st));
  }
  if (n > 0)
    hipLaunchKernelGGL(find_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, op->rowptr, op->col,
                       op->val, (int)n, op->dpos, op->diag);
  HIP_TRY(hipGetLastError());
  op->bytes += op->nrows * 12;
  return AMGH_OK;
}

// Block-inverse data of one sweep direction: the outer matrix (operator minus the in-block triangle
// and diagonal) and the dense inverses of the in-block triangles.
int blockgs_build_dir(GsSchedule::Outer* o, bool backward, int super, int64_t n, const int32_t* rowptr, const int32_t* col,
                      const double* val, const std::vector<double>& diag, double* max_cond) {
  const int B = kBlk;
  const int nblk = (int)((n + B - 1) / B);
  std::vector<int32_t> orow(n + 1, 0), ocol, xrow(n + 1, 0), xcol, prow(n + 1, 0), pcol;
  std::vector<double> oval, xval, pval;
  ocol.reserve(rowptr[n]); oval.reserve(rowptr[n]);
  xcol.reserve(rowptr[n]); xval.reserve(rowptr[n]);
  std::vector<double> tinv((size_t)nblk * B * B, 0.0), T((size_t)B * B);
  // near list: outer entries of a block that reference the block swept just before it (blk - 1 forward, blk + 1
  // backward), as {position in the block's outer range, column - first row of that block} + value
  std::vector<int32_t> near_ptr(nblk + 1, 0);
  std::vector<i2_t> near_pi;
  std::vector<double> near_val;
  for (int blk = 0; blk < nblk; ++blk) {
    const int64_t i0 = (int64_t)blk * B, i1 = std::min<int64_t>(i0 + B, n);
    const int64_t q0 = backward ? i0 + B : i0 - B;  // first row of the previously swept block
    // rows of this block's superblock: [s0, s1)
    const int64_t sb = super > 0 ? blk / super : 0;
    const int64_t s0 = super > 0 ? sb * super * B : 0, s1 = super > 0 ? std::min<int64_t>(n, (sb + 1) * super * B) : n;
    const size_t blk_first = ocol.size();
    std::fill(T.begin(), T.end(), 0.0);
    for (int64_t i = i0; i < i1; ++i) {
      const bool skip = diag[i] == 0.0;  // row without a usable diagonal keeps its x: T row = e_i
      for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
        const int32_t c = col[j];
        const bool in_tri = c >= i0 && c < i1 && (backward ? c >= i : c <= i);
        if (in_tri) {
          if (!skip) T[(size_t)(i - i0) * B + (c - i0)] += val[j];
        } else if (!skip && (c >= i0 && c < i1 ? true : (backward ? c < i0 : c >= i1))) {
          xcol.push_back(c); xval.push_back(val[j]);  // old x: in-block other triangle, or a block swept later
        } else if (!skip && (c < s0 || c >= s1)) {
          pcol.push_back(c); pval.push_back(val[j]);  // an earlier superblock
        } else if (!skip) {
          if (c >= q0 && c < q0 + B) {
            near_pi.push_back(i2_t{(int32_t)(ocol.size() - blk_first), (int32_t)(c - q0)});
            near_val.push_back(val[j]);
          }
          ocol.push_back(c); oval.push_back(val[j]);
        }
      }
      if (skip) T[(size_t)(i - i0) * B + (i - i0)] = 1.0;
      orow[i + 1] = (int32_t)ocol.size();
      xrow[i + 1] = (int32_t)xcol.size();
      prow[i + 1] = (int32_t)pcol.size();
    }
    near_ptr[blk + 1] = (int32_t)near_pi.size();
    for (int64_t i = i1; i < i0 + B; ++i) T[(size_t)(i - i0) * B + (i - i0)] = 1.0;  // padding rows
    // invert the triangle column by column (forward / backward substitution on the identity)
    double* X = tinv.data() + (size_t)blk * B * B;
    for (int c = 0; c < B; ++c) {
      if (!backward) {
        for (int i = c; i < B; ++i) {
          double s = (i == c) ? 1.0 : 0.0;
          for (int j = c; j < i; ++j) s -= T[(size_t)i * B + j] * X[(size_t)j * B + c];
          X[(size_t)i * B + c] = s / T[(size_t)i * B + i];
        }
      } else {
        for (int i = c; i >= 0; --i) {
          double s = (i == c) ? 1.0 : 0.0;
          for (int j = i + 1; j <= c; ++j) s -= T[(size_t)i * B + j] * X[(size_t)j * B + c];
          X[(size_t)i * B + c] = s / T[(size_t)i * B + i];
        }
      }
    }
    // inf-norm condition estimate of the triangle: an explicit inverse is only as accurate as
    // cond(T) * eps, the caller falls back to the exact-order sweeps when a block is badly conditioned
    double nt = 0.0, nx = 0.0;
    for (int i = 0; i < B; ++i) {
      double rt = 0.0, rx = 0.0;
      for (int j = 0; j < B; ++j) { rt += std::fabs(T[(size_t)i * B + j]); rx += std::fabs(X[(size_t)i * B + j]); }
      nt = std::max(nt, rt); nx = std::max(nx, rx);
    }
    const double cond = nt * nx;
    if (!(cond <= *max_cond)) *max_cond = std::isfinite(cond) ? cond : 1e300;
  }
  RC_TRY(dev_upload(&o->rowptr, orow.data(), n + 1));
  RC_TRY(dev_upload(&o->col, ocol.data(), (int64_t)ocol.size()));
  RC_TRY(dev_upload(&o->val, oval.data(), (int64_t)oval.size()));
  RC_TRY(dev_upload(&o->tinv, tinv.data(), (int64_t)tinv.size()));
  RC_TRY(dev_upload(&o->near_ptr, near_ptr.data(), (int64_t)near_ptr.size()));
  RC_TRY(dev_upload(&o->near_pi, near_pi.data(), (int64_t)near_pi.size()));
  RC_TRY(dev_upload(&o->near_val, near_val.data(), (int64_t)near_val.size()));
  RC_TRY(dev_upload(&o->nx_rowptr, xrow.data(), n + 1));
  RC_TRY(dev_upload(&o->nx_col, xcol.data(), (int64_t)xcol.size()));
  RC_TRY(dev_upload(&o->nx_val, xval.data(), (int64_t)xval.size()));
  RC_TRY(dev_upload(&o->sp_rowptr, prow.data(), n + 1));
  RC_TRY(dev_upload(&o->sp_col, pcol.data(), (int64_t)pcol.size()));
  RC_TRY(dev_upload(&o->sp_val, pval.data(), (int64_t)pval.size()));
  return AMGH_OK;
}

// Build the dependency-level schedule from HOST arrays of the smoother matrix.
int gs_build(GsSchedule* g, int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* col,
             const double* val) {
  const int64_t n = nrows;
  std::vector<int32_t> lev(n, 0);
  int32_t maxlev = -1;
  for (int64_t i = 0; i < n; ++i) {
    int32_t li = lev[i];
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      const int32_t c = col[j];
      if (c < i) li = std::max(li, lev[c] + 1);
    }
    lev[i] = li;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      const int32_t c = col[j];
      if (c > i && c < n) lev[c] = std::max(lev[c], li + 1);
    }
    maxlev = std::max(maxlev, li);
  }
  g->nlev = (int)(maxlev + 1);
  g->lvl_ptr.assign(g->nlev + 1, 0);
  for (int64_t i = 0; i < n; ++i) g->lvl_ptr[lev[i] + 1]++;
  for (int l = 0; l < g->nlev; ++l) g->lvl_ptr[l + 1] += g->lvl_ptr[l];
  std::vector<int32_t> perm(n), next(g->lvl_ptr.begin(), g->lvl_ptr.end() - (g->nlev > 0 ? 1 : 0));
  if (g->nlev == 0) next.clear();
  for (int64_t i = 0; i < n; ++i) perm[next[lev[i]]++] = (int32_t)i;  // ascending row id inside a level
  std::vector<int32_t>().swap(lev);
  // x is kept in dependency-level order during the sweeps: position p holds x[perm[p]]; columns
  // beyond the square block (halo entries of a sharded operator) keep their place.  Each level
  // then reads and writes contiguous stretches of x (coalesced, TLB-friendly) instead of a
  // hyperplane scattered over the whole vector.
  std::vector<int32_t> inv(std::max<int64_t>(ncols, n));
  for (int64_t c = 0; c < (int64_t)inv.size(); ++c) inv[c] = (int32_t)c;
  for (int64_t p2 = 0; p2 < n; ++p2) inv[perm[p2]] = (int32_t)p2;
  const int64_t nnz = rowptr[n];
  std::vector<int32_t> prow(n + 1), pcol(nnz), pdpos(n);
  std::vector<double> pval(nnz), pdiag(n);
  int64_t w = 0;
  prow[0] = 0;
  for (int64_t p = 0; p < n; ++p) {
    const int32_t i = perm[p];
    int32_t dp = -1;
    double d = 0.0;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      pcol[w] = inv[col[j]];   // entries stay in the row's original column order (sum order)
      pval[w] = val[j];
      if (col[j] == i) { dp = (int32_t)w; d = val[j]; }
      ++w;
    }
    prow[p + 1] = (int32_t)w;
    pdpos[p] = dp;
    pdiag[p] = d;
  }
  RC_TRY(dev_upload(&g->rowptr, prow.data(), n + 1));
  RC_TRY(dev_upload(&g->col, pcol.data(), nnz));
  RC_TRY(dev_upload(&g->val, pval.data(), nnz));
  RC_TRY(dev_upload(&g->perm, perm.data(), n));
  RC_TRY(dev_upload(&g->dpos, pdpos.data(), n));
  RC_TRY(dev_upload(&g->diag, pdiag.data(), n));
  RC_TRY(dev_upload(&g->d_lvl_ptr, g->lvl_ptr.data(), g->nlev + 1));
  {
    std::vector<i4_t> meta(n), desc(g->nlev);
    for (int64_t p = 0; p < n; ++p) meta[p] = i4_t{prow[p], prow[p + 1], pdpos[p], perm[p]};
    for (int l = 0; l < g->nlev; ++l)
      desc[l] = i4_t{g->lvl_ptr[l], g->lvl_ptr[l + 1], prow[g->lvl_ptr[l]], prow[g->lvl_ptr[l + 1]]};
    RC_TRY(dev_upload(&g->rowmeta, meta.data(), n));
    RC_TRY(dev_upload(&g->desc, desc.data(), g->nlev));
  }
  RC_TRY(dev_alloc(&g->bp, n));
  g->ncols = std::max<int64_t>(ncols, n);
  RC_TRY(dev_alloc(&g->xp, g->ncols));
  {
    std::vector<int32_t> permx(g->ncols);
    for (int64_t c = 0; c < g->ncols; ++c) permx[c] = c < n ? perm[c] : (int32_t)c;
    RC_TRY(dev_upload(&g->permx, permx.data(), g->ncols));
  }
  g->n = n;
  g->bytes = (n + 1) * 4 + nnz * 12 + n * 16 + (g->nlev + 1) * 4 + n * 24 + g->nlev * 16 + g->ncols * 12;
  // Block-inverse path: worth it when level scheduling has degenerated (many more dependency levels
  // than index blocks) and the dense blocks stay small.
  {
    const int nblk = (int)((n + kBlk - 1) / kBlk);
    if (n >= 16 && n <= 262144 && g->nlev >= 3 * nblk) {
      std::vector<double> dg(n, 0.0);
      for (int64_t i = 0; i < n; ++i)
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
          if (col[j] == i) dg[i] = val[j];
      double max_cond = 0.0;
      g->super = (g_gs_super > 0 && nblk > g_gs_super) ? g_gs_super : 0;
      RC_TRY(blockgs_build_dir(&g->blk_f, false, g->super, n, rowptr, col, val, dg, &max_cond));
      RC_TRY(blockgs_build_dir(&g->blk_b, true, g->super, n, rowptr, col, val, dg, &max_cond));
      RC_TRY(dev_upload(&g->blk_diag, dg.data(), n));
      RC_TRY(dev_alloc(&g->blk_s, n));
      g->blk_cond = max_cond;
      if (getenv("AMGH_VERBOSE"))
        fprintf(stderr, "[amghip] n=%lld dependency levels=%d index blocks=%d max triangle cond=%.3g -> %s\n", (long long)n,
                g->nlev, nblk, max_cond, max_cond <= 1e4 ? "block-inverse sweeps" : "exact-order sweeps");
      // explicit triangle inverses lose ~cond * eps: keep the 1e-10 contract with margin
      if (max_cond <= 1e4) {
        g->nblk = nblk;
        g->bytes += 2 * ((int64_t)nblk * kBlk * kBlk * 8 + nnz * 12 + (n + 1) * 4) + n * 8;
      }
    }
  }
  // segments: runs of narrow dependency levels are chained in one workgroup
  // A dependency level is chained (stays inside one workgroup) when it has at most
  // one row per thread and its products fit one LDS pass; anything larger is worth
  // a launch of its own that spreads over the CUs.
  auto narrow = [&](int lv) {
    const int width = g->lvl_ptr[lv + 1] - g->lvl_ptr[lv];
    const int lnnz = prow[g->lvl_ptr[lv + 1]] - prow[g->lvl_ptr[lv]];
    return width <= kChainWidth && lnnz <= kChainLds;
  };
  // workgroup size class of a chained level: the smallest of 64 / 256 / 1024 threads with one
  // thread per row and at most ~4 nonzeros per thread (the per-level loop is instruction-issue bound:
  // more threads = the level's nonzeros spread over all four SIMDs of the CU)
  auto chain_class = [&](int lv) {
    const int width = g->lvl_ptr[lv + 1] - g->lvl_ptr[lv];
    const int lnnz = prow[g->lvl_ptr[lv + 1]] - prow[g->lvl_ptr[lv]];
    if (width <= 64 && lnnz <= 64 * 4) return 64;
    if (width <= 256 && lnnz <= 256 * 4) return 256;
    return 1024;
  };
  g->segs.clear();
  int l = 0;
  while (l < g->nlev) {
    if (narrow(l)) {
      // a segment = run of chained levels of one class; a class change only starts a new launch if
      // the new run is long enough to pay for it (each launch costs a few microseconds)
      int e = l + 1;
      int cls = chain_class(l);
      while (e < g->nlev && narrow(e)) {
        const int ce = chain_class(e);
        if (ce != cls) {
          int run = 1;  // length of the run of class ce starting at e
          while (e + run < g->nlev && narrow(e + run) && chain_class(e + run) == ce && run < 16) ++run;
          if (ce < cls && run < 16) { ++e; continue; }   // short dip to a smaller class: absorb it
          if (ce > cls && e - l < 16) { cls = ce; ++e; continue; }  // short prefix: promote the segment
          break;
        }
        ++e;
      }
      g->segs.push_back({l, e, true, cls, 0, 0});
      l = e;
    } else {
      // rows per workgroup for this level's launch: about one LDS pass of products per
      // workgroup, as many workgroups as the level can feed (the launch is latency-bound)
      const int width = g->lvl_ptr[l + 1] - g->lvl_ptr[l];
      const int lnnz = prow[g->lvl_ptr[l + 1]] - prow[g->lvl_ptr[l]];
      const double avg = std::max(1.0, (double)lnnz / width);
      g->segs.push_back({l, l + 1, false, (int)std::min(1e6, avg * 16.0), 0, -1});  // 16 x mean row length
      ++l;
    }
  }
  // slot layout for the wide levels whose rows all fit a slot
  {
    std::vector<int32_t> wcol, slot_row;
    std::vector<double> wval;
    std::vector<i4_t> wmeta(n, i4_t{0, 0, -1, 0});
    for (auto& sg : g->segs) {
      if (sg.chain) continue;
      const int ra = g->lvl_ptr[sg.l0], rb = g->lvl_ptr[sg.l0 + 1];
      bool fits = true;
      for (int p2 = ra; p2 < rb && fits; ++p2) fits = prow[p2 + 1] - prow[p2] <= kSlot;
      if (!fits) continue;
      sg.slot0 = (int)slot_row.size();
      int fill = kSlot;  // force a new slot for the level's first row
      for (int p2 = ra; p2 < rb; ++p2) {
        const int len = prow[p2 + 1] - prow[p2];
        if (fill + len > kSlot) {  // open a new slot (pad the previous one)
          wcol.resize(slot_row.size() * (size_t)kSlot, 0);
          wval.resize(slot_row.size() * (size_t)kSlot, 0.0);
          slot_row.push_back(p2);
          fill = 0;
        }
        const int32_t start = (int32_t)((slot_row.size() - 1) * (size_t)kSlot + fill);
        for (int32_t j = prow[p2]; j < prow[p2 + 1]; ++j) { wcol.push_back(pcol[j]); wval.push_back(pval[j]); }
        wmeta[p2] = i4_t{start, start + len, pdpos[p2] >= 0 ? start + (pdpos[p2] - prow[p2]) : -1, 0};
        fill += len;
      }
      sg.nslots = (int)slot_row.size() - sg.slot0;
    }
    if (!slot_row.empty()) {
      wcol.resize(slot_row.size() * (size_t)kSlot, 0);
      wval.resize(slot_row.size() * (size_t)kSlot, 0.0);
      // slot_row[s + 1] must close the last slot of every level: append per-level end markers by
      // storing, for each slot, its end row in a parallel array packed as slot_row2
      std::vector<int32_t> sr2(2 * slot_row.size());
      for (auto& sg : g->segs) {
        if (sg.chain || sg.nslots <= 0) continue;
        for (int q = 0; q < sg.nslots; ++q) {
          const int sidx = sg.slot0 + q;
          sr2[2 * sidx] = slot_row[sidx];
          sr2[2 * sidx + 1] = (q + 1 < sg.nslots) ? slot_row[sidx + 1] : g->lvl_ptr[sg.l0 + 1];
        }
      }
      RC_TRY(dev_upload(&g->wcol, wcol.data(), (int64_t)wcol.size()));
      RC_TRY(dev_upload(&g->wval, wval.data(), (int64_t)wval.size()));
      RC_TRY(dev_upload(&g->slot_row, sr2.data(), (int64_t)sr2.size()));
      RC_TRY(dev_upload(&g->wmeta, wmeta.data(), n));
      g->bytes += (int64_t)wcol.size() * 12 + (int64_t)sr2.size() * 4 + n * 16;
    }
  }
  return AMGH_OK;
}

int csr_ensure_gs(amgh_csr* op) {
  if (op->gs) return AMGH_OK;
  const int64_t n = op->nrows;
  std::vector<int32_t> rowptr(n + 1), col(op->nnz);
  std::vector<double> val(op->nnz);
  HIP_TRY(hipMemcpy(rowptr.data(), op->rowptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));
  if (op->nnz) {
    HIP_TRY(hipMemcpy(col.data(), op->col, sizeof(int32_t) * op->nnz, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(val.data(), op->val, sizeof(double) * op->nnz, hipMemcpyDeviceToHost));
  }
  GsSchedule* g = new GsSchedule;
  int rc = gs_build(g, n, op->ncols, rowptr.data(), col.data(), val.data());
  if (rc != AMGH_OK) { g->free_dev(); delete g; return rc; }
  op->gs = g;
  op->bytes += g->bytes;
  return AMGH_OK;
}

template <int MODE, class CFG = DefaultCfg>
int launch_stream(const StreamArgs& a0, hipStream_t st, int ncolv = 1) {
  const int nrows = a0.row_end - a0.row_begin;
  if (nrows <= 0) return AMGH_OK;
  StreamArgs a = a0;
  a.ncolv = ncolv;
  const int nb = (nrows + CFG::ROWS - 1) / CFG::ROWS;
  // multi-column launches: tiles padded to a multiple of 8, times ncolv (multi_column_block)
  const int64_t grid = (CFG::XCD || ncolv > 1) ? (int64_t)((nb + kNumXcd - 1) / kNumXcd) * kNumXcd * ncolv : nb;
  hipLaunchKernelGGL((csr_stream_kernel<MODE, CFG>), dim3((unsigned)grid), dim3(CFG::THREADS), 0, st, a);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}

// One dependency level of a Gauss-Seidel / SOR sweep: a latency-bound launch, so the
// rows are spread over many small workgroups (rows per workgroup chosen at schedule
// build time from the level's average row length).
template <int MODE>
int launch_gs_level(const StreamArgs& a, int rows, hipStream_t st, int ncolv = 1) {
  // latency-bound launch: prefer many small workgroups over few full ones — a CU's
  // texture-address unit serialises the x gathers of all its waves
  // `avg16` = 16 x the level's mean row length (schedule build time).  Measured on MI355X
  // (tools/gs_tune.py): about one nonzero per thread is the fastest shape at every level.
  const int width = a.row_end - a.row_begin;
  const int avg16 = std::max(16, rows);
  rows = 256;
  while (rows > g_gs_min_rows && (int64_t)rows * avg16 > (int64_t)g_gs_nnz_per_wg * 16) rows >>= 1;
  while (rows > g_gs_min_rows && width / rows < g_gs_block_target && (int64_t)rows * avg16 > 16 * 64) rows >>= 1;
  if (g_gs_threads == 64) {
    switch (rows) {
      case 8: return launch_stream<MODE, StreamCfg<64, 8, 2048, 1, false, false>>(a, st, ncolv);
      case 16: return launch_stream<MODE, StreamCfg<64, 16, 2048, 1, false, false>>(a, st, ncolv);
      case 32: return launch_stream<MODE, StreamCfg<64, 32, 2048, 1, false, false>>(a, st, ncolv);
      case 64: return launch_stream<MODE, StreamCfg<64, 64, 2048, 1, false, false>>(a, st, ncolv);
      default: break;
    }
  }
  switch (rows) {
    case 4: return launch_stream<MODE, StreamCfg<256, 4, 2048, 1, false, false>>(a, st, ncolv);
    case 8: return launch_stream<MODE, StreamCfg<256, 8, 2048, 1, false, false>>(a, st, ncolv);
    case 16: return launch_stream<MODE, StreamCfg<256, 16, 2048, 1, false, false>>(a, st, ncolv);
    case 32: return launch_stream<MODE, StreamCfg<256, 32, 2048, 1, false, false>>(a, st, ncolv);
    case 64: return launch_stream<MODE, StreamCfg<256, 64, 2048, 1, false, false>>(a, st, ncolv);
    case 128: return launch_stream<MODE, StreamCfg<256, 128, 2048, 1, false, false>>(a, st, ncolv);
    default: return launch_stream<MODE, StreamCfg<256, 256, 2048, 2, false, false>>(a, st, ncolv);
  }
}

// ncolv right-hand-side columns (x: ncols apart, y and b: nrows apart) in one launch
int csr_apply(const amgh_csr* op, int mode, const double* x, const double* b, double* y, hipStream_t st,
              int ncolv = 1) {
  StreamArgs a{};
  a.rowptr = op->rowptr; a.col = op->col; a.val = op->val;
  a.x = x; a.y = y; a.b = b;
  a.row_begin = 0; a.row_end = (int32_t)op->nrows;
  a.ldx = op->ncols; a.ldy = op->nrows; a.ldb = op->nrows;
  switch (mode) {
    case M_SPMV: return launch_stream<M_SPMV>(a, st, ncolv);
    case M_RESID: return launch_stream<M_RESID>(a, st, ncolv);
    case M_ADD: return launch_stream<M_ADD>(a, st, ncolv);
  }
  return AMGH_EINVAL;
}

int csr_jacobi(amgh_csr* op, double omega, const double* xin, const double* b, double* xout, hipStream_t st,
               int ncolv = 1) {
  RC_TRY(csr_ensure_diag(op, st));
  StreamArgs a{};
  a.rowptr = op->rowptr; a.col = op->col; a.val = op->val;
  a.x = xin; a.y = xout; a.b = b; a.dpos = op->dpos; a.diag = op->diag; a.omega = omega;
  a.row_begin = 0; a.row_end = (int32_t)op->nrows;
  a.ldx = op->ncols; a.ldy = op->nrows; a.ldb = op->nrows;
  return launch_stream<M_JACOBI>(a, st, ncolv);
}

template <int T, int PF>
int launch_chain_t(const ChainArgs& c, bool sor, bool ldsx, int nx, hipStream_t st, int ncolv) {
  if (sor && ldsx) hipLaunchKernelGGL((gs_chain_kernel<true, true, T, PF>), dim3(ncolv), dim3(T), 0, st, c, nx);
  else if (sor) hipLaunchKernelGGL((gs_chain_kernel<true, false, T, PF>), dim3(ncolv), dim3(T), 0, st, c, nx);
  else if (ldsx) hipLaunchKernelGGL((gs_chain_kernel<false, true, T, PF>), dim3(ncolv), dim3(T), 0, st, c, nx);
  else hipLaunchKernelGGL((gs_chain_kernel<false, false, T, PF>), dim3(ncolv), dim3(T), 0, st, c, nx);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
// threads = workgroup size class of the segment (64 / 256 / 1024), see gs_build
int launch_chain(const ChainArgs& c, bool sor, bool ldsx, int threads, int nx, hipStream_t st, int ncolv) {
  switch (threads) {
    case 64: return launch_chain_t<64, 4>(c, sor, ldsx, nx, st, ncolv);
    case 256: return launch_chain_t<256, 4>(c, sor, ldsx, nx, st, ncolv);
    default: return launch_chain_t<1024, 4>(c, sor, ldsx, nx, st, ncolv);  // PF = 8 spills at 1024 threads (128 VGPRs)
  }
}

template <int NCV>
int launch_slot_t(const SlotArgs& sa, bool sor, int grid, hipStream_t st) {
  if (sor) hipLaunchKernelGGL((gs_slot_kernel<true, NCV>), dim3(grid), dim3(kSlot), 0, st, sa);
  else hipLaunchKernelGGL((gs_slot_kernel<false, NCV>), dim3(grid), dim3(kSlot), 0, st, sa);
  return AMGH_OK;
}
int launch_slot(const SlotArgs& sa, bool sor, int ncv, int grid, hipStream_t st) {
  switch (ncv) {
    case 8: return launch_slot_t<8>(sa, sor, grid, st);
    case 4: return launch_slot_t<4>(sa, sor, grid, st);
    case 2: return launch_slot_t<2>(sa, sor, grid, st);
    default: return launch_slot_t<1>(sa, sor, grid, st);
  }
}

// One Gauss-Seidel / SOR sweep, forward or backward, exact lexicographic order.
// first: gather b and x into dependency-level order (once per smooth! call);
// last: scatter x back to natural order.  Between the two x lives in g->xp.
// ncolv > 1: x (ncols apart) and b (nrows apart) hold ncolv independent right-hand-side columns; every launch
// covers all of them (gridDim.y, or one workgroup per column in the single-workgroup kernels), so a block of
// right-hand sides costs the dependency-level latency chain once.
int csr_gs_sweep(amgh_csr* op, bool backward, bool sor, double omega, double* x, const double* b, hipStream_t st,
                 bool first = true, bool last = true, int ncolv = 1) {
  RC_TRY(csr_ensure_gs(op));
  GsSchedule* g = op->gs;
  if (g->n <= 0) return AMGH_OK;
  if (ncolv > g->cols_alloc) {  // grow the per-column scratch (first block solve on this operator)
    HIP_TRY(hipStreamSynchronize(st));
    hipFree(g->bp); hipFree(g->xp); g->bp = g->xp = nullptr;
    RC_TRY(dev_alloc(&g->bp, g->n * ncolv));
    RC_TRY(dev_alloc(&g->xp, g->ncols * ncolv));
    int64_t grown = 8 * (g->n + g->ncols) * (ncolv - g->cols_alloc);
    if (g->blk_s) {
      hipFree(g->blk_s); g->blk_s = nullptr;
      RC_TRY(dev_alloc(&g->blk_s, g->n * ncolv));
      grown += 8 * g->n * (ncolv - g->cols_alloc);
    }
    g->bytes += grown;
    op->bytes += grown;
    g->cols_alloc = ncolv;
  }
  if (g->nblk > 0 && g_gs_block_inverse && !sor) {
    // small densely coupled operator: n/128 sequential block steps in natural row order
    BlockArgs ba{};
    const GsSchedule::Outer& o = backward ? g->blk_b : g->blk_f;
    ba.rowptr = o.rowptr; ba.col = o.col; ba.val = o.val; ba.tinv = o.tinv; ba.diag = g->blk_diag;
    {  // s = b - O_next x: every entry read here keeps its old value during this sweep
      StreamArgs ra{};
      ra.rowptr = o.nx_rowptr; ra.col = o.nx_col; ra.val = o.nx_val;
      ra.x = x; ra.b = b; ra.y = g->blk_s;
      ra.row_begin = 0; ra.row_end = (int32_t)g->n;
      ra.ldx = g->n; ra.ldy = g->n; ra.ldb = g->n;
      // few, long rows: 64 rows per workgroup so that the launch still covers the chip
      RC_TRY((launch_stream<M_RESID, StreamCfg<256, 64, 4096, 2, false, false>>(ra, st, ncolv)));
    }
    ba.x = x; ba.b = g->blk_s; ba.n = (int32_t)g->n; ba.backward = backward ? 1 : 0;
    ba.ld = g->n;  // block path: square operator, x and b in natural order
    ba.tim = g_chain_tim;
    ba.near_ptr = o.near_ptr; ba.near_pi = o.near_pi; ba.near_val = o.near_val;
    const int S = g->super > 0 ? g->super : g->nblk;
    const int nsuper = (g->nblk + S - 1) / S;
    for (int q = 0; q < nsuper; ++q) {
      const int J = backward ? nsuper - 1 - q : q;
      ba.blk0 = J * S;
      ba.nblk = std::min(S, g->nblk - ba.blk0);
      if (q > 0) {  // s -= O_sp x on this superblock's rows: every superblock swept so far is final
        StreamArgs pa{};
        pa.rowptr = o.sp_rowptr; pa.col = o.sp_col; pa.val = o.sp_val;
        pa.x = x; pa.b = g->blk_s; pa.y = g->blk_s;
        pa.row_begin = ba.blk0 * kBlk; pa.row_end = (int32_t)std::min<int64_t>(g->n, (int64_t)(ba.blk0 + ba.nblk) * kBlk);
        pa.ldx = g->n; pa.ldy = g->n; pa.ldb = g->n;
        RC_TRY((launch_stream<M_RESID, StreamCfg<256, 16, 2048, 2, false, false>>(pa, st, ncolv)));
      }
      if (g_gs_block_pipe) hipLaunchKernelGGL(gs_block_pipe_kernel, dim3(ncolv), dim3(kPipeThreads), 0, st, ba);
      else hipLaunchKernelGGL(gs_block_kernel, dim3(ncolv), dim3(kBlkThreads), 0, st, ba);
    }
    HIP_TRY(hipGetLastError());
    return AMGH_OK;
  }
  if (first) {
    hipLaunchKernelGGL(gather_perm_kernel, dim3(grid_for(g->n), ncolv), dim3(256), 0, st, b, g->perm, g->bp, (int)g->n,
                       (int64_t)g->n, (int64_t)g->n);
    hipLaunchKernelGGL(gather_perm_kernel, dim3(grid_for(g->ncols), ncolv), dim3(256), 0, st, (const double*)x, g->permx,
                       g->xp, (int)g->ncols, (int64_t)g->ncols, (int64_t)g->ncols);
    HIP_TRY(hipGetLastError());
  }
  double* xp = g->xp;
  const int ns = (int)g->segs.size();
  for (int k = 0; k < ns; ++k) {
    const GsSchedule::Seg& s = g->segs[backward ? ns - 1 - k : k];
    if (s.chain) {
      ChainArgs c{};
      c.col = g->col; c.val = g->val; c.x = xp; c.bp = g->bp; c.diag = g->diag;
      c.rowmeta = g->rowmeta; c.desc = g->desc; c.omega = omega; c.tim = g_chain_tim;
      if (!backward) { c.lvl_begin = s.l0; c.lvl_end = s.l1; c.step = 1; }
      else { c.lvl_begin = s.l1 - 1; c.lvl_end = s.l0 - 1; c.step = -1; }
      c.ldx = g->ncols; c.ldb = g->n;
      const bool ldsx = g->ncols <= kChainLdsX;  // x (halo included) fits LDS
      RC_TRY(launch_chain(c, sor, ldsx, s.rows, (int)g->ncols, st, ncolv));
    } else if (s.nslots > 0 && g_gs_slots) {
      SlotArgs sa{};
      sa.wcol = g->wcol; sa.wval = g->wval; sa.slot_row = g->slot_row; sa.wmeta = g->wmeta;
      sa.diag = g->diag; sa.bp = g->bp; sa.x = xp; sa.omega = omega; sa.slot0 = s.slot0;
      sa.nslots = s.nslots; sa.xcd_map = g_gs_xcd_map;
      sa.ldx = g->ncols; sa.ldb = g->n;
      // columns per workgroup: the largest of 8 / 4 / 2 / 1 that divides the block size
      const int ncv = (ncolv % 8 == 0) ? 8 : (ncolv % 4 == 0) ? 4 : (ncolv % 2 == 0) ? 2 : 1;
      sa.ncolv = ncolv / ncv;
      const int grid = ((g_gs_xcd_map || sa.ncolv > 1) ? ((s.nslots + kNumXcd - 1) / kNumXcd) * kNumXcd : s.nslots) * sa.ncolv;
      RC_TRY(launch_slot(sa, sor, ncv, grid, st));
      HIP_TRY(hipGetLastError());
    } else {
      StreamArgs a{};
      a.rowptr = g->rowptr; a.col = g->col; a.val = g->val;
      a.x = xp; a.y = xp; a.b = g->bp; a.dpos = g->dpos; a.diag = g->diag; a.perm = nullptr; a.omega = omega;
      a.row_begin = g->lvl_ptr[s.l0]; a.row_end = g->lvl_ptr[s.l0 + 1];
      a.ldx = g->ncols; a.ldy = g->ncols; a.ldb = g->n;
      RC_TRY(sor ? launch_gs_level<M_SOR>(a, s.rows, st, ncolv) : launch_gs_level<M_GS>(a, s.rows, st, ncolv));
    }
  }
  if (last) {
    hipLaunchKernelGGL(scatter_perm_kernel, dim3(grid_for(g->n), ncolv), dim3(256), 0, st, (const double*)xp, g->perm, x,
                       (int)g->n, (int64_t)g->ncols, (int64_t)g->n);
    HIP_TRY(hipGetLastError());
  }
  return AMGH_OK;
}

struct Level {
  int64_t n = 0, nc = 0;
  amgh_csr A, S, P, R;
  bool has_S = false;  // S distinct from A
  amgh_smoother_t pre{}, post{};
  double *res = nullptr, *cx = nullptr, *cb = nullptr, *tmp = nullptr;
  amgh_csr* smat() { return has_S ? &S : &A; }
};

}  // namespace

struct amgh_handle {
  int device = 0;
  int nrhs = 1;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;  // the stream created by amgh_create (when an external one is in use)
  bool ext_stream = false;
  std::vector<Level*> levels;
  // coarsest
  int64_t ncoarse = -1;
  amgh_csr finalA;
  bool has_finalA = false;
  double* coarse_op = nullptr;
  amgh_coarse_fn coarse_fn = nullptr;  // host-side pluggable coarse solver
  void* coarse_user = nullptr;
  std::vector<double> coarse_hb, coarse_hx;
  double* res_final = nullptr;  // res_vecs[1] when there are no levels
  bool finalized = false;
  // reductions / scalars
  double* partial = nullptr;   // kRedBlocks
  double* scal = nullptr;      // device scalars: [0] norm/dot out, [1] rho, [2] rho_prev, [3] alpha, [4] beta, [5] tmp
  // internal fine-level buffers for host-pointer entry points and PCG
  double *x0 = nullptr, *b0 = nullptr, *pc_r = nullptr, *pc_c = nullptr, *pc_u = nullptr;
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
  // hipGraph cache of whole cycles, keyed by the (x, b, cycle) they were captured on
  struct CycleGraph { const double* x; const double* b; int cyc; hipGraphExec_t exec; };
  std::vector<CycleGraph> graphs;
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

int vec_fill(amgh_t* h, double* x, int64_t n, double v) {
  if (n <= 0) return AMGH_OK;
  if (v == 0.0) { HIP_TRY(hipMemsetAsync(x, 0, sizeof(double) * n, h->stream)); return AMGH_OK; }
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, x, n, v);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
int vec_copy(amgh_t* h, double* dst, const double* src, int64_t n) {
  if (n <= 0 || dst == src) return AMGH_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToDevice, h->stream));
  return AMGH_OK;
}
// out_d[0] = sum x*y (op 0) or sqrt(sum x*x) (op 1)
int vec_dot(amgh_t* h, const double* x, const double* y, int64_t n, double* out_d, int op) {
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kRedBlocks, (n + kThreads - 1) / kThreads));
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(kThreads), 0, h->stream, x, y, n, h->partial);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, h->stream, h->partial, nb, out_d, op);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
int vec_norm_host(amgh_t* h, const double* x, int64_t n, double* out) {
  RC_TRY(vec_dot(h, x, x, n, h->scal, 1));
  HIP_TRY(hipMemcpyAsync(out, h->scal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return AMGH_OK;
}

int coarse_solve(amgh_t* h, double* x, const double* b) {
  const int n = (int)h->ncoarse;
  if (n <= 0) return AMGH_OK;
  if (h->coarse_fn) {
    HIP_TRY(hipMemcpyAsync(h->coarse_hb.data(), b, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->coarse_fn(h->coarse_user, h->coarse_hb.data(), h->coarse_hx.data(), n) != 0) return AMGH_ESTATE;
    HIP_TRY(hipMemcpyAsync(x, h->coarse_hx.data(), sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return AMGH_OK;
  }
  hipLaunchKernelGGL(dense_gemv_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->coarse_op, b, x, n);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}

// smooth!(x, smoother, b).  `xc` is the buffer that currently holds x; Jacobi
// sweeps ping-pong between xc and xo (swapped in place).
int smooth(amgh_t* h, Level* L, const amgh_smoother_t& s, double*& xc, double*& xo, const double* b, int ncolv = 1) {
  amgh_csr* M = L->smat();
  for (int it = 0; it < s.iter; ++it) {
    switch (s.kind) {
      case AMGH_SMOOTH_NONE: break;
      case AMGH_SMOOTH_JACOBI:
        RC_TRY(csr_jacobi(M, s.omega, xc, b, xo, h->stream, ncolv));
        std::swap(xc, xo);
        break;
      case AMGH_SMOOTH_GS:
      case AMGH_SMOOTH_SOR: {
        const bool sor = s.kind == AMGH_SMOOTH_SOR;
        // x and b enter dependency-level order before the first sweep of this smooth! call and x
        // returns to natural order after the last one
        const bool sym = s.sweep == AMGH_SWEEP_SYMMETRIC;
        const bool first_it = (it == 0), last_it = (it == s.iter - 1);
        if (s.sweep == AMGH_SWEEP_FORWARD || sym)
          RC_TRY(csr_gs_sweep(M, false, sor, s.omega, xc, b, h->stream, first_it, last_it && !sym, ncolv));
        if (s.sweep == AMGH_SWEEP_BACKWARD || sym)
          RC_TRY(csr_gs_sweep(M, true, sor, s.omega, xc, b, h->stream, first_it && !sym, last_it, ncolv));
        break;
      }
      default: return AMGH_EINVAL;
    }
  }
  return AMGH_OK;
}

int cycle(amgh_t* h, int l, double* x, const double* b, int cyc);

// __solve_next! (multilevel.jl:200-212)
int cycle_next(amgh_t* h, int l, double* x, const double* b, int cyc) {
  switch (cyc) {
    case AMGH_CYCLE_V: return cycle(h, l, x, b, AMGH_CYCLE_V);
    case AMGH_CYCLE_W:
      RC_TRY(cycle(h, l, x, b, AMGH_CYCLE_W));
      return cycle(h, l, x, b, AMGH_CYCLE_W);
    case AMGH_CYCLE_F:
      RC_TRY(cycle(h, l, x, b, AMGH_CYCLE_F));
      return cycle(h, l, x, b, AMGH_CYCLE_V);
  }
  return AMGH_EINVAL;
}

// __solve! (multilevel.jl:214-239).  x, b: n x bs column-major (bs = workspace block size,
// multilevel.jl:28-59).  The reference loops the columns inside every operator (smoother.jl:77,117); here
// each launch covers all bs columns (one grid row / one workgroup per column): per column the arithmetic and
// its order are those of the single-column path, so the results are bitwise the same.
int cycle(amgh_t* h, int l, double* x, const double* b, int cyc) {
  Level* L = h->levels[l];
  const int bs = h->nrhs;
  const int64_t n = L->n, nc = L->nc;
  double* xc = x;
  double* xo = L->tmp;
  {
    ProfScope p(h, AMGH_T_PRESMOOTH, l);
    RC_TRY(smooth(h, L, L->pre, xc, xo, b, bs));
  }
  {
    ProfScope p(h, AMGH_T_RESIDUAL, l);
    RC_TRY(csr_apply(&L->A, M_RESID, xc, b, L->res, h->stream, bs));
  }
  {
    ProfScope p(h, AMGH_T_RESTRICT, l);
    RC_TRY(csr_apply(&L->R, M_SPMV, L->res, nullptr, L->cb, h->stream, bs));
  }
  RC_TRY(vec_fill(h, L->cx, nc * bs, 0.0));
  if (l == (int)h->levels.size() - 1) {
    ProfScope p(h, AMGH_T_COARSE, l + 1);
    for (int c = 0; c < bs; ++c) RC_TRY(coarse_solve(h, L->cx + c * nc, L->cb + c * nc));
  } else {
    RC_TRY(cycle_next(h, l + 1, L->cx, L->cb, cyc));
  }
  {
    ProfScope p(h, AMGH_T_PROLONG, l);
    RC_TRY(csr_apply(&L->P, M_ADD, L->cx, nullptr, xc, h->stream, bs));
  }
  {
    ProfScope p(h, AMGH_T_POSTSMOOTH, l);
    RC_TRY(smooth(h, L, L->post, xc, xo, b, bs));
  }
  if (xc != x) RC_TRY(vec_copy(h, x, xc, n * bs));
  return AMGH_OK;
}

// one application of the hierarchy: a cycle, or the coarse solve if no levels
int apply_once(amgh_t* h, double* x, const double* b, int cyc) {
  if (h->levels.empty()) {
    ProfScope p(h, AMGH_T_COARSE, 0);
    for (int c = 0; c < h->nrhs; ++c) RC_TRY(coarse_solve(h, x + c * h->ncoarse, b + c * h->ncoarse));
    return AMGH_OK;
  }
  return cycle(h, 0, x, b, cyc);
}

// apply_once through a captured hipGraph: a cycle is thousands of short,
// launch-bound kernels (one per Gauss-Seidel dependency level), replaying them from a
// graph takes the host launch cost off the critical path.  Falls back to eager
// launches when profiling, with a host coarse solver, or if capture fails.
int apply_cycle(amgh_t* h, double* x, const double* b, int cyc) {
  if (!h->use_graph || h->profile || h->coarse_fn || h->levels.empty()) return apply_once(h, x, b, cyc);
  for (auto& g : h->graphs)
    if (g.x == x && g.b == b && g.cyc == cyc) {
      HIP_TRY(hipGraphLaunch(g.exec, h->stream));
      return AMGH_OK;
    }
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return apply_once(h, x, b, cyc);
  }
  const int rc = apply_once(h, x, b, cyc);
  const hipError_t e = hipStreamEndCapture(h->stream, &graph);
  if (rc != AMGH_OK || e != hipSuccess || !graph) {
    if (graph) hipGraphDestroy(graph);
    (void)hipGetLastError();
    if (rc != AMGH_OK) return rc;
    h->use_graph = false;  // capture unsupported here: stay eager
    return apply_once(h, x, b, cyc);
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (ei != hipSuccess) {
    (void)hipGetLastError();
    h->use_graph = false;
    return apply_once(h, x, b, cyc);
  }
  if (h->graphs.size() >= 6) {
    hipGraphExecDestroy(h->graphs.front().exec);
    h->graphs.erase(h->graphs.begin());
  }
  h->graphs.push_back({x, b, cyc, exec});
  HIP_TRY(hipGraphLaunch(exec, h->stream));
  return AMGH_OK;
}

int fine_residual(amgh_t* h, const double* x, const double* b, double* r) {
  if (h->levels.empty() && !h->has_finalA) return AMGH_ESTATE;
  const amgh_csr* A = h->levels.empty() ? &h->finalA : &h->levels[0]->A;
  RC_TRY(csr_apply(A, M_RESID, x, b, r, h->stream, h->nrhs));
  return AMGH_OK;
}
int fine_spmv(amgh_t* h, const double* x, double* y) {
  if (h->levels.empty()) {
    if (!h->has_finalA) return AMGH_ESTATE;
    return csr_apply(&h->finalA, M_SPMV, x, nullptr, y, h->stream);
  }
  return csr_apply(&h->levels[0]->A, M_SPMV, x, nullptr, y, h->stream);
}

// _solve! (multilevel.jl:158-198) on device pointers
int solve_dev(amgh_t* h, const double* b, double* x, int cyc, int maxiter, double abstol, double reltol,
              int calc_res, double* hist, int* iters) {
  const int64_t n = fine_n(h) * h->nrhs;  // norm(b) of an n x bs matrix is the Frobenius norm
  double normb = 0.0;
  RC_TRY(vec_norm_host(h, b, n, &normb));
  double normres = normb;
  if (normb != 0.0) abstol = std::max(reltol * normb, abstol);
  if (hist) hist[0] = normb;
  double* res = h->levels.empty() ? h->res_final : h->levels[0]->res;
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
  return AMGH_OK;
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
int pcg_dev(amgh_t* h, const double* b, double* x, int cyc, int use_precond, int maxiter, double abstol,
            double reltol, double* hist, int* iters) {
  const int64_t n = fine_n(h);
  double* r = h->pc_r; double* c = h->pc_c; double* u = h->pc_u;
  RC_TRY(vec_fill(h, x, n, 0.0));
  RC_TRY(vec_fill(h, u, n, 0.0));
  RC_TRY(vec_copy(h, r, b, n));
  double residual = 0.0;
  RC_TRY(vec_norm_host(h, r, n, &residual));
  const double tol = std::max(reltol * residual, abstol);
  if (hist) hist[0] = residual;
  const double one = 1.0;
  HIP_TRY(hipMemcpyAsync(h->scal + 1, &one, sizeof(double), hipMemcpyHostToDevice, h->stream));  // rho = 1
  int it = 0;
  while (it < maxiter && residual > tol) {
    if (use_precond) {
      RC_TRY(vec_fill(h, c, n, 0.0));
      RC_TRY(apply_cycle(h, c, r, cyc));
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
  return AMGH_OK;
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
  if (const char* e = getenv("AMGH_USE_GRAPH")) h->use_graph = (e[0] == '1');
  *hp = h;
  return AMGH_OK;
}

void amgh_destroy(amgh_t* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (Level* L : h->levels) {
    csr_free(&L->A); csr_free(&L->S); csr_free(&L->P); csr_free(&L->R);
    hipFree(L->res); hipFree(L->cx); hipFree(L->cb); hipFree(L->tmp);
    delete L;
  }
  csr_free(&h->finalA);
  hipFree(h->coarse_op); hipFree(h->res_final); hipFree(h->partial); hipFree(h->scal);
  hipFree(h->x0); hipFree(h->b0); hipFree(h->pc_r); hipFree(h->pc_c); hipFree(h->pc_u);
  for (auto& e : h->pending) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  for (auto& g : h->graphs) hipGraphExecDestroy(g.exec);
  if (h->t0) hipEventDestroy(h->t0);
  if (h->t1) hipEventDestroy(h->t1);
  if (h->own_stream) hipStreamDestroy(h->own_stream);
  else if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int amgh_push_level(amgh_t* h, int64_t n, int64_t nc, const int32_t* A_rowptr, const int32_t* A_col,
                    const double* A_val, const int32_t* S_rowptr, const int32_t* S_col, const double* S_val,
                    const int32_t* P_rowptr, const int32_t* P_col, const double* P_val, const int32_t* R_rowptr,
                    const int32_t* R_col, const double* R_val, const amgh_smoother_t* pre,
                    const amgh_smoother_t* post) {
  if (!h || n <= 0 || nc < 0 || !A_rowptr || !P_rowptr || !R_rowptr) return AMGH_EINVAL;
  if (!smoother_valid(pre) || !smoother_valid(post)) return AMGH_EINVAL;
  if (h->finalized) return AMGH_ESTATE;
  if (!h->levels.empty() && h->levels.back()->nc != n) return AMGH_EINVAL;
  if (P_rowptr[n] != R_rowptr[nc]) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  Level* L = new Level;
  L->n = n; L->nc = nc; L->pre = *pre; L->post = *post;
  int rc = csr_upload(&L->A, h->device, n, n, A_rowptr, A_col, A_val);
  if (rc == AMGH_OK && S_rowptr) {
    L->has_S = true;
    rc = csr_upload(&L->S, h->device, n, n, S_rowptr, S_col, S_val);
  }
  if (rc == AMGH_OK) rc = csr_upload(&L->P, h->device, n, nc, P_rowptr, P_col, P_val);
  if (rc == AMGH_OK) rc = csr_upload(&L->R, h->device, nc, n, R_rowptr, R_col, R_val);
  const bool need_gs = pre->kind == AMGH_SMOOTH_GS || pre->kind == AMGH_SMOOTH_SOR ||
                       post->kind == AMGH_SMOOTH_GS || post->kind == AMGH_SMOOTH_SOR;
  if (rc == AMGH_OK && need_gs) {
    // schedule built from the host arrays while we still have them
    amgh_csr* M = L->smat();
    GsSchedule* g = new GsSchedule;
    rc = L->has_S ? gs_build(g, n, n, S_rowptr, S_col, S_val) : gs_build(g, n, n, A_rowptr, A_col, A_val);
    if (rc == AMGH_OK) { M->gs = g; M->bytes += g->bytes; }
    else { g->free_dev(); delete g; }
  }
  if (rc != AMGH_OK) {
    csr_free(&L->A); csr_free(&L->S); csr_free(&L->P); csr_free(&L->R);
    delete L;
    return rc;
  }
  h->levels.push_back(L);
  return AMGH_OK;
}

int amgh_set_coarse(amgh_t* h, int64_t n, const int32_t* A_rowptr, const int32_t* A_col, const double* A_val,
                    const double* dense_op) {
  if (!h || n < 0 || !dense_op) return AMGH_EINVAL;
  if (h->finalized || h->ncoarse >= 0) return AMGH_ESTATE;
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

int amgh_set_coarse_host(amgh_t* h, int64_t n, const int32_t* A_rowptr, const int32_t* A_col, const double* A_val,
                         amgh_coarse_fn fn, void* user) {
  if (!h || n < 0 || !fn) return AMGH_EINVAL;
  if (h->finalized || h->ncoarse >= 0) return AMGH_ESTATE;
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
  if (h->finalized || h->ncoarse < 0) return AMGH_ESTATE;
  if (h->levels.empty() && !h->has_finalA) return AMGH_ESTATE;
  HIP_TRY(hipSetDevice(h->device));
  int64_t ws = 0;
  for (Level* L : h->levels) {
    RC_TRY(dev_alloc(&L->res, L->n * h->nrhs));
    RC_TRY(dev_alloc(&L->cx, L->nc * h->nrhs));
    RC_TRY(dev_alloc(&L->cb, L->nc * h->nrhs));
    ws += 8 * (L->n + 2 * L->nc) * h->nrhs;
    if (L->pre.kind == AMGH_SMOOTH_JACOBI || L->post.kind == AMGH_SMOOTH_JACOBI) {
      RC_TRY(dev_alloc(&L->tmp, L->n * h->nrhs));
      ws += 8 * L->n * h->nrhs;
      RC_TRY(csr_ensure_diag(L->smat(), h->stream));
    }
  }
  const int64_t n = fine_n(h);
  if (h->levels.empty()) { RC_TRY(dev_alloc(&h->res_final, n * h->nrhs)); ws += 8 * n * h->nrhs; }
  RC_TRY(dev_alloc(&h->partial, kRedBlocks));
  RC_TRY(dev_alloc(&h->scal, 8));
  RC_TRY(dev_alloc(&h->x0, n * h->nrhs));
  RC_TRY(dev_alloc(&h->b0, n * h->nrhs));
  ws += 16 * n * h->nrhs;
  h->ws_bytes = ws;
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->finalized = true;
  return AMGH_OK;
}

int amgh_num_levels(const amgh_t* h) { return h ? (int)h->levels.size() : 0; }
int64_t amgh_level_size(const amgh_t* h, int l) {
  if (!h || l < 0 || l > (int)h->levels.size()) return -1;
  return l == (int)h->levels.size() ? h->ncoarse : h->levels[l]->n;
}
int64_t amgh_device_bytes(const amgh_t* h) {
  if (!h) return 0;
  int64_t b = h->ws_bytes + ((h->ncoarse > 0 && h->coarse_op) ? h->ncoarse * h->ncoarse * 8 : 0) + h->finalA.bytes;
  for (Level* L : h->levels) b += L->A.bytes + L->S.bytes + L->P.bytes + L->R.bytes;
  return b;
}
int amgh_gs_num_dependency_levels(const amgh_t* h, int l) {
  if (!h || l < 0 || l >= (int)h->levels.size()) return -1;
  amgh_csr* M = h->levels[l]->smat();
  return M->gs ? M->gs->nlev : 0;
}

static int ensure_pcg_bufs(amgh_t* h) {
  if (h->pc_r) return AMGH_OK;
  const int64_t n = fine_n(h);
  RC_TRY(dev_alloc(&h->pc_r, n));
  RC_TRY(dev_alloc(&h->pc_c, n));
  RC_TRY(dev_alloc(&h->pc_u, n));
  h->ws_bytes += 24 * n;
  return AMGH_OK;
}

int amgh_solve_d(amgh_t* h, const double* b_d, double* x_d, int cycle_, int maxiter, double abstol, double reltol,
                 int calculate_residual, double* resid_hist, int* iters) {
  RC_TRY(check_ready(h));
  if (!b_d || !x_d || cycle_ < 0 || cycle_ > 2 || maxiter < 0) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  return solve_dev(h, b_d, x_d, cycle_, maxiter, abstol, reltol, calculate_residual, resid_hist, iters);
}

int amgh_solve(amgh_t* h, const double* b, double* x, int cycle_, int maxiter, double abstol, double reltol,
               int calculate_residual, double* resid_hist, int* iters) {
  RC_TRY(check_ready(h));
  if (!b || !x || cycle_ < 0 || cycle_ > 2 || maxiter < 0) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  const int64_t n = fine_n(h) * h->nrhs;
  HIP_TRY(hipMemcpyAsync(h->b0, b, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->x0, x, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  RC_TRY(solve_dev(h, h->b0, h->x0, cycle_, maxiter, abstol, reltol, calculate_residual, resid_hist, iters));
  HIP_TRY(hipMemcpy(x, h->x0, sizeof(double) * n, hipMemcpyDeviceToHost));
  return AMGH_OK;
}

int amgh_precond_apply_d(amgh_t* h, const double* r_d, double* z_d, int cycle_) {
  RC_TRY(check_ready(h));
  if (!r_d || !z_d || cycle_ < 0 || cycle_ > 2) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(vec_fill(h, z_d, fine_n(h) * h->nrhs, 0.0));
  return apply_cycle(h, z_d, r_d, cycle_);
}

int amgh_precond_apply(amgh_t* h, const double* r, double* z, int cycle_) {
  RC_TRY(check_ready(h));
  if (!r || !z) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  const int64_t n = fine_n(h) * h->nrhs;
  HIP_TRY(hipMemcpyAsync(h->b0, r, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  RC_TRY(amgh_precond_apply_d(h, h->b0, h->x0, cycle_));
  HIP_TRY(hipStreamSynchronize(h->stream));
  RC_TRY(prof_flush(h));
  HIP_TRY(hipMemcpy(z, h->x0, sizeof(double) * n, hipMemcpyDeviceToHost));
  return AMGH_OK;
}

int amgh_cycle_d(amgh_t* h, int level, double* x_d, const double* b_d, int cycle_) {
  RC_TRY(check_ready(h));
  if (!x_d || !b_d || cycle_ < 0 || cycle_ > 2 || level < 0 || level > (int)h->levels.size()) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  if (level == (int)h->levels.size()) return coarse_solve(h, x_d, b_d);
  return cycle(h, level, x_d, b_d, cycle_);
}

int amgh_set_stream(amgh_t* h, void* stream) {
  if (!h) return AMGH_EINVAL;
  h->ext_stream = true;
  h->own_stream = h->own_stream ? h->own_stream : h->stream;
  h->stream = (hipStream_t)stream;  // NULL = the default (null) stream, e.g. torch's current stream
  return AMGH_OK;
}

int amgh_pcg_d(amgh_t* h, const double* b_d, double* x_d, int cycle_, int use_precond, int maxiter, double abstol,
               double reltol, double* resid_hist, int* iters) {
  RC_TRY(check_ready(h));
  if (!b_d || !x_d || cycle_ < 0 || cycle_ > 2 || maxiter < 0) return AMGH_EINVAL;
  if (h->nrhs != 1) return AMGH_EUNSUPPORTED;  // IterativeSolvers' cg takes vectors
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(ensure_pcg_bufs(h));
  return pcg_dev(h, b_d, x_d, cycle_, use_precond, maxiter, abstol, reltol, resid_hist, iters);
}

int amgh_pcg(amgh_t* h, const double* b, double* x, int cycle_, int use_precond, int maxiter, double abstol,
             double reltol, double* resid_hist, int* iters) {
  RC_TRY(check_ready(h));
  if (!b || !x) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  const int64_t n = fine_n(h);
  HIP_TRY(hipMemcpyAsync(h->b0, b, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  RC_TRY(amgh_pcg_d(h, h->b0, h->x0, cycle_, use_precond, maxiter, abstol, reltol, resid_hist, iters));
  HIP_TRY(hipMemcpy(x, h->x0, sizeof(double) * n, hipMemcpyDeviceToHost));
  return AMGH_OK;
}

static amgh_csr* level_op(amgh_t* h, int level, int which) {
  const int L = (int)h->levels.size();
  if (level == L && which == AMGH_OP_A && h->has_finalA) return &h->finalA;
  if (level < 0 || level >= L) return nullptr;
  switch (which) {
    case AMGH_OP_A: return &h->levels[level]->A;
    case AMGH_OP_P: return &h->levels[level]->P;
    case AMGH_OP_R: return &h->levels[level]->R;
  }
  return nullptr;
}

int amgh_level_spmv_d(amgh_t* h, int level, int which, const double* x_d, double* y_d) {
  RC_TRY(check_ready(h));
  amgh_csr* op = level_op(h, level, which);
  if (!op || !x_d || !y_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(csr_apply(op, M_SPMV, x_d, nullptr, y_d, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return AMGH_OK;
}

int amgh_level_spmv(amgh_t* h, int level, int which, const double* x, double* y) {
  RC_TRY(check_ready(h));
  amgh_csr* op = level_op(h, level, which);
  if (!op || !x || !y) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  double *xd = nullptr, *yd = nullptr;
  RC_TRY(dev_upload(&xd, x, op->ncols));
  int rc = dev_alloc(&yd, op->nrows);
  if (rc == AMGH_OK) rc = csr_apply(op, M_SPMV, xd, nullptr, yd, h->stream);
  if (rc == AMGH_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = -1000 - (int)hipGetLastError();
  if (rc == AMGH_OK && hipMemcpy(y, yd, sizeof(double) * op->nrows, hipMemcpyDeviceToHost) != hipSuccess) rc = -1001;
  hipFree(xd); hipFree(yd);
  return rc;
}

int amgh_level_residual_d(amgh_t* h, int level, const double* x_d, const double* b_d, double* r_d) {
  RC_TRY(check_ready(h));
  amgh_csr* op = level_op(h, level, AMGH_OP_A);
  if (!op || !x_d || !b_d || !r_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(csr_apply(op, M_RESID, x_d, b_d, r_d, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return AMGH_OK;
}

static int level_smooth_enqueue(amgh_t* h, int level, int post, double* x_d, const double* b_d) {
  Level* L = h->levels[level];
  const amgh_smoother_t& s = post ? L->post : L->pre;
  double* xc = x_d;
  double* xo = L->tmp;
  if (s.kind == AMGH_SMOOTH_JACOBI && !xo) return AMGH_ESTATE;
  RC_TRY(smooth(h, L, s, xc, xo, b_d));
  if (xc != x_d) RC_TRY(vec_copy(h, x_d, xc, L->n));
  return AMGH_OK;
}

int amgh_level_smooth_d(amgh_t* h, int level, int post, double* x_d, const double* b_d) {
  RC_TRY(check_ready(h));
  if (level < 0 || level >= (int)h->levels.size() || !x_d || !b_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  RC_TRY(level_smooth_enqueue(h, level, post, x_d, b_d));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return AMGH_OK;
}

int amgh_level_smooth(amgh_t* h, int level, int post, double* x, const double* b) {
  RC_TRY(check_ready(h));
  if (level < 0 || level >= (int)h->levels.size() || !x || !b) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(h->device));
  const int64_t n = h->levels[level]->n;
  double *xd = nullptr, *bd = nullptr;
  RC_TRY(dev_upload(&xd, x, n));
  int rc = dev_upload(&bd, b, n);
  if (rc == AMGH_OK) rc = level_smooth_enqueue(h, level, post, xd, bd);
  if (rc == AMGH_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = -1001;
  if (rc == AMGH_OK && hipMemcpy(x, xd, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = -1001;
  hipFree(xd); hipFree(bd);
  return rc;
}

// ---- stand-alone operators ------------------------------------------------
int amgh_csr_create(amgh_csr_t** opp, int device, int64_t nrows, int64_t ncols, const int32_t* rowptr,
                    const int32_t* col, const double* val) {
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
int amgh_csr_spmv_d(amgh_csr_t* op, const double* x_d, double* y_d, void* stream) {
  if (!op || !x_d || !y_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_apply(op, M_SPMV, x_d, nullptr, y_d, (hipStream_t)stream);
}
int amgh_csr_residual_d(amgh_csr_t* op, const double* x_d, const double* b_d, double* r_d, void* stream) {
  if (!op || !x_d || !b_d || !r_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_apply(op, M_RESID, x_d, b_d, r_d, (hipStream_t)stream);
}
int amgh_csr_spmv_add_d(amgh_csr_t* op, const double* x_d, double* y_d, void* stream) {
  if (!op || !x_d || !y_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_apply(op, M_ADD, x_d, nullptr, y_d, (hipStream_t)stream);
}
int amgh_csr_jacobi_d(amgh_csr_t* op, double omega, const double* xin_d, const double* b_d, double* xout_d,
                      void* stream) {
  if (!op || !xin_d || !b_d || !xout_d || xin_d == xout_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_jacobi(op, omega, xin_d, b_d, xout_d, (hipStream_t)stream);
}
int amgh_csr_gs_d(amgh_csr_t* op, int backward, double omega, int is_sor, double* x_d, const double* b_d,
                  void* stream) {
  if (!op || !x_d || !b_d) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(op->device));
  return csr_gs_sweep(op, backward != 0, is_sor != 0, omega, x_d, b_d, (hipStream_t)stream);
}

int amgh_gather_d(int device, int64_t n, const int32_t* idx_d, const double* src_d, double* dst_d, void* stream) {
  if (n < 0 || (n > 0 && (!idx_d || !src_d || !dst_d)) || n >= INT32_MAX) return AMGH_EINVAL;
  if (n == 0) return AMGH_OK;
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(gather_perm_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src_d, idx_d, dst_d,
                     (int)n, (int64_t)0, (int64_t)0);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}

int amgh_dot_d(int device, int64_t n, const double* x_d, const double* y_d, double* scratch_d, double* out,
               void* stream) {
  if (n < 0 || !scratch_d || !out || (n > 0 && (!x_d || !y_d))) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kRedBlocks, (n + kThreads - 1) / kThreads));
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(kThreads), 0, st, x_d, y_d, n, scratch_d + 1);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, st, scratch_d + 1, nb, scratch_d, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, scratch_d, sizeof(double), hipMemcpyDeviceToHost, st));
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
  return AMGH_OK;
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
  amgh_csr* op = level_op(h, level, which <= AMGH_OP_R ? which : AMGH_OP_A);
  if (!op) return AMGH_EINVAL;
  double *x = nullptr, *y = nullptr, *b = nullptr;
  const int64_t nx = std::max(op->ncols, op->nrows);
  RC_TRY(dev_alloc(&x, nx));
  RC_TRY(dev_alloc(&y, nx));
  RC_TRY(dev_alloc(&b, nx));
  // deterministic non-trivial contents (a zero fill would flatter DVFS)
  {
    std::vector<double> hx(nx);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int64_t i = 0; i < nx; ++i) {
      s += 0x9E3779B97F4A7C15ull;
      uint64_t z = s;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      hx[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
    hipMemcpy(x, hx.data(), sizeof(double) * nx, hipMemcpyHostToDevice);
    hipMemcpy(b, hx.data(), sizeof(double) * nx, hipMemcpyHostToDevice);
    hipMemcpy(y, hx.data(), sizeof(double) * nx, hipMemcpyHostToDevice);
  }
  int rc = AMGH_OK;
  auto run = [&]() -> int {
    if (which <= AMGH_OP_R) return csr_apply(op, M_SPMV, x, nullptr, y, h->stream);
    if (which == 3) return csr_apply(op, M_RESID, x, b, y, h->stream);
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

int amgh_debug_set_tunable(const char* name, int value) {
  if (!name) return AMGH_EINVAL;
  if (!strcmp(name, "gs_block_target")) g_gs_block_target = value;
  else if (!strcmp(name, "gs_min_rows")) g_gs_min_rows = value;
  else if (!strcmp(name, "gs_nnz_per_wg")) g_gs_nnz_per_wg = value;
  else if (!strcmp(name, "gs_threads")) g_gs_threads = value;
  else if (!strcmp(name, "gs_block_inverse")) g_gs_block_inverse = value;
  else if (!strcmp(name, "gs_slots")) g_gs_slots = value;
  else if (!strcmp(name, "gs_xcd_map")) g_gs_xcd_map = value;
  else if (!strcmp(name, "gs_super")) g_gs_super = value;
  else if (!strcmp(name, "gs_block_pipe")) g_gs_block_pipe = value;
  else return AMGH_EINVAL;
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
  return AMGH_OK;
}

}  // extern "C"
