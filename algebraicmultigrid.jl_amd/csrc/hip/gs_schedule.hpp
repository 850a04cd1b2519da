// gs_schedule.hpp — host-side construction of the Gauss-Seidel / SOR execution schedule of one operator:
// dependency levels on the symmetrised pattern, the level-permuted matrix copy, chain segments, slot layout,
// and the block-inverse data (in-block inverted triangles, old-x part, earlier-superblock part, near lists).
// Included by amghip.hip after amghip_internal.hpp.
#pragma once

namespace {

// Block-inverse data of one sweep direction: the outer matrix (operator minus the in-block triangle
// and diagonal) and the dense inverses of the in-block triangles.
int blockgs_build_dir(GsSchedule::Outer* o, bool backward, int super, int64_t n, const int32_t* rowptr, const int32_t* col,
                      const real* val, const std::vector<real>& diag, double* max_cond) {
  const int B = kBlk;
  const int nblk = (int)((n + B - 1) / B);
  // a few blocks only: an extra launch costs more than it saves, all couplings stay in the one sequential kernel
  const bool single = nblk <= kBlkSingle;
  std::vector<int32_t> orow(n + 1, 0), ocol, xrow(n + 1, 0), xcol, prow(n + 1, 0), pcol;
  std::vector<real> oval, xval, pval;
  ocol.reserve(rowptr[n]); oval.reserve(rowptr[n]);
  xcol.reserve(rowptr[n]); xval.reserve(rowptr[n]);
  std::vector<real> tinv((size_t)nblk * B * B, 0.0), T((size_t)B * B);
  // near list: outer entries of a block that reference the block swept just before it (blk - 1 forward, blk + 1
  // backward), as {position in the block's outer range, column - first row of that block} + value
  std::vector<int32_t> near_ptr(nblk + 1, 0);
  std::vector<i2_t> near_pi;
  std::vector<real> near_val;
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
        } else if (!skip && !single && (c >= n || (c >= i0 && c < i1) || (backward ? c < i0 : c >= i1))) {
          xcol.push_back(c); xval.push_back(val[j]);  // old x: halo column (frozen), in-block other triangle, block swept later
        } else if (!skip && !single && (c < s0 || c >= s1)) {
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
    real* X = tinv.data() + (size_t)blk * B * B;
    for (int c = 0; c < B; ++c) {
      if (!backward) {
        for (int i = c; i < B; ++i) {
          real s = (i == c) ? 1.0 : 0.0;
          for (int j = c; j < i; ++j) s -= T[(size_t)i * B + j] * X[(size_t)j * B + c];
          X[(size_t)i * B + c] = s / T[(size_t)i * B + i];
        }
      } else {
        for (int i = c; i >= 0; --i) {
          real s = (i == c) ? 1.0 : 0.0;
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

// Footprint policy (tunable "gs_lean" or AMGH_LEAN in the environment; read when a schedule is built):
//   trim (default; AMGH_LEAN=2 / unset)  keep what the default cycle touches: no un-merged slot copy and no CSR copy of
//        slotted composite rows where both directions run merged groups (the CSR copy serves the SELL-like build first),
//        no natural-order P / R / coarse-level A where the cycle runs level-ordered (the stand-alone hooks then go
//        through the level-ordered copies), the backward pre-pass triangle built on first use;
//   lean (AMGH_LEAN=1)  trim + the composite rows compacted the moment they are laid out (lowest peak during the build)
//        and no SELL-like copies;
//   full (AMGH_LEAN=0)  every copy kept (run-time tunables can then switch between all execution paths at full speed).
// Same kernels on the same numbers in all three: the results are bitwise the same.
int gs_footprint() {
  if (g_gs_lean >= 0) return g_gs_lean == 0 ? 0 : g_gs_lean == 1 ? 2 : 1;
  const char* e = getenv("AMGH_LEAN");
  if (!e || !e[0]) return 1;
  return e[0] == '0' ? 0 : e[0] == '1' ? 2 : 1;
}
bool gs_lean() { return gs_footprint() == 2; }
bool gs_trim() { return gs_footprint() >= 1; }

// host threads for the schedule builds: the container's CPU quota, AMGH_BUILD_THREADS overrides
int merge_threads() {
  unsigned hw = std::thread::hardware_concurrency();
  long quota = -1, period = -1;
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // do not oversubscribe a container's CPU quota
    char q[64] = {0};
    if (std::fscanf(f, "%63s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atol(q);
    std::fclose(f);
  }
  int t = (int)std::max(1u, std::min(hw ? hw : 8u, 32u));
  if (quota > 0 && period > 0) t = (int)std::min<long>(t, std::max<long>(1, (quota + period - 1) / period));
  if (const char* e = getenv("AMGH_BUILD_THREADS")) t = std::max(1, atoi(e));
  return t;
}

// fn(t) for t in [0, T): on T threads when they can be had, inline otherwise (a thread that cannot be created
// must not take the process down)
template <class F>
void run_threads(int T, F fn) {
  std::vector<std::thread> th;
  int started = 0;
  for (; started < T - 1; ++started) {
    try { th.emplace_back(fn, started); } catch (const std::system_error&) { break; }
  }
  for (int t = started; t < T; ++t) fn(t);
  for (auto& x : th) x.join();
}

// A triangular system in dependency-level order on the host: row p updates x[p] from
//   diag[p] * x[p] = rhs[p] - sum_{entries != dpos[p]} val * x[col]
// and rows of one level [lvl_ptr[l], lvl_ptr[l+1]) do not reference each other.
struct HostLevelCsr {
  int64_t n = 0;
  int nlev = 0;
  std::vector<int32_t> lvl_ptr, prow, pcol, pdpos;
  std::vector<real> pval, pdiag;
};

// ---- schedule construction on the device ------------------------------------------------------------------------------
// rows of a level-ordered CSR copied into their slots (positions computed on the host: wmeta), padding stays zero
__global__ void slot_fill_kernel(const int32_t* prow, const int32_t* pcol, const real* pval, const i4_t* wmeta, int n,
                                 int32_t* wcol, real* wval) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const i4_t m = wmeta[p];
  const int len = m.y - m.x;
  if (len <= 0) return;  // row not in a slotted segment
  const int32_t src = prow[p];
  for (int e = 0; e < len; ++e) {
    wcol[m.x + e] = pcol[src + e];
    wval[m.x + e] = pval[src + e];
  }
}
// dependency level of every level-ordered row
__global__ void lev_of_kernel(const int32_t* lvl_ptr, int nlev, int n, int32_t* lev_of) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int lo = 0, hi = nlev;  // largest l with lvl_ptr[l] <= p
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (lvl_ptr[mid] <= p) lo = mid; else hi = mid;
  }
  lev_of[p] = lo;
}
// the triangle (plus halo columns) a sweep direction does not substitute over, counted (tp == nullptr) or written
__global__ void tri_kernel(const int32_t* prow, const int32_t* pcol, const real* pval, const real* pdiag,
                           const int32_t* lev_of, int n, int backward, real diag_shift, const int32_t* tp, int32_t* tc,
                           real* tv, int32_t* cnt) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int lp = lev_of[p];
  int32_t c = 0, o = tp ? tp[p] : 0;
  for (int32_t j = prow[p]; j < prow[p + 1]; ++j) {
    const int32_t col = pcol[j];
    if (col == p) continue;
    const bool other = col >= n || (backward ? lev_of[col] < lp : lev_of[col] > lp);
    if (other) {
      if (tp) { tc[o] = col; tv[o] = pval[j]; ++o; }
      ++c;
    }
  }
  if (diag_shift != 0.0 && pdiag[p] != 0.0) {
    if (tp) { tc[o] = p; tv[o] = -diag_shift * pdiag[p]; }
    ++c;
  }
  if (!tp) cnt[p] = c;
}

// s = b - T x pre-pass matrix of one direction, built on the device from the schedule's own level-ordered copy
int tri_build_dev(GsSchedule::Tri* t, const GsSchedule* g, bool backward, int64_t* bytes, real diag_shift,
                  int64_t* nnz_out) {
  const int64_t n = g->n;
  if (n <= 0) return AMGH_OK;
  int32_t *lev_of = nullptr, *cnt = nullptr;
  RC_TRY(dev_alloc(&lev_of, n));
  int rc = dev_alloc(&cnt, n + 1);
  const unsigned grid = (unsigned)((n + 255) / 256);
  int64_t total = 0;
  if (rc == AMGH_OK) {
    hipLaunchKernelGGL(lev_of_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)g->d_lvl_ptr, g->nlev, (int)n, lev_of);
    hipLaunchKernelGGL(tri_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)g->rowptr, (const int32_t*)g->col,
                       (const real*)g->val, (const real*)g->diag, (const int32_t*)lev_of, (int)n, backward ? 1 : 0,
                       diag_shift, (const int32_t*)nullptr, (int32_t*)nullptr, (real*)nullptr, cnt);
    rc = dev_alloc(&t->rowptr, n + 1);
  }
  if (rc == AMGH_OK) rc = dev_exclusive_scan(cnt, t->rowptr, n, &total, nullptr);
  if (rc == AMGH_OK) rc = dev_alloc(&t->col, total);
  if (rc == AMGH_OK) rc = dev_alloc(&t->val, total);
  if (rc == AMGH_OK) {
    hipLaunchKernelGGL(tri_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)g->rowptr, (const int32_t*)g->col,
                       (const real*)g->val, (const real*)g->diag, (const int32_t*)lev_of, (int)n, backward ? 1 : 0,
                       diag_shift, (const int32_t*)t->rowptr, t->col, t->val, (int32_t*)nullptr);
    if (hipDeviceSynchronize() != hipSuccess) rc = -1001;
  }
  hipFree(lev_of); hipFree(cnt);
  if (rc != AMGH_OK) return rc;
  *bytes += (n + 1) * 4 + total * kEntB;
  if (nnz_out) *nnz_out = total;
  return AMGH_OK;
}

// Upload one level-ordered system and derive its execution layout: row / level descriptors for the chain
// kernel, segments (runs of narrow levels chained in one workgroup, one launch per wide level), slot arrays.
// `orig` = original row id of each level-ordered row (rowmeta.w), may be null.
// compact (memory-lean mode, merged children only): the CSR copy keeps just the rows no slot launch covers (chained
// narrow groups, rows too long for a slot); every other row lives in the slot arrays only.
// dev_src (optional): the rows' entries are already on the device as a contiguous CSR (h.pcol / h.pval are empty; only
// h.prow, the row lengths, is used on the host); ownership of the three arrays passes to this function.
struct DevCsr { int32_t* rowptr = nullptr; int32_t* col = nullptr; real* val = nullptr; };
__global__ void compact_rows_kernel(const int32_t* prow, const int32_t* pcol, const real* pval, const int32_t* cprow, int n,
                                    int32_t* ccol, real* cval) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int len = cprow[p + 1] - cprow[p];
  const int32_t src = prow[p], dst = cprow[p];
  for (int e = 0; e < len; ++e) { ccol[dst + e] = pcol[src + e]; cval[dst + e] = pval[src + e]; }
}
int layout_upload(GsSchedule* g, const HostLevelCsr& h, const int32_t* orig, int slot_entries = kSlot,
                  bool compact = false, DevCsr* dev_src = nullptr, real* shared_diag = nullptr) {
  const int64_t n = h.n;
  const std::vector<int32_t>& prow = h.prow;
  const std::vector<int32_t>& pcol = h.pcol;
  const std::vector<int32_t>& pdpos = h.pdpos;
  const std::vector<real>& pval = h.pval;
  const int64_t nnz = prow[n];
  g->n = n;
  g->nnz = nnz;
  g->nlev = h.nlev;
  g->lvl_ptr = h.lvl_ptr;
  g->slot_entries = slot_entries;
  bool full_csr_on_device = false;  // the slot fill already put the whole CSR copy on the device
  const int SE = slot_entries;                                    // entries per slot
  const int max_rows = SE == kSlot ? kSlot : kBigRows;           // rows per slot
  if (shared_diag) { g->diag = shared_diag; g->diag_shared = true; }   // a merged child sweeps the parent's rows: one copy of their diagonal
  else RC_TRY(dev_upload(&g->diag, h.pdiag.data(), n));
  RC_TRY(dev_upload(&g->d_lvl_ptr, g->lvl_ptr.data(), g->nlev + 1));
  g->bytes += (shared_diag ? 0 : n * kRealB) + (g->nlev + 1) * 4;
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
    std::vector<int32_t> slot_row;
    std::vector<i4_t> wmeta(n, i4_t{0, 0, -1, 0});
    for (auto& sg : g->segs) {
      if (sg.chain) continue;
      const int ra = g->lvl_ptr[sg.l0], rb = g->lvl_ptr[sg.l0 + 1];
      bool fits = true;
      for (int p2 = ra; p2 < rb && fits; ++p2) fits = prow[p2 + 1] - prow[p2] <= SE;
      // slot positions are int32: a level that would push the slot arrays past 2^31 entries keeps the CSR kernels
      if (fits && ((int64_t)slot_row.size() + (prow[rb] - prow[ra]) / std::max(1, SE / 2) + (rb - ra) / max_rows + 2) * SE >= (int64_t)INT32_MAX) fits = false;
      if (!fits) continue;
      sg.slot0 = (int)slot_row.size();
      int fill = 0, rows_in = 0;
      bool open = false;  // the level's first row always opens a slot (composite rows can be empty: fill + 0 > SE never fires)
      for (int p2 = ra; p2 < rb; ++p2) {  // pass 1: positions only (the entries are copied in parallel below)
        const int len = prow[p2 + 1] - prow[p2];
        if (!open || fill + len > SE || rows_in >= max_rows) {  // open a new slot (the previous one stays zero padded)
          open = true;
          slot_row.push_back(p2);
          fill = 0;
          rows_in = 0;
        }
        const int32_t start = (int32_t)((slot_row.size() - 1) * (size_t)SE + fill);
        wmeta[p2] = i4_t{start, start + len, pdpos[p2] >= 0 ? start + (pdpos[p2] - prow[p2]) : -1, 0};
        fill += len;
        ++rows_in;
      }
      sg.nslots = (int)slot_row.size() - sg.slot0;
    }
    if (!slot_row.empty()) {
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
      if (getenv("AMGH_VERBOSE")) {  // self-check of the slot layout
        int64_t bad = 0;
        for (auto& sg : g->segs) {
          if (sg.chain || sg.nslots <= 0) continue;
          for (int q = 0; q < sg.nslots; ++q) {
            const int sidx = sg.slot0 + q;
            const int ra = sr2[2 * sidx], rb = sr2[2 * sidx + 1];
            if (rb - ra > max_rows || rb <= ra) ++bad;
            for (int r = ra; r < rb; ++r)
              if (wmeta[r].x < (int64_t)sidx * SE || wmeta[r].y > (int64_t)(sidx + 1) * SE || wmeta[r].y < wmeta[r].x) ++bad;
          }
        }
        for (auto& sg : g->segs) {  // every row of a slotted level belongs to exactly one of its slots
          if (sg.chain || sg.nslots <= 0) continue;
          if (sr2[2 * sg.slot0] != g->lvl_ptr[sg.l0]) ++bad;
          for (int q = 0; q + 1 < sg.nslots; ++q)
            if (sr2[2 * (sg.slot0 + q) + 1] != sr2[2 * (sg.slot0 + q + 1)]) ++bad;
        }
        if (bad) fprintf(stderr, "[amghip] slot layout self-check: %lld inconsistencies (SE=%d)\n", (long long)bad, SE);
      }
      // the entries go into their slots ON THE DEVICE, from the level-ordered CSR (one upload of the matrix instead of
      // two, no host-side slot arrays): the host only decided the positions
      const int64_t wtotal = (int64_t)slot_row.size() * SE;
      RC_TRY(dev_alloc(&g->wcol, wtotal));
      RC_TRY(dev_alloc(&g->wval, wtotal));
      HIP_TRY(dev_zero(g->wcol, sizeof(int32_t) * (size_t)wtotal, nullptr));
      HIP_TRY(dev_zero(g->wval, sizeof(real) * (size_t)wtotal, nullptr));
      RC_TRY(dev_upload(&g->slot_row, sr2.data(), (int64_t)sr2.size()));
      RC_TRY(dev_upload(&g->wmeta, wmeta.data(), n));
      int32_t *t_row = nullptr, *t_col = nullptr;
      real* t_val = nullptr;
      int rcf = AMGH_OK;
      if (dev_src) {
        t_row = dev_src->rowptr; t_col = dev_src->col; t_val = dev_src->val;
      } else {
        rcf = dev_upload(&t_row, prow.data(), n + 1);
        if (rcf == AMGH_OK) rcf = dev_upload(&t_col, pcol.data(), nnz);
        if (rcf == AMGH_OK) rcf = dev_upload(&t_val, pval.data(), nnz);
      }
      if (rcf == AMGH_OK) {
        hipLaunchKernelGGL(slot_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const int32_t*)t_row,
                           (const int32_t*)t_col, (const real*)t_val, (const i4_t*)g->wmeta, (int)n, g->wcol, g->wval);
        if (hipDeviceSynchronize() != hipSuccess) rcf = -1001;
      }
      if (rcf == AMGH_OK && !compact) {  // the same arrays are this system's CSR copy
        g->rowptr = t_row; g->col = t_col; g->val = t_val;
        t_row = t_col = nullptr; t_val = nullptr;
        full_csr_on_device = true;
        if (dev_src) *dev_src = DevCsr();
      }
      if (!dev_src) { hipFree(t_row); hipFree(t_col); hipFree(t_val); }
      RC_TRY(rcf);
      g->bytes += wtotal * kEntB + (int64_t)sr2.size() * 4 + n * 16;
      g->slot_total = wtotal;
      g->slot_bytes = wtotal * kEntB + (int64_t)sr2.size() * 4 + n * 16;
    }
  }
  // the CSR copy (chain kernel, stream-kernel fallback): all rows, or — compact — only those no slot launch covers
  {
    std::vector<int32_t> cprow, ccol, cdpos;
    std::vector<real> cval;
    const std::vector<int32_t>* urow = &prow;
    const std::vector<int32_t>* ucol = &pcol;
    const std::vector<int32_t>* udpos = &pdpos;
    const std::vector<real>* uval = &pval;
    if (compact) {
      std::vector<char> keep(n, 0);
      for (const auto& sg : g->segs)
        if (sg.chain || sg.nslots <= 0)
          for (int32_t p2 = g->lvl_ptr[sg.l0]; p2 < g->lvl_ptr[sg.l1]; ++p2) keep[p2] = 1;
      cprow.assign(n + 1, 0);
      for (int64_t p2 = 0; p2 < n; ++p2) cprow[p2 + 1] = cprow[p2] + (keep[p2] ? prow[p2 + 1] - prow[p2] : 0);
      cdpos.assign(n, -1);
      if (!dev_src) {
        ccol.resize(cprow[n]);
        cval.resize(cprow[n]);
      }
      for (int64_t p2 = 0; p2 < n; ++p2) {
        if (!keep[p2]) continue;
        if (!dev_src) {
          std::copy(pcol.begin() + prow[p2], pcol.begin() + prow[p2 + 1], ccol.begin() + cprow[p2]);
          std::copy(pval.begin() + prow[p2], pval.begin() + prow[p2 + 1], cval.begin() + cprow[p2]);
        }
        if (pdpos[p2] >= 0) cdpos[p2] = pdpos[p2] - prow[p2] + cprow[p2];
      }
      urow = &cprow; ucol = &ccol; udpos = &cdpos; uval = &cval;
      g->compacted = true;
    }
    const int64_t unnz = (*urow)[n];
    if (!full_csr_on_device && dev_src) {
      // entries already on the device: all rows (no slotted level at all) or the kept rows gathered by a kernel
      if (!compact) {
        if (!dev_src->rowptr) RC_TRY(dev_upload(&dev_src->rowptr, prow.data(), n + 1));
        g->rowptr = dev_src->rowptr; g->col = dev_src->col; g->val = dev_src->val;
        *dev_src = DevCsr();
      } else {
        RC_TRY(dev_upload(&g->rowptr, urow->data(), n + 1));
        RC_TRY(dev_alloc(&g->col, unnz));
        RC_TRY(dev_alloc(&g->val, unnz));
        if (!dev_src->rowptr) RC_TRY(dev_upload(&dev_src->rowptr, prow.data(), n + 1));
        if (n > 0)
          hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr,
                             (const int32_t*)dev_src->rowptr, (const int32_t*)dev_src->col, (const real*)dev_src->val,
                             (const int32_t*)g->rowptr, (int)n, g->col, g->val);
        HIP_TRY(hipDeviceSynchronize());
      }
    } else if (!full_csr_on_device) {
      RC_TRY(dev_upload(&g->rowptr, urow->data(), n + 1));
      RC_TRY(dev_upload(&g->col, ucol->data(), unnz));
      RC_TRY(dev_upload(&g->val, uval->data(), unnz));
    }
    if (dev_src) { hipFree(dev_src->rowptr); hipFree(dev_src->col); hipFree(dev_src->val); *dev_src = DevCsr(); }
    RC_TRY(dev_upload(&g->dpos, udpos->data(), n));
    std::vector<i4_t> meta(n), desc(g->nlev);
    for (int64_t p2 = 0; p2 < n; ++p2)
      meta[p2] = i4_t{(*urow)[p2], (*urow)[p2 + 1], (*udpos)[p2], orig ? orig[p2] : (int32_t)p2};
    for (int l2 = 0; l2 < g->nlev; ++l2)
      desc[l2] = i4_t{g->lvl_ptr[l2], g->lvl_ptr[l2 + 1], (*urow)[g->lvl_ptr[l2]], (*urow)[g->lvl_ptr[l2 + 1]]};
    RC_TRY(dev_upload(&g->rowmeta, meta.data(), n));
    RC_TRY(dev_upload(&g->desc, desc.data(), g->nlev));
    g->csr_bytes = (n + 1) * 4 + unnz * kEntB + n * 4 + n * 16 + g->nlev * 16;
    g->bytes += g->csr_bytes;
    // small enough to be swept out of LDS as a whole (gs_chain_tiny_kernel)?
    int64_t widest = 0;
    for (int l2 = 0; l2 < g->nlev; ++l2) widest = std::max<int64_t>(widest, prow[g->lvl_ptr[l2 + 1]] - prow[g->lvl_ptr[l2]]);
    g->tiny_ok = !compact && n <= kTinyRows && nnz <= kTinyNnz && widest <= kTinyLvlNnz && g->nlev <= kTinyRows;
    // ... and as one record for the single-wave walk (gs_wave_kernel)?  Positions are level-order rows: a column beyond
    // the square block (halo of a sharded operator) rules it out.
    if (g->tiny_ok && n > 0 && n < 65535) {
      int maxlen = 0;
      bool square = true;
      for (int64_t p2 = 0; p2 < n; ++p2) {
        int off = 0;
        for (int32_t j = prow[p2]; j < prow[p2 + 1]; ++j) {
          if (h.pcol[j] >= n) square = false;
          if (j != h.pdpos[p2]) ++off;
        }
        maxlen = std::max(maxlen, off);
      }
      const int maxk = ((std::max(1, maxlen) + 5) / 6) * 6;
      std::vector<uint16_t> stp;
      for (int l2 = 0; l2 < g->nlev; ++l2)
        for (int32_t q = g->lvl_ptr[l2]; q < g->lvl_ptr[l2 + 1]; q += 64) stp.push_back((uint16_t)q);
      const int steps = (int)stp.size();
      stp.push_back((uint16_t)n);
      const size_t rs = (size_t)wave_row_bytes(maxk);
      const size_t recb = ((size_t)n * rs + (size_t)(steps + 1) * 2 + 15) & ~(size_t)15;
      const size_t lds = (((size_t)(n + 1) * sizeof(real) + 15) & ~(size_t)15) + (((size_t)n * sizeof(real) + 15) & ~(size_t)15) + recb;
      if (square && maxk <= kWaveMaxK && steps <= kWaveMaxSteps && lds <= 150 * 1024 && (size_t)(n + 1) * sizeof(real) <= 65535) {
        std::vector<unsigned char> rec(recb, 0);
        const int nvc = wave_nvc(maxk);
        const uint16_t zoff = (uint16_t)((size_t)n * sizeof(real));   // the LDS slot that holds 0: padding entries are 0 * 0
        for (int64_t p2 = 0; p2 < n; ++p2) {
          real* v = (real*)(rec.data() + (size_t)p2 * rs);
          uint16_t* cc = (uint16_t*)(rec.data() + (size_t)p2 * rs + (size_t)16 * nvc);
          for (int k = 0; k < 8 * wave_ncc(maxk); ++k) cc[k] = zoff;
          int k = 0;
          for (int32_t j = prow[p2]; j < prow[p2 + 1]; ++j) {
            if (j == h.pdpos[p2]) continue;
            v[k] = h.pval[j];
            cc[k] = (uint16_t)((size_t)h.pcol[j] * sizeof(real));
            ++k;
          }
          const real dg = h.pdiag[p2];
          v[maxk] = dg;
          // reciprocal for the division-free quotient; 0 = "divide" (a diagonal whose reciprocal or products may leave the normal range)
          const double ad = std::fabs((double)dg);
          const bool safe = sizeof(real) == 8 ? (ad > 1e-100 && ad < 1e100) : (ad > 1e-12 && ad < 1e12);
          v[maxk + 1] = safe ? (real)1 / dg : (real)0;
        }
        std::copy(stp.begin(), stp.end(), (uint16_t*)(rec.data() + (size_t)n * rs));
        RC_TRY(dev_upload(&g->ww_rec, rec.data(), (int64_t)recb));
        g->ww_S = 0; g->ww_maxk = maxk; g->ww_steps = steps; g->ww_lds = lds;
        g->bytes += (int64_t)recb;
        g->csr_bytes += (int64_t)recb;
      }
      // ... and with four lanes per row (gs_waveq_kernel): rows long enough for a quarter to be worth a lane, steps of 16 rows
      if (g->ww_rec && g_gs_wave_quad && maxlen >= 8) {
        const int e4 = (maxlen + kWaveQ - 1) / kWaveQ;
        const int E = e4 <= 3 ? 3 : e4 <= 5 ? 5 : e4 <= 7 ? 7 : 9;
        std::vector<uint16_t> sq;
        for (int l2 = 0; l2 < g->nlev; ++l2)
          for (int32_t q = g->lvl_ptr[l2]; q < g->lvl_ptr[l2 + 1]; q += kWave / kWaveQ) sq.push_back((uint16_t)q);
        const int qsteps = (int)sq.size();
        sq.push_back((uint16_t)n);
        const size_t qrs = (size_t)wave_row_bytes(E);
        const size_t qrecb = ((size_t)n * kWaveQ * qrs + (size_t)(qsteps + 1) * 2 + 15) & ~(size_t)15;
        const size_t qlds = (((size_t)(n + 1) * sizeof(real) + 15) & ~(size_t)15) + (((size_t)n * sizeof(real) + 15) & ~(size_t)15) + qrecb;
        if (e4 <= 9 && qsteps <= kWaveMaxSteps && qlds <= 150 * 1024 && (size_t)n * kWaveQ < (1u << 20)) {
          std::vector<unsigned char> qrec(qrecb, 0);
          const int qnvc = wave_nvc(E);
          const uint16_t zoff = (uint16_t)((size_t)n * sizeof(real));
          for (int64_t p2 = 0; p2 < n; ++p2) {
            const real dg = h.pdiag[p2];
            const double ad = std::fabs((double)dg);
            const bool safe = sizeof(real) == 8 ? (ad > 1e-100 && ad < 1e100) : (ad > 1e-12 && ad < 1e12);
            for (int sub = 0; sub < kWaveQ; ++sub) {
              unsigned char* mr = qrec.data() + ((size_t)p2 * kWaveQ + sub) * qrs;
              real* v = (real*)mr;
              uint16_t* cc = (uint16_t*)(mr + (size_t)16 * qnvc);
              for (int k = 0; k < 8 * wave_ncc(E); ++k) cc[k] = zoff;
              v[E] = dg;
              v[E + 1] = safe ? (real)1 / dg : (real)0;
            }
            int k = 0;
            for (int32_t j = prow[p2]; j < prow[p2 + 1]; ++j) {
              if (j == h.pdpos[p2]) continue;
              unsigned char* mr = qrec.data() + ((size_t)p2 * kWaveQ + (k % kWaveQ)) * qrs;
              ((real*)mr)[k / kWaveQ] = h.pval[j];
              ((uint16_t*)(mr + (size_t)16 * qnvc))[k / kWaveQ] = (uint16_t)((size_t)h.pcol[j] * sizeof(real));
              ++k;
            }
          }
          std::copy(sq.begin(), sq.end(), (uint16_t*)(qrec.data() + (size_t)n * kWaveQ * qrs));
          RC_TRY(dev_upload(&g->wq_rec, qrec.data(), (int64_t)qrecb));
          g->wq_E = E; g->wq_steps = qsteps; g->wq_lds = qlds;
          g->bytes += (int64_t)qrecb;
          g->csr_bytes += (int64_t)qrecb;
        }
      }
    }
  }
  return AMGH_OK;
}

// Trimmed footprint: after the SELL-like build has read it, the CSR copy of a merged child keeps only the rows no slot
// (or SELL) launch covers — chained narrow groups, rows too long for a slot; what layout_upload(compact = true) does at
// once, done afterwards.  Composite rows carry no diagonal position (dpos = -1 everywhere).
int child_compact_late(GsSchedule* g, const std::vector<int32_t>& prow, const int32_t* orig) {
  if (g->compacted || !g->rowptr) return AMGH_OK;
  const int64_t n = g->n;
  std::vector<int32_t> cprow(n + 1, 0);
  bool any_kept = false;
  {
    std::vector<char> keep(n, 0);
    for (const auto& sg : g->segs)
      if (sg.chain || sg.nslots <= 0)
        for (int32_t p2 = g->lvl_ptr[sg.l0]; p2 < g->lvl_ptr[sg.l1]; ++p2) keep[p2] = 1;
    for (int64_t p2 = 0; p2 < n; ++p2) {
      cprow[p2 + 1] = cprow[p2] + (keep[p2] ? prow[p2 + 1] - prow[p2] : 0);
      any_kept = any_kept || keep[p2];
    }
  }
  const int64_t unnz = cprow[n];
  int32_t *nrp = nullptr, *ncol = nullptr;
  real* nval = nullptr;
  int rc = dev_upload(&nrp, cprow.data(), n + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&ncol, unnz);
  if (rc == AMGH_OK) rc = dev_alloc(&nval, unnz);
  if (rc == AMGH_OK && n > 0) {
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const int32_t*)g->rowptr,
                       (const int32_t*)g->col, (const real*)g->val, (const int32_t*)nrp, (int)n, ncol, nval);
    if (hipDeviceSynchronize() != hipSuccess) rc = -1001;
  }
  if (rc != AMGH_OK) { hipFree(nrp); hipFree(ncol); hipFree(nval); return rc; }
  hipFree(g->rowptr); hipFree(g->col); hipFree(g->val);
  g->rowptr = nrp; g->col = ncol; g->val = nval;
  g->compacted = true;
  // row / level descriptors of the chain kernel follow the new positions
  hipFree(g->rowmeta); hipFree(g->desc); hipFree(g->dpos);
  g->rowmeta = nullptr; g->desc = nullptr; g->dpos = nullptr;
  int64_t meta_bytes = 0;
  if (any_kept) {
    std::vector<i4_t> meta(n), desc(g->nlev);
    for (int64_t p2 = 0; p2 < n; ++p2) meta[p2] = i4_t{cprow[p2], cprow[p2 + 1], -1, orig ? orig[p2] : (int32_t)p2};
    for (int l2 = 0; l2 < g->nlev; ++l2)
      desc[l2] = i4_t{g->lvl_ptr[l2], g->lvl_ptr[l2 + 1], cprow[g->lvl_ptr[l2]], cprow[g->lvl_ptr[l2 + 1]]};
    std::vector<int32_t> dp(n, -1);
    RC_TRY(dev_upload(&g->rowmeta, meta.data(), n));
    RC_TRY(dev_upload(&g->desc, desc.data(), g->nlev));
    RC_TRY(dev_upload(&g->dpos, dp.data(), n));
    meta_bytes = n * 4 + n * 16 + g->nlev * 16;
  }
  const int64_t now = (n + 1) * 4 + unnz * kEntB + meta_bytes;
  g->bytes += now - g->csr_bytes;
  g->csr_bytes = now;
  return AMGH_OK;
}

// SELL-like copy of the slotted groups of a merged child whose rows are long enough for it to pay (see gs_sell_kernel):
// lanes per row K by the group's mean row length (about 4 entries per lane), iterations per chunk by the longest row
// of the chunk.  Filled on the device from the child's CSR copy (which must hold all rows: not for compacted children).
int sell_build(GsSchedule* g, const std::vector<int32_t>& prow) {
  if (g->compacted || !g->rowptr) return AMGH_OK;
  std::vector<i2_t> chunk;
  int64_t total64 = 0;
  for (auto& sg : g->segs) {
    sg.sell_k = 0;
    if (sg.chain || sg.nslots <= 0) continue;
    const int32_t ra = g->lvl_ptr[sg.l0], rb = g->lvl_ptr[sg.l0 + 1];
    const int rows = rb - ra;
    if (rows <= 0) continue;
    const double mean = (double)(prow[rb] - prow[ra]) / rows;
    // measured on the 256^3 hierarchy (profiles/r02_gs_sell_levels.log): -29 % on the 38 k-row level (rows of 230-330
    // entries, where it replaces the long-row slot kernel), -5 % at ~110 entries per row, nothing at 60, +2 % at 27
    // (composite rows of one chunk differ a lot in length: the chunk runs as long as its longest row)
    if (mean < 96.0) continue;
    int K = 8;
    while (K < 64 && K * 4 < mean) K <<= 1;
    const int C = kWave / K;
    const int nch = (rows + C - 1) / C;
    int64_t seg64 = 0;
    const size_t first = chunk.size();
    bool ok = true;
    for (int c = 0; c < nch && ok; ++c) {
      int32_t mx = 0;
      for (int32_t r = ra + c * C; r < std::min<int32_t>(rb, ra + (c + 1) * C); ++r) mx = std::max(mx, prow[r + 1] - prow[r]);
      const int T = (mx + K - 1) / K;
      if (total64 + seg64 >= (int64_t)UINT32_MAX - 4096) ok = false;
      chunk.push_back(i2_t{(int32_t)(uint32_t)(total64 + seg64), T});
      seg64 += T;
    }
    // padding beyond 1.6 x the group's entries: not worth it (rows of very different lengths side by side)
    if (!ok || seg64 * kWave > 1.6 * (double)(prow[rb] - prow[ra]) + 4096) { chunk.resize(first); continue; }
    sg.sell_k = K; sg.sell_chunk0 = (int)first; sg.sell_nchunks = nch;
    total64 += seg64;
  }
  if (chunk.empty()) return AMGH_OK;
  RC_TRY(dev_upload(&g->schunk, chunk.data(), (int64_t)chunk.size()));
  RC_TRY(dev_alloc(&g->scol, total64 * kWave));
  RC_TRY(dev_alloc(&g->sval, total64 * kWave));
  for (const auto& sg : g->segs) {
    if (sg.sell_k <= 0) continue;
    const int32_t ra = g->lvl_ptr[sg.l0], rb = g->lvl_ptr[sg.l0 + 1];
    const int64_t threads = (int64_t)sg.sell_nchunks * kWave;
    hipLaunchKernelGGL(sell_fill_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr,
                       (const int32_t*)g->rowptr, (const int32_t*)g->col, (const real*)g->val, (const i2_t*)g->schunk,
                       sg.sell_chunk0, sg.sell_nchunks, (int)ra, (int)(rb - ra), sg.sell_k, g->scol, g->sval);
  }
  HIP_TRY(hipDeviceSynchronize());
  g->sell_total = total64 * kWave;
  g->sell_bytes = total64 * kWave * kEntB + (int64_t)chunk.size() * 8;
  g->bytes += g->sell_bytes;
  return AMGH_OK;
}

// ---- merged dependency levels -------------------------------------------------------------------
// A wide dependency level costs one kernel boundary (~3.3 us) however few rows it has.  Substitution removes
// boundaries: inside a GROUP of m consecutive levels, a row that reads x_c of an earlier level of the same group
// gets that reference replaced by row c's own update formula,
//   x_c = (s_c - sum_e val_e * ext[col_e]) / d_c ,
// so that every row of the group reads only rows of EARLIER groups (final) and right-hand-side entries: all rows
// of a group are independent and run in one launch.  To keep the substitution one-sided the other triangle is
// applied first, s = b - T x_old (one full-chip residual launch per sweep), and a row may then reference s of
// other rows: the kernels see one extended vector ext = [x (ncols) ; s (n)], column ncols + c = s_c.
// Same iterate as the row-by-row sweep in exact arithmetic; rounding differs at the 1e-16 level (like the
// block-inverse path, unlike the unmerged schedule, which reproduces the scalar loop bit for bit).
// Fill grows with m and with the row length, launches shrink with m: the group size is chosen per operator and
// direction from a cost model on the measured fill.
struct MergeResult {
  HostLevelCsr sys;      // grouped levels, composite rows (no diagonal entry: dpos = -1)
  int64_t max_row = 0;
  double growth = 0.0;   // largest (sum of |coefficients| / |diagonal|) of a composite row: substitution amplifies like
                         // prod |l_ij / d_j| — harmless for diagonally dominant operators, a warning sign otherwise
};

// grouping of the dependency levels of `base` into groups of m (counted from level 0 forward, from the last level
// backward); groups are numbered by ascending position in both directions
struct MergeGroups {
  int m = 1, ngrp = 0;
  bool backward = false;
  std::vector<int32_t> gptr;  // rows of group q: [gptr[q], gptr[q + 1])
  std::vector<int32_t> lev_of;
  int group_of_level(int l, int nlev) const { return backward ? ngrp - 1 - (nlev - 1 - l) / m : l / m; }
};

MergeGroups merge_groups(const HostLevelCsr& base, int m, bool backward) {
  MergeGroups G;
  G.m = m; G.backward = backward;
  const int nlev = base.nlev;
  G.ngrp = nlev ? (nlev + m - 1) / m : 0;
  G.gptr.assign(G.ngrp + 1, 0);
  for (int l = 0; l < nlev; ++l) {
    const int q = G.group_of_level(l, nlev);
    G.gptr[q + 1] = std::max(G.gptr[q + 1], base.lvl_ptr[l + 1]);
  }
  for (int q = 0; q < G.ngrp; ++q) G.gptr[q + 1] = std::max(G.gptr[q + 1], G.gptr[q]);
  G.lev_of.resize(base.n);
  for (int l = 0; l < nlev; ++l)
    for (int32_t p = base.lvl_ptr[l]; p < base.lvl_ptr[l + 1]; ++p) G.lev_of[p] = l;
  return G;
}

// composite rows of the groups [q0, q1): rows come out in ascending row order; len[p - gptr[q0]] entries each.
// Returns the longest row, or INT32_MAX when a row outgrows a slot (the caller gives up on this m).
struct MergeChunk {
  std::vector<int32_t> len, col;
  std::vector<real> val;
  double growth = 0.0;  // max over rows of sum |composite coefficient| / |diagonal|
};
int64_t merge_chunk(const HostLevelCsr& base, const MergeGroups& G, int64_t ncols, int q0, int q1, MergeChunk* out,
                    int row_cap = kBigSlot) {
  constexpr int kCap = 4 * kBigSlot;  // open-addressing accumulator: rows longer than row_cap are rejected
  std::vector<int32_t> key(kCap, -1);
  std::vector<std::pair<int32_t, int32_t>> order;  // {column, accumulator slot} of the row being built
  std::vector<real> acc(kCap);
  std::vector<int64_t> off;       // per row of the current group: offset into gcol / gval
  std::vector<int32_t> gcol, glen;
  std::vector<real> gval;
  int64_t max_row = 0;
  const int32_t row0 = G.gptr[q0];
  out->len.assign(G.gptr[q1] - row0, 0);
  out->col.clear();
  out->val.clear();
  out->growth = 0.0;
  for (int q = q0; q < q1; ++q) {
    const int32_t ga = G.gptr[q], gb = G.gptr[q + 1];
    const int32_t gn = gb - ga;
    off.assign(gn + 1, 0);
    glen.assign(gn, 0);
    gcol.clear();
    gval.clear();
    std::vector<int64_t> start(gn, 0);
    for (int32_t it = 0; it < gn; ++it) {
      const int32_t p = G.backward ? gb - 1 - it : ga + it;
      const int lp = G.lev_of[p];
      order.clear();
      bool overflow = false;
      auto add = [&](int32_t c, real v) {
        uint32_t h = ((uint32_t)c * 2654435761u) & (kCap - 1);
        while (key[h] != -1 && key[h] != c) h = (h + 1) & (kCap - 1);
        if (key[h] == -1) {
          if ((int)order.size() >= row_cap) { overflow = true; return; }
          key[h] = c; acc[h] = v; order.push_back({c, (int32_t)h});
        } else {
          acc[h] += v;
        }
      };
      for (int32_t j = base.prow[p]; j < base.prow[p + 1] && !overflow; ++j) {
        const int32_t c = base.pcol[j];
        if (c == p || c >= base.n) continue;  // diagonal; halo entries belong to the pre-pass
        const int lc = G.lev_of[c];
        const bool tri = G.backward ? lc > lp : lc < lp;
        if (!tri) continue;                    // the other triangle belongs to the pre-pass
        const real v = base.pval[j];
        if (c >= ga && c < gb && base.pdiag[c] != 0.0) {  // same group: substitute row c's formula
          const real f = v / base.pdiag[c];
          add((int32_t)(ncols + c), f);
          const int32_t lc2 = c - ga;
          for (int64_t e = start[lc2]; e < start[lc2] + glen[lc2] && !overflow; ++e) add(gcol[e], -f * gval[e]);
        } else {
          add(c, v);  // an earlier group (final), or a row that keeps its x (zero diagonal)
        }
      }
      if (overflow) return INT32_MAX;  // (accumulator left dirty: the caller abandons this m)
      std::sort(order.begin(), order.end());
      const int32_t lp2 = p - ga;
      start[lp2] = (int64_t)gcol.size();
      glen[lp2] = (int32_t)order.size();
      double rowsum = 0.0;
      for (const auto& cs : order) {
        gcol.push_back(cs.first);
        gval.push_back(acc[cs.second]);
        rowsum += std::fabs(acc[cs.second]);
        key[cs.second] = -1;  // release the slot
      }
      if (base.pdiag[p] != 0.0) out->growth = std::max(out->growth, rowsum / std::fabs(base.pdiag[p]));
      max_row = std::max<int64_t>(max_row, (int64_t)order.size());
    }
    // emit the group's rows in ascending row order
    for (int32_t r = 0; r < gn; ++r) {
      out->len[ga - row0 + r] = glen[r];
      out->col.insert(out->col.end(), gcol.begin() + start[r], gcol.begin() + start[r] + glen[r]);
      out->val.insert(out->val.end(), gval.begin() + start[r], gval.begin() + start[r] + glen[r]);
    }
  }
  return max_row;
}



// Fill estimate for group size m from a sample of groups: {entries per row, longest row}
std::pair<double, int64_t> merge_estimate(const HostLevelCsr& base, int64_t ncols, int m, bool backward) {
  MergeGroups G = merge_groups(base, m, backward);
  if (G.ngrp == 0) return {0.0, 0};
  // about 600 k rows in the sample, 8..48 groups, evenly spaced (the estimate only ranks group sizes; the build
  // re-checks the longest row)
  const int64_t rows_per_group = std::max<int64_t>(1, base.n / G.ngrp);
  const int nsample = (int)std::min<int64_t>(G.ngrp, std::max<int64_t>(8, std::min<int64_t>(48, 600000 / rows_per_group)));
  const int T = std::max(1, std::min(merge_threads(), nsample));
  std::vector<int64_t> rows(T, 0), ents(T, 0), mx(T, 0);
  run_threads(T, [&](int t) {
    MergeChunk ch;
    for (int k = t; k < nsample; k += T) {
      const int q = (int)((int64_t)k * G.ngrp / nsample);
      const int64_t r = merge_chunk(base, G, ncols, q, q + 1, &ch);
      mx[t] = std::max(mx[t], r);
      if (r == INT32_MAX) return;
      rows[t] += (int64_t)ch.len.size();
      ents[t] += (int64_t)ch.col.size();
    }
  });
  int64_t R = 0, E = 0, M = 0;
  for (int t = 0; t < T; ++t) { R += rows[t]; E += ents[t]; M = std::max(M, mx[t]); }
  if (M == INT32_MAX) return {1e30, INT32_MAX};
  return {R ? (double)E / R : 0.0, M};
}

MergeResult merge_build(const HostLevelCsr& base, int64_t ncols, int m, bool backward) {
  const int64_t n = base.n;
  MergeResult R;
  HostLevelCsr& S = R.sys;
  MergeGroups G = merge_groups(base, m, backward);
  S.n = n;
  S.pdiag = base.pdiag;
  S.pdpos.assign(n, -1);
  S.nlev = G.ngrp;
  S.lvl_ptr.assign(G.gptr.begin(), G.gptr.end());
  // groups are independent of each other: contiguous ranges of groups (balanced by rows) per thread
  const int T = std::max(1, std::min(merge_threads(), G.ngrp));
  std::vector<int> qcut(T + 1, G.ngrp);
  qcut[0] = 0;
  for (int t = 1; t < T; ++t) {
    const int64_t target = n * t / T;
    qcut[t] = (int)(std::lower_bound(G.gptr.begin(), G.gptr.end(), (int32_t)target) - G.gptr.begin());
    qcut[t] = std::min(std::max(qcut[t], qcut[t - 1]), G.ngrp);
  }
  std::vector<MergeChunk> chunks(T);
  std::vector<int64_t> mx(T, 0);
  run_threads(T, [&](int t) { mx[t] = merge_chunk(base, G, ncols, qcut[t], qcut[t + 1], &chunks[t]); });
  int64_t total = 0;
  for (int t = 0; t < T; ++t) {
    R.max_row = std::max(R.max_row, mx[t]);
    R.growth = std::max(R.growth, chunks[t].growth);
    total += (int64_t)chunks[t].col.size();
  }
  if (R.max_row == INT32_MAX || total >= (int64_t)INT32_MAX - 4096) { R.max_row = INT32_MAX; return R; }
  S.prow.assign(n + 1, 0);
  S.pcol.resize(total);
  S.pval.resize(total);
  int64_t w = 0;
  for (int t = 0; t < T; ++t) {
    const int32_t r0 = G.gptr[qcut[t]];
    for (size_t r = 0; r < chunks[t].len.size(); ++r) S.prow[r0 + r + 1] = chunks[t].len[r];
    std::copy(chunks[t].col.begin(), chunks[t].col.end(), S.pcol.begin() + w);
    std::copy(chunks[t].val.begin(), chunks[t].val.end(), S.pval.begin() + w);
    w += (int64_t)chunks[t].col.size();
    MergeChunk().len.swap(chunks[t].len);
    std::vector<int32_t>().swap(chunks[t].col);
    std::vector<real>().swap(chunks[t].val);
  }
  for (int64_t p = 0; p < n; ++p) S.prow[p + 1] += S.prow[p];
  return R;
}

// Substituted coefficients grow like prod |l_ij / d_j| along the chains inside a group: bounded by 1 for diagonally
// dominant operators; a group whose composite rows outgrow this (relative to the diagonal) is not merged (the
// same 1e-10 contract and reasoning as the condition guard of the block-inverse sweeps).
// (in Float32 the same amplification costs 2^29 times more relative accuracy: the guard tightens with the unit roundoff)
// block-inverse sweeps replace the recurrence inside a block by a dense triangular inverse: accurate to cond * eps.
// 1e4 in Float64; in Float32 the same error budget admits no block worth inverting, the sweeps stay exact-order.
constexpr double kBlockCondMax = sizeof(real) == 8 ? 1e4 : 1e4 * 2.220446049250313e-16 / 1.1920929e-07;
constexpr int64_t kDenseTriMax = 8192;   // whole-triangle dense inverses up to this many rows (2 n^2 reals: 1 GB in Float64)
constexpr int64_t kDenseBlkMax = 65536;  // dense diagonal blocks up to this many rows (above: merged groups / 128-row blocks)
constexpr double kMergeGrowthMax = sizeof(real) == 8 ? 1e4 : 1e2;

// estimated time of one sweep over a grouped system: a kernel boundary per group + streaming its entries
double merge_cost(int64_t ngroups, int64_t nnz) { return ngroups * 3.8e-6 + 12.0 * (double)nnz / 2.5e12; }

}  // namespace
#include "gs_merge_dev.hpp"
#include "gs_relay.hpp"
namespace {

// Build the dependency-level schedule from HOST arrays of the smoother matrix.
struct BuildTimer {  // AMGH_VERBOSE: where the host time of a schedule build goes
  bool on = getenv("AMGH_VERBOSE") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void lap(const char* what, int64_t n) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[amghip] n=%lld build: %-28s %.2f s\n", (long long)n, what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  }
};

void permuted_matrix(int64_t n, int64_t ncols, const int32_t* rowptr, const int32_t* col, const real* val,
                     const std::vector<int32_t>& perm, HostLevelCsr& base);

// dependency levels of the symmetrised pattern, the level order (perm: level-ordered row -> original row) and
// the matrix in that order (columns renumbered, entries of a row in their original order)
void level_order(int64_t n, int64_t ncols, const int32_t* rowptr, const int32_t* col, const real* val,
                 HostLevelCsr& base, std::vector<int32_t>& perm) {
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
  base.n = n;
  base.nlev = (int)(maxlev + 1);
  base.lvl_ptr.assign(base.nlev + 1, 0);
  for (int64_t i = 0; i < n; ++i) base.lvl_ptr[lev[i] + 1]++;
  for (int l = 0; l < base.nlev; ++l) base.lvl_ptr[l + 1] += base.lvl_ptr[l];
  perm.resize(n);
  std::vector<int32_t> next(base.lvl_ptr.begin(), base.lvl_ptr.end() - (base.nlev > 0 ? 1 : 0));
  if (base.nlev == 0) next.clear();
  for (int64_t i = 0; i < n; ++i) perm[next[lev[i]]++] = (int32_t)i;  // ascending row id inside a level
  std::vector<int32_t>().swap(lev);
  permuted_matrix(n, ncols, rowptr, col, val, perm, base);
}

// the matrix with its rows in the order `perm` (position p = row perm[p]) and its columns renumbered to positions
void permuted_matrix(int64_t n, int64_t ncols, const int32_t* rowptr, const int32_t* col, const real* val,
                     const std::vector<int32_t>& perm, HostLevelCsr& base) {
  base.n = n;
  // x is kept in dependency-level order during the sweeps: position p holds x[perm[p]]; columns
  // beyond the square block (halo entries of a sharded operator) keep their place.  Each level
  // then reads and writes contiguous stretches of x (coalesced, TLB-friendly) instead of a
  // hyperplane scattered over the whole vector.
  std::vector<int32_t> inv(std::max<int64_t>(ncols, n));
  for (int64_t c = 0; c < (int64_t)inv.size(); ++c) inv[c] = (int32_t)c;
  for (int64_t p2 = 0; p2 < n; ++p2) inv[perm[p2]] = (int32_t)p2;
  const int64_t nnz = rowptr[n];
  base.prow.resize(n + 1);
  base.pcol.resize(nnz);
  base.pdpos.resize(n);
  base.pval.resize(nnz);
  base.pdiag.resize(n);
  base.prow[0] = 0;
  for (int64_t p = 0; p < n; ++p) base.prow[p + 1] = base.prow[p] + (rowptr[perm[p] + 1] - rowptr[perm[p]]);
  const int T = std::max(1, std::min<int>(merge_threads(), 16));
  run_threads(T, [&](int t) {
    for (int64_t p = n * t / T; p < n * (t + 1) / T; ++p) {
      const int32_t i = perm[p];
      int32_t dp = -1;
      real d = 0.0;
      int64_t w = base.prow[p];
      for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
        base.pcol[w] = inv[col[j]];   // entries stay in the row's original column order (sum order)
        base.pval[w] = val[j];
        if (col[j] == i) { dp = (int32_t)w; d = val[j]; }
        ++w;
      }
      base.pdpos[p] = dp;
      base.pdiag[p] = d;
    }
  });
}

// The word a bounded poll of the chained / dataflow sweeps raises when it gives up (gs_blocks.hpp, gs_flow.hpp): ONE word
// per process in pinned host memory that the kernels write through its device address — the host reads it after any
// synchronisation at no cost (no copy), and every entry point that synchronises turns a raised word into AMGH_ESTATE
// (bw_err_check) instead of handing out the numbers of a sweep that went on with stale values.
inline int32_t* bw_err_word() {
  static int32_t* word = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p) { word = (int32_t*)p; *word = 0; }
    else (void)hipGetLastError();
  });
  return word;
}
// AMGH_ESTATE (and the word lowered again) when a sweep since the last check gave up a poll; call behind a synchronisation
inline int bw_err_check() {
  int32_t* w = bw_err_word();
  if (!w) return AMGH_OK;
  if (*(volatile int32_t*)w == 0) return AMGH_OK;
  // (read and lowered in ONE atomic exchange: of two threads that synchronise at once exactly one reports the give-up,
  // and a word raised between a plain read and a plain store of 0 cannot get lost)
  const int32_t v = __atomic_exchange_n(w, 0, __ATOMIC_ACQ_REL);
  if (v == 0) return AMGH_OK;
  if (getenv("AMGH_VERBOSE")) fprintf(stderr, "[amghip] a block sweep gave up a poll (code %d): AMGH_ESTATE\n", (int)v);
  return AMGH_ESTATE;
}

// The schedule as a WAVEFRONT OF BLOCKS (gs_blocks.hpp) where it pays: rows in block order, one packed record per block,
// a launch per depth of the quotient DAG, one wave walking each block — no substitution, the scalar loop's arithmetic.
// Sets g->bw (and everything the level-ordered cycle needs: perm, the permuted CSR copy, bp / xp) or leaves g untouched
// when the operator is not eligible / the cost model says no.
// (set by amghip_dist.hpp around the schedule build of a row-sharded operator whose sweeps may be pipelined across the ranks)
thread_local const bw::FlowHalo* tl_flow_halo = nullptr;
// (... and the rows of the whole level the operator is a shard of: the size thresholds of the block layouts are about the level —
// a shard below them that keeps the level schedules would be swept in turns instead of in the pipeline)
thread_local int64_t tl_gs_level_rows = 0;

int bw_build(GsSchedule* g, int64_t n, int64_t ncols, const int32_t* rowptr, const int32_t* col, const real* val, BuildTimer& tm, int nrhs_hint = 1) {
  bw::Params prm;
  prm.target_rows = std::max(64, g_gs_bw_rows);
  // (few host threads: the plan is off the setup's critical path, which is a host thread of its own — the sequential C/F
  // splitting — sharing a CPU quota with it: 16 threads here cost the 256^3 setup 0.3 s, 4 do not)
  prm.threads = std::max(1, std::min<int>(merge_threads(), 4));
  if (const char* e = getenv("AMGH_BW_THREADS")) prm.threads = std::max(1, atoi(e));   // (measurement hook)
  // (the cost-model branch below insists on three offset classes, or two on operators of >= gs_bw_two_min_rows rows)
  prm.require_three = g_gs_bw != 2 && !(g_gs_bw_two_min_rows > 0 && n >= g_gs_bw_two_min_rows);
  prm.require_two = g_gs_bw != 2;
  // (rows beyond the records' 18 entries: not eligible — found from the row pointers alone, before any pass over the entries)
  {
    int32_t longest = 0;
    for (int64_t i = 0; i < n; ++i) longest = std::max(longest, rowptr[i + 1] - rowptr[i]);
    if (longest - 1 > bw::kPlanMaxK) return AMGH_OK;
  }
  // the sweep as a dataflow (gs_flow.hpp) where the data dependencies carry the anti-dependencies: structurally symmetric
  // patterns (halo columns of a row-sharded operator: never written, read as they stand).  Blocks of right-hand sides only
  // have that execution (the chained kernel is a single-column one): other patterns keep the level schedules
  const bool flow_ok = g_gs_bw_flow && bw::structurally_symmetric(n, rowptr, col, prm.threads);
  if (nrhs_hint > 1 && !flow_ok) return AMGH_OK;
  // (the default footprint keeps the dataflow layout only: a block's LDS then holds its x, never its record, and the record's
  // bytes do not bound the rows of a block)
  prm.flow_only = flow_ok && gs_trim();
  bw::Plan P;
  try {   // (gigabytes of host memory: out of it, the level keeps the level schedules — nothing crosses the C ABI)
    if (!bw::plan<real>(n, rowptr, col, val, prm, &P)) return AMGH_OK;
  } catch (const std::exception&) {
    if (getenv("AMGH_VERBOSE")) fprintf(stderr, "[amghip] n=%lld wavefront of blocks: the plan ran out of host memory, level schedules kept\n", (long long)n);
    return AMGH_OK;
  }
  tm.lap("block partition + records", n);
  const int64_t nnz = rowptr[n];
  const int nlaunch = (int)P.launch_ptr.size() - 1;
  if (g_gs_bw == 2 && getenv("AMGH_VERBOSE")) {
    const double merged = merge_cost((P.nlevels + 2) / 3, (int64_t)(1.8 * (double)nnz));
    fprintf(stderr, "[amghip] n=%lld wavefront of blocks (forced): %zu blocks, %d depths (dependency levels %d), ranges %d x %d x %d, model %.3f ms (%.3f as one launch per depth) vs %.3f ms merged\n",
            (long long)n, P.blocks.size(), nlaunch, P.nlevels, P.range[0], P.range[1], P.range[2], P.est_chain_seconds * 1e3, P.est_seconds * 1e3, merged * 1e3);
  }
  if (g_gs_bw != 2) {
    // three independent directions (a stencil-like operator: with fewer the blocks are slabs and their walk is long), and
    // clearly cheaper than groups of ~3 merged levels with ~1.8 x the entries (what such operators get otherwise).
    // Two directions (2-D grids): the wavefront is a line of at most sqrt(n) blocks' worth of rows and both models
    // overprice it alike; measured on the 4096^2 Poisson hierarchy (profiles/r04_2d_blocks.log): the 16.8 M-row level
    // 8.5 -> 6.3 ms per smoother, the 8.4 M-row level 7.9 -> 7.4, the 2.1 M-row level 3.7 -> 4.9 — large operators only
    const bool three = P.range[0] > 1 && P.range[1] > 1 && P.range[2] > 1;
    const int ndir = (P.range[0] > 1) + (P.range[1] > 1) + (P.range[2] > 1);
    const bool two = ndir == 2 && g_gs_bw_two_min_rows > 0 && n >= g_gs_bw_two_min_rows;
    const double merged = merge_cost((P.nlevels + 2) / 3, (int64_t)(1.8 * (double)nnz));
    // (the execution the level will get: the dataflow where the pattern allows it, else chained by flags / launched per depth)
    const double est = flow_ok ? P.est_flow_seconds : g_gs_bw_chain ? P.est_chain_seconds : P.est_seconds;
    // (a shard of a row-sharded level that can be swept in the pipeline across the ranks: the alternative to the block layout is
    // not the merged groups but the ranks sweeping in turn)
    const bool used = (three && est < 0.8 * merged) || (two && est < 0.6 * merged) || (three && tl_flow_halo != nullptr);
    if (getenv("AMGH_VERBOSE"))
      fprintf(stderr, "[amghip] n=%lld wavefront of blocks: %zu blocks, %d depths (dependency levels %d), %d directions, model %.3f ms (%s; %.3f as one launch per depth) vs %.3f ms merged -> %s\n",
              (long long)n, P.blocks.size(), nlaunch, P.nlevels, ndir, est * 1e3, flow_ok ? "dataflow" : "chained", P.est_seconds * 1e3, merged * 1e3, used ? "used" : "not used");
    if (!used) return AMGH_OK;
  }
  HostLevelCsr base;
  permuted_matrix(n, ncols, rowptr, col, val, P.perm, base);
  g->n = n; g->ncols = std::max<int64_t>(ncols, n); g->nnz = nnz; g->nlev = P.nlevels;
  g->lvl_ptr.clear(); g->segs.clear();
  RC_TRY(dev_upload(&g->rowptr, base.prow.data(), n + 1));
  RC_TRY(dev_upload(&g->col, base.pcol.data(), nnz));
  RC_TRY(dev_upload(&g->val, base.pval.data(), nnz));
  RC_TRY(dev_upload(&g->dpos, base.pdpos.data(), n));
  RC_TRY(dev_upload(&g->diag, base.pdiag.data(), n));
  g->csr_bytes = (n + 1) * 4 + nnz * kEntB + n * 4;
  g->bytes += g->csr_bytes + n * kRealB;
  RC_TRY(dev_upload(&g->perm, P.perm.data(), n));
  {
    std::vector<int32_t> permx(g->ncols);   // halo columns of a row-sharded operator keep their place behind the rows
    for (int64_t c = 0; c < g->ncols; ++c) permx[c] = c < n ? P.perm[c] : (int32_t)c;
    RC_TRY(dev_upload(&g->permx, permx.data(), g->ncols));
  }
  g->h_perm = P.perm;
  g->bytes += n * 4 + g->ncols * 4;
  RC_TRY(dev_upload(&g->bw.blocks, P.blocks.data(), (int64_t)P.blocks.size()));
  if (P.ext_col.empty()) P.ext_col.push_back(0);   // (a single block has no external column)
  RC_TRY(dev_upload(&g->bw.ext_col, P.ext_col.data(), (int64_t)P.ext_col.size()));
  g->bw.launch_ptr = P.launch_ptr;
  g->bw.err = bw_err_word();
  if (!g->bw.err) return -1001;
  const int64_t B = (int64_t)P.blocks.size();
  RC_TRY(dev_alloc(&g->bw.head, 2));   // ([0]: the chained sweep's ticket counter, [1]: the dataflow sweep's {sweeps, tickets})
  if (hipMemset(g->bw.head, 0, 16) != hipSuccess) return -1001;
  g->bw.nblocks = (int32_t)B;
  if (flow_ok) {
    bw::Flow F;
    bool fok = false;
    // (the default footprint keeps the dataflow layout only: its records are made from the plan's where they lie)
    const bool inplace = gs_trim();
    // (every limit is checked before a record is touched: `false` leaves the plan as it was; an exception — memory — with the
    // records half rewritten leaves no block layout at all: the level schedules take the level)
    try { fok = bw::flow_build<real>(P, prm.threads, &F, inplace, tl_flow_halo, g_gs_bw_dict != 0); }
    catch (const std::exception&) {
      fok = false;
      if (inplace) { g->free_dev(); return AMGH_OK; }
    }
    // (a plan made for the dataflow kernel alone has blocks whose records need not fit LDS: without that layout, the level schedules)
    if (!fok && prm.flow_only) { g->free_dev(); return AMGH_OK; }
    if (fok) {
      tm.lap("dataflow layout", n);
      GsSchedule::Bw::FlowDev& fl = g->bw.flow;
      RC_TRY(dev_upload(&fl.fd, F.fd.data(), (int64_t)F.fd.size()));
      // (the trimmed footprint keeps ONE record layout: the dictionary one where the operator has it — 0.27 instead of 1.34 GB
      // on the 256^3 fine level —, every dataflow kernel has a variant that reads it)
      const bool only_dict = inplace && F.dc.on;
      if (only_dict) fl.srec = nullptr;
      else if (inplace) RC_TRY(dev_upload(&fl.srec, P.rec.data(), (int64_t)P.rec.size()));
      else RC_TRY(dev_upload(&fl.srec, F.srec.data(), (int64_t)F.srec.size()));
      if (F.aux.empty()) F.aux.push_back(0);
      RC_TRY(dev_upload(&fl.aux, F.aux.data(), (int64_t)F.aux.size()));
      RC_TRY(dev_upload(&fl.fl_mb, F.fl_mb.data(), (int64_t)F.fl_mb.size()));
      RC_TRY(dev_upload(&fl.fl_slot, F.fl_slot.data(), (int64_t)F.fl_slot.size()));
      const size_t mbytes = (size_t)(F.nmail + 1024) * bw::Mail<real>::kBytes;
      if (hipMalloc(&fl.mbox, mbytes) != hipSuccess) { (void)hipGetLastError(); return AMGH_ENOMEM; }
      if (hipMemset(fl.mbox, 0, mbytes) != hipSuccess) return -1001;   // (epoch 0 is never a sweep's)
      fl.nmail = F.nmail; fl.lds_max = F.lds_max; fl.mail_stride = (int64_t)mbytes; fl.mcols = 1;
      fl.bytes = (int64_t)F.fd.size() * (int64_t)sizeof(bw::FlowDesc) + (only_dict ? 0 : (int64_t)P.rec.size()) + (int64_t)F.aux.size() * 4 + (int64_t)F.fl_mb.size() * 6 + (int64_t)mbytes;
      fl.on = true;
      fl.late_ok = P.late_ok && g->ncols == n;
      if (F.x.on) {   // the extended lists; their halo entries wait for the neighbours' cells (amgh_dist_finalize patches and uploads xfl_mb)
        RC_TRY(dev_upload(&fl.xaux, F.x.aux.data(), (int64_t)F.x.aux.size()));
        RC_TRY(dev_upload(&fl.xfl_slot, F.x.fl_slot.data(), (int64_t)F.x.fl_slot.size()));
        RC_TRY(dev_upload(&fl.xlist, F.x.list.data(), (int64_t)F.x.list.size()));
        RC_TRY(dev_alloc(&fl.xfl_mb, (int64_t)F.x.fl_mb.size()));
        fl.h_xfl_mb = std::move(F.x.fl_mb); fl.h_row_cell_f = std::move(F.x.row_cell_f); fl.h_row_cell_b = std::move(F.x.row_cell_b);
        fl.xon = true;
        fl.bytes += (int64_t)F.x.aux.size() * 4 + (int64_t)fl.h_xfl_mb.size() * 6 + (int64_t)F.x.list.size() * 4;
      }
      if (F.dc.on) {   // the dictionary layout beside the plain records (a schedule serves one column and blocks of them)
        RC_TRY(dev_upload(&fl.crec, F.dc.crec.data(), (int64_t)F.dc.crec.size()));
        F.dc.dict.resize(F.dc.dict.size() + 16, 0);
        RC_TRY(dev_upload(&fl.dict, F.dc.dict.data(), (int64_t)F.dc.dict.size()));
        RC_TRY(dev_upload(&fl.dict_ent, F.dc.ent.data(), (int64_t)F.dc.ent.size()));
        fl.dict_lds = F.dc.lds_max; fl.dict_on = true;
        fl.bytes += (int64_t)F.dc.crec.size() + (int64_t)F.dc.dict.size() + (int64_t)F.dc.ent.size() * 4;
      }
      g->bytes += fl.bytes;
      tm.lap("dataflow upload", n);
    }
  }
  if (nrhs_hint > 1 && !g->bw.flow.on) { g->free_dev(); return AMGH_OK; }   // (a limit of the dataflow layout: the level schedules)
  // the row-major records, the quotient graph and the flags of the launched / chained sweeps: all of it where the dataflow
  // layout is absent, and under the `full` footprint (run-time tunables then switch between the three executions)
  const bool keep_rec = !g->bw.flow.on || !gs_trim();
  if (keep_rec) {
    RC_TRY(dev_upload(&g->bw.rec, P.rec.data(), (int64_t)P.rec.size()));
    if (P.dep.empty()) P.dep.push_back(0);
    if (P.sdep.empty()) P.sdep.push_back(0);
    RC_TRY(dev_upload(&g->bw.dep_ptr, P.dep_ptr.data(), B + 1));
    RC_TRY(dev_upload(&g->bw.dep, P.dep.data(), (int64_t)P.dep.size()));
    RC_TRY(dev_upload(&g->bw.sdep_ptr, P.sdep_ptr.data(), B + 1));
    RC_TRY(dev_upload(&g->bw.sdep, P.sdep.data(), (int64_t)P.sdep.size()));
    RC_TRY(dev_alloc(&g->bw.flags, B));
    if (hipMemset(g->bw.flags, 0, (size_t)B * 4) != hipSuccess) return -1001;
    g->bytes += (2 * (B + 1) + (int64_t)P.dep.size() + (int64_t)P.sdep.size() + B) * 4 + 12;
  }
  g->bw.on = true;
  g->bw.lds_max = P.lds_max; g->bw.maxk = P.blocks[0].maxk;
  g->bw.rec_bytes = (keep_rec ? (int64_t)P.rec.size() : 0) + g->bw.flow.bytes + (int64_t)P.blocks.size() * (int64_t)sizeof(bw::Desc) + (int64_t)P.ext_col.size() * 4;
  g->bw.rec_entries = n * (int64_t)g->bw.maxk;
  g->bw.sum_depth = P.sum_depth; g->bw.est_seconds = P.est_seconds;
  g->slot_bytes = g->bw.rec_bytes;
  g->bytes += g->bw.rec_bytes - g->bw.flow.bytes;   // (the dataflow layout was counted as it went up)
  g->xstride = g->ncols;
  g->diag_nonzero = true;
  for (int64_t p2 = 0; p2 < n && g->diag_nonzero; ++p2) g->diag_nonzero = base.pdiag[p2] != 0.0;
  RC_TRY(dev_alloc(&g->bp, n));
  RC_TRY(dev_alloc(&g->xp, g->xstride));
  g->bytes += 8 * (n + g->xstride);
  tm.lap("block layout upload", n);
  if (getenv("AMGH_VERBOSE"))
    fprintf(stderr, "[amghip] n=%lld fwd / bwd: wavefront of %zu blocks, %d depths, records %.2f GB%s\n", (long long)n, P.blocks.size(), nlaunch,
            (double)P.rec.size() / 1e9, g->bw.flow.on ? (keep_rec ? ", dataflow layout (+ the chained one)" : ", dataflow layout only") : ", chained (no dataflow layout)");
  return AMGH_OK;
}

// nrhs_hint: right-hand-side columns the sweeps of this schedule will carry (0 = unknown: the level schedules, which serve
// any block size at its best known cost); the wavefront of blocks is chosen for single columns and — as a dataflow, whose
// workgroups carry up to 8 columns past one record stream — for blocks of right-hand sides
int gs_build(GsSchedule* g, int64_t nrows, int64_t ncols, const int32_t* rowptr, const int32_t* col,
             const real* val, int nrhs_hint = 0) {
  const int64_t n = nrows;
  BuildTimer tm;
  g->bytes = 0;
  // (size thresholds measured with the chained kernel, profiles/r03_bw_threshold.log: 7-point rows pay from ~1.5 M rows — 128^3:
  // 7.38 -> 7.22 ms per cycle, 96^3: 4.40 -> 4.52 —, 19-point rows from ~3 M — the 2.0 M-row second level of 160^3: 9.95 -> 10.51 ms)
  const int64_t n_level = std::max<int64_t>(n, tl_gs_level_rows);
  bool bw_size_ok = n_level >= g_gs_bw_min_rows;
  if (!bw_size_ok && n_level >= g_gs_bw_min_rows / 2 && n > 0 && rowptr[n] <= 7 * n) bw_size_ok = true;
  if (g_gs_bw > 0 && (nrhs_hint == 1 || (nrhs_hint > 1 && g_gs_bw_nrhs)) && n > 0 && (g_gs_bw == 2 || bw_size_ok)) {
    const int rcb = bw_build(g, n, ncols, rowptr, col, val, tm, nrhs_hint);
    if (rcb != AMGH_OK) return rcb;
    if (g->bw.on) return AMGH_OK;
  }
  HostLevelCsr base;
  std::vector<int32_t> perm;
  level_order(n, ncols, rowptr, col, val, base, perm);
  const int64_t nnz = rowptr[n];
  g->bytes = 0;
  tm.lap("levels + permuted matrix", n); dbg_pending("levels + permuted matrix");
  RC_TRY(layout_upload(g, base, perm.data()));
  tm.lap("base layout", n); dbg_pending("base layout");
  RC_TRY(dev_upload(&g->perm, perm.data(), n));
  g->h_perm = perm;
  g->ncols = std::max<int64_t>(ncols, n);
  {
    std::vector<int32_t> permx(g->ncols);
    for (int64_t c = 0; c < g->ncols; ++c) permx[c] = c < n ? perm[c] : (int32_t)c;
    RC_TRY(dev_upload(&g->permx, permx.data(), g->ncols));
  }
  g->bytes += n * 4 + g->ncols * 4;
  // Block-inverse path: worth it when level scheduling has degenerated (many more dependency levels
  // than index blocks) and the dense blocks stay small.
  {
    const int nblk = (int)((n + kBlk - 1) / kBlk);
    if (g_gs_block_inverse && n >= 16 && n <= 262144 && g->nlev >= 3 * nblk) {
      std::vector<real> dg(n, 0.0);
      for (int64_t i = 0; i < n; ++i)
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
          if (col[j] == i) dg[i] = val[j];
      double max_cond = 0.0;
      g->super = (g_gs_super > 0 && nblk > g_gs_super && nblk > kBlkSingle) ? g_gs_super : 0;
      RC_TRY(blockgs_build_dir(&g->blk_f, false, g->super, n, rowptr, col, val, dg, &max_cond));
      RC_TRY(blockgs_build_dir(&g->blk_b, true, g->super, n, rowptr, col, val, dg, &max_cond));
      RC_TRY(dev_upload(&g->blk_diag, dg.data(), n));
      RC_TRY(dev_alloc(&g->blk_s, n));
      g->blk_cond = max_cond;
      if (getenv("AMGH_VERBOSE"))
        fprintf(stderr, "[amghip] n=%lld dependency levels=%d index blocks=%d max triangle cond=%.3g -> %s\n", (long long)n,
                g->nlev, nblk, max_cond, max_cond <= kBlockCondMax ? "block-inverse sweeps" : "exact-order sweeps");
      // explicit triangle inverses lose ~cond * eps: keep the 1e-10 contract with margin
      if (max_cond <= kBlockCondMax) {
        g->nblk = nblk;
        g->bytes += 2 * ((int64_t)nblk * kBlk * kBlk * kRealB + nnz * kEntB + (n + 1) * 4) + n * kRealB;
      }
      // Small enough for the triangles of LARGE diagonal blocks to be inverted densely — the whole triangle up to
      // kDenseTriMax rows (2 n^2 reals), blocks of g_gs_dense_blk rows up to kDenseBlkMax rows (8 B n bytes): a sweep is
      // then two launches per block (pre-pass over everything outside the block's triangle on the whole chip, one
      // triangular GEMV) instead of B / 128 sequential steps of one workgroup per superblock, or ~100 merged groups of
      // long-row slots.  256^3 hierarchy: 5 195 rows 0.30 -> 0.05 ms per sweep, 38 260 rows 0.72 -> ~0.2 ms.
      // Same error budget as the 128-row blocks: cond(T_blk) * eps.
      bool all_diag = true;
      for (int64_t i = 0; i < n; ++i) all_diag = all_diag && dg[i] != 0.0;
      if (g->nblk > 0 && g_gs_dense_tri && n <= kDenseBlkMax && ncols <= n && all_diag) {
        const int B = n <= kDenseTriMax ? (int)n : std::max(256, std::min(g_gs_dense_blk, (int)kDenseTriMax));
        const int nb = (int)((n + B - 1) / B);
        g->dti_B = B;
        g->dti_off.assign(nb + 1, 0);
        for (int k = 0; k < nb; ++k) { const int64_t rb = std::min<int64_t>(B, n - (int64_t)k * B); g->dti_off[k + 1] = g->dti_off[k] + rb * rb; }
        int32_t* d_rp = nullptr; int32_t* d_ci = nullptr; real* d_va = nullptr; real* d_rs = nullptr; int64_t* d_off = nullptr;
        int rc3 = dev_upload(&d_rp, rowptr, n + 1);
        if (rc3 == AMGH_OK) rc3 = dev_upload(&d_ci, col, nnz);
        if (rc3 == AMGH_OK) rc3 = dev_upload(&d_va, val, nnz);
        if (rc3 == AMGH_OK) rc3 = dev_upload(&d_off, g->dti_off.data(), nb + 1);
        if (rc3 == AMGH_OK) rc3 = dev_alloc(&d_rs, 2 * n);
        if (rc3 == AMGH_OK) rc3 = dev_alloc(&g->dti_f, g->dti_off[nb]);
        if (rc3 == AMGH_OK) g->dti_b = g->dti_f;   // one square per block: lower triangle forward, upper backward
        double cond = 0.0;
        if (rc3 == AMGH_OK) {
          std::vector<real> rs(2 * n);
          if (dev_zero(g->dti_f, sizeof(real) * (size_t)g->dti_off[nb], nullptr) != hipSuccess) rc3 = -1001;
          if (rc3 == AMGH_OK) {
            hipLaunchKernelGGL(tri_inverse_kernel, dim3((unsigned)((B + 63) / 64), nb, 2), dim3(64), 0, nullptr, (const int32_t*)d_rp,
                               (const int32_t*)d_ci, (const real*)d_va, B, (int)n, (const int64_t*)d_off, g->dti_f, g->dti_b);
            hipLaunchKernelGGL(dense_abs_rowsum_kernel, dim3((unsigned)((B + 3) / 4), nb, 2), dim3(4 * kWave), 0, nullptr,
                               (const real*)g->dti_f, (const real*)g->dti_b, B, (int)n, (const int64_t*)d_off, d_rs, d_rs + n);
            if (hipMemcpy(rs.data(), d_rs, sizeof(real) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) rc3 = -1001;
          }
          for (int dir = 0; dir < 2 && rc3 == AMGH_OK; ++dir) {
            for (int k = 0; k < nb; ++k) {   // ||T_blk||_inf on the host, ||T_blk^-1||_inf from the device rows
              const int64_t r0 = (int64_t)k * B, r1 = std::min<int64_t>(n, r0 + B);
              double nt = 0.0, nx = 0.0;
              for (int64_t i = r0; i < r1; ++i) {
                double r = 0.0;
                for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
                  if (col[j] >= r0 && col[j] < r1 && (dir ? col[j] >= i : col[j] <= i)) r += std::fabs((double)val[j]);
                nt = std::max(nt, r);
                const double xr = (double)rs[(size_t)dir * n + i];
                nx = std::isfinite(xr) ? std::max(nx, xr) : 1e300;
              }
              cond = std::max(cond, nt * nx);
            }
          }
        }
        // the pre-pass matrices in natural order: A without the in-block lower (forward) / upper (backward) triangle and
        // the diagonal — what is left multiplies values that are final (earlier blocks) or still old (the other side)
        for (int dir = 0; dir < 2 && rc3 == AMGH_OK && cond <= kBlockCondMax; ++dir) {
          std::vector<int32_t> tp(n + 1, 0), tc;
          std::vector<real> tv;
          tc.reserve(nnz); tv.reserve(nnz);
          for (int64_t i = 0; i < n; ++i) {
            const int64_t r0 = (i / B) * B, r1 = std::min<int64_t>(n, r0 + B);
            for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
              const bool in_tri = col[j] >= r0 && col[j] < r1 && (dir == 0 ? col[j] <= i : col[j] >= i);
              if (!in_tri) { tc.push_back(col[j]); tv.push_back(val[j]); }
            }
            tp[i + 1] = (int32_t)tc.size();
          }
          GsSchedule::Tri& t = dir ? g->dtri_b : g->dtri_f;
          rc3 = dev_upload(&t.rowptr, tp.data(), n + 1);
          if (rc3 == AMGH_OK) rc3 = dev_upload(&t.col, tc.data(), (int64_t)tc.size());
          if (rc3 == AMGH_OK) rc3 = dev_upload(&t.val, tv.data(), (int64_t)tv.size());
        }
        hipFree(d_rp); hipFree(d_ci); hipFree(d_va); hipFree(d_rs); hipFree(d_off);
        g->dti_cond = cond;
        if (getenv("AMGH_VERBOSE"))
          fprintf(stderr, "[amghip] n=%lld dense triangle inverses, %d block(s) of %d rows: cond %.3g -> %s\n", (long long)n, nb, B, cond,
                  (rc3 == AMGH_OK && cond <= kBlockCondMax) ? "dense triangular sweeps" : "kept the block sweeps");
        if (rc3 != AMGH_OK || !(cond <= kBlockCondMax)) {
          hipFree(g->dti_f); g->dti_f = g->dti_b = nullptr;
          for (GsSchedule::Tri* t : {&g->dtri_f, &g->dtri_b}) { hipFree(t->rowptr); hipFree(t->col); hipFree(t->val); *t = GsSchedule::Tri(); }
          if (rc3 == AMGH_ENOMEM) { rc3 = AMGH_OK; (void)hipGetLastError(); }   // no room for the dense blocks: the 128-row block sweeps stay
          RC_TRY(rc3);
        } else {
          g->bytes += g->dti_off[nb] * kRealB + 2 * (nnz * kEntB + (n + 1) * 4);
        }
      }
    }
  }
  tm.lap("block-inverse data", n); dbg_pending("block-inverse data");
  // Merged levels.  Candidates are compared, per direction, with what would run otherwise: a boundary per
  // dependency level, or the block-inverse sweep (which merged groups with long-row slots replace when cheaper).
  g->xstride = g->ncols;
  if (g_gs_merge > 1 && n >= 4096 && g->nlev >= 64 && (g->nblk == 0 || g_gs_bigslot) && !g->dti_f) {
    const int S = g->super > 0 ? g->super : std::max(1, g->nblk);
    const double block_cost = g->nblk * 5.9e-6 + 2.0 * ((g->nblk + S - 1) / S) * 3.5e-6;
    int chosen_m[2] = {1, 1}, chosen_cap[2] = {kSlot, kSlot};
    const int force_m = (g_gs_merge_force > 1 && (g_gs_merge_force_maxn <= 0 || n <= g_gs_merge_force_maxn)) ? g_gs_merge_force : 0;
    const bool host_merge = getenv("AMGH_HOST_MERGE") != nullptr;  // the host construction (reference for the device one)
    if (!host_merge) {
      // Candidates are BUILT on the device (milliseconds each: gs_merge_dev.hpp) instead of estimated from a sample on
      // the host: exact entry counts and longest rows for the cost model, and the chosen one is already there.
      int32_t* d_lev_of = nullptr;
      RC_TRY(dev_alloc(&d_lev_of, n));
      hipLaunchKernelGGL(lev_of_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const int32_t*)g->d_lvl_ptr,
                         g->nlev, (int)n, d_lev_of);
      MergeDev chosen[2];
      bool sampled[2] = {false, false};
      int rc2 = AMGH_OK;
      // average composite entries per row of every uniform depth that was built (index = depth; 0 = not built)
      std::vector<double> fill[2] = {std::vector<double>(kMergeMaxRounds + 2, 0.0), std::vector<double>(kMergeMaxRounds + 2, 0.0)};
      for (int dir = 0; dir < 2 && rc2 == AMGH_OK; ++dir) {
        const bool backward = dir == 1;
        double best = g->nblk > 0 ? block_cost : merge_cost(base.nlev, nnz);
        int worse = 0;
        // the backward grouping mirrors the forward one (same pattern, levels counted from the other end): its best
        // group size sits next to the forward one, so only that neighbourhood is built
        // (if nothing is accepted there — e.g. its rows outgrow the slots earlier — the search restarts from 2)
        for (int attempt = 0; attempt < 2 && rc2 == AMGH_OK; ++attempt) {
        int m_lo = (attempt == 0 && dir == 1 && chosen_m[0] > 2) ? chosen_m[0] - 1 : 2;
        if (attempt == 1 && (chosen_m[dir] > 1 || !(dir == 1 && chosen_m[0] > 2))) break;
        worse = 0;
        int64_t prev_max = 0;  // longest composite row of the previous (shallower) candidate: rows only grow with m
        int m_hi = std::min(g_gs_merge, kMergeMaxRounds);
        if (force_m > 1) m_lo = m_hi = std::min(force_m, kMergeMaxRounds);   // measurement hook: exactly this depth
        for (int m = m_lo; m <= m_hi; ++m) {
          MergeDev md;
          // candidates are COST-MODEL builds: every stride-th group only (entries per row, longest row and growth are bulk
          // properties of a grouping) — all but one of them are thrown away; the chosen one is built in full below
          const int stride = (force_m > 1 || !g_gs_sample) ? 1 : std::max(1, std::min(8, ((g->nlev + m - 1) / m) / 12));
          rc2 = merge_build_dev(g, d_lev_of, g->diag, MergeGrouping::uniform(g->nlev, m, backward), &md,
                                prev_max > 440 ? 2 : prev_max > 110 ? 1 : 0, stride);
          prev_max = md.max_row;
          if (rc2 != AMGH_OK) break;
          const bool md_sampled = stride > 1;
          if (stride > 1 && !md.failed)   // scale the sample to the operator
            md.total = (int64_t)((double)md.total * (double)n / (double)std::max<int64_t>(1, md.sampled_rows));
          if (!md.failed) fill[dir][m] = (double)md.total / (double)n;
          if (md.failed || md.max_row > kBigSlot - kBigSlot / 8) {  // fill has exploded
            if (getenv("AMGH_VERBOSE")) fprintf(stderr, "[amghip] n=%lld %s m=%d given up: failed %d, max row %lld\n", (long long)n, backward ? "bwd" : "fwd", m, (int)md.failed, (long long)md.max_row);
            md.free_dev();
            break;
          }
          int cap = kSlot;
          if (md.max_row > kSlot || g_gs_bigslot == 2) cap = kBigSlot;  // (2 = always: test hook)
          if (cap == kBigSlot && !g_gs_bigslot) { md.free_dev(); break; }
          if (md.growth > kMergeGrowthMax) {  // substitution amplifies on this operator: no deeper groups
            if (getenv("AMGH_VERBOSE"))
              fprintf(stderr, "[amghip] n=%lld %s merge m=%d rejected: coefficient growth %.3g\n", (long long)n,
                      backward ? "bwd" : "fwd", m, md.growth);
            md.free_dev();
            break;
          }
          const int ngrp = (base.nlev + m - 1) / m;
          // + the pre-pass; a long-row launch costs ~2.5 us more (a wave per row, rows of very different lengths)
          // long-row slots: the 2 048-entry slot kernel costs ~2.5 us more per launch; groups whose rows average >= 96
          // entries run from the SELL-like copy instead (sell_build), where the surcharge is ~0.5 us (measured on the
          // 228 k-row level of the 256^3 hierarchy: groups of 8-9 levels 1.45 ms per pass, 5-6 levels 1.78 ms)
          const bool sell_rows = g_gs_sell && !gs_lean() && (double)md.total / (double)n >= 96.0;
          const double c = merge_cost(ngrp, md.total) + 12.0 * (double)nnz / 2 / 4e12 +
                           (cap == kBigSlot ? ngrp * (sell_rows ? 0.5e-6 : 2.5e-6) : 0.0);
          if (getenv("AMGH_VERBOSE"))
            fprintf(stderr, "[amghip] n=%lld %s merge m=%d: %d groups, %.1f entries/row (max %lld), est. %.2f ms vs %.2f ms\n",
                    (long long)n, backward ? "bwd" : "fwd", m, ngrp, (double)md.total / n, (long long)md.max_row, 1e3 * c, 1e3 * best);
          const bool stream_bound = 12.0 * (double)md.total / 2.5e12 > best;  // streaming the composite rows alone costs more
          if (c < 0.97 * best || force_m > 1) {
            best = c; chosen_m[dir] = m; chosen_cap[dir] = cap; worse = 0;
            chosen[dir].free_dev();
            chosen[dir] = std::move(md);
            sampled[dir] = md_sampled;   // (of the candidate that is KEPT: the stride varies with the number of groups)
            md = MergeDev();
          } else {
            md.free_dev();
            // well past the minimum — but the step to long-row slots is a bump, not the end: operators that compete with
            // the block-inverse path (small, long rows) only pay off at deep groups, keep looking there
            // (large operators: the cost curve is smooth and every further candidate costs a full build of more entries)
            if (++worse >= (g->nblk > 0 ? 8 : 2)) break;
          }
          if (stream_bound) break;
        }
        }
      }
      tm.lap("merge candidates (device)", n); dbg_pending("merge candidates (device)");
      const bool use = g->nblk == 0 ? true : (chosen_m[0] > 1 && chosen_m[1] > 1);
      // Groups of DIFFERENT depth along the sweep.  Where the dependency levels are small — the two ends of a sweep over
      // a grid-like operator — a launch is at its latency floor whatever it carries, so deeper groups there are nearly
      // free, while in the middle the uniform optimum stands.  Per level: the depth d minimising
      //   t0 / d + b * rows(level) * fill(d)
      // with fill(d) measured on the uniform candidates above and extrapolated geometrically beyond them; groups are
      // then cut greedily along the sweep (a group is as deep as every level in it tolerates), BUILT exactly, and kept
      // when the per-group cost model says they win.  Measured (profiles/r02_zone_sweep.log): -12 % launches on the two
      // finest levels for -3 % sweep time; pricing a launch higher (deeper middle) LOSES: the long composite rows of deep
      // groups cost more than the launches they save.
      for (int dir = 0; dir < 2 && rc2 == AMGH_OK && use && g_gs_zone && force_m <= 1; ++dir) {
        const bool backward = dir == 1;
        if (chosen_m[dir] <= 1 || chosen[dir].failed) continue;
        const int depth_cap = std::min(g_gs_merge, kMergeMaxRounds);
        if (depth_cap <= chosen_m[dir]) continue;
        std::vector<double> fl(depth_cap + 1, 0.0);
        fl[1] = 1.0 + 0.5 * std::max<double>(0.0, (double)nnz / (double)n - 1.0);    // one triangle + the s column
        int last = 1;
        for (int d = 2; d <= depth_cap; ++d) {
          double f = fill[dir][d] > 0.0 ? fill[dir][d] : fill[1 - dir][d];   // (the mirrored direction fills alike)
          if (f > 0.0) { fl[d] = f; last = d; }
          else {
            const double ratio = last >= 2 && fl[last - 1] > 0.0 ? std::max(1.05, fl[last] / fl[last - 1]) : 1.6;
            fl[d] = fl[d - 1] * ratio;
          }
        }
        // the constants of merge_cost (fitted on whole sweeps of all levels): per byte they are pessimistic for the fine
        // level (its gathers are regular, ~5 TB/s while streaming) and right for the coarser ones (~2.6 TB/s) — a per-entry
        // price that is too LOW deepens the middle of a sweep and loses (measured: 32.1 -> 34.1 ms with 6 us + 4.8 TB/s)
        // and below ~6 MB a launch does not get cheaper any more: 4.7 us of kernel + the boundary (the same probe)
        const double t0 = 1e-9 * g_gs_zone_t0_ns, bsec = 12.0 / 2.5e12, floor_s = 1e-9 * g_gs_zone_floor_ns;
        auto group_cost = [&](int64_t entries, bool big) {
          return std::max(floor_s, t0 + bsec * (double)entries) + (big ? 2.5e-6 : 0.0);
        };
        auto sweep_cost = [&](const MergeDev& md, const MergeGrouping& G, bool big) {
          double c = 0.0;
          const int st = md.sample_stride;   // a sampled (cost-model) build holds every st-th group: scale per entry per row
          int64_t se = 0, sr = 0;
          if (st > 1)
            for (int q = st / 2; q < G.ngrp(); q += st)
              for (int32_t p2 = base.lvl_ptr[G.gb[q]]; p2 < base.lvl_ptr[G.gb[q + 1]]; ++p2) { se += md.h_clen[p2]; ++sr; }
          const double per_row = sr > 0 ? (double)se / (double)sr : 0.0;
          for (int q = 0; q < G.ngrp(); ++q) {
            int64_t e = 0;
            if (st > 1 && q % st != st / 2) e = (int64_t)(per_row * (double)(base.lvl_ptr[G.gb[q + 1]] - base.lvl_ptr[G.gb[q]]));
            else
              for (int32_t p2 = base.lvl_ptr[G.gb[q]]; p2 < base.lvl_ptr[G.gb[q + 1]]; ++p2) e += md.h_clen[p2];
            c += group_cost(e, big);
          }
          return c;
        };
        const bool big = chosen_cap[dir] == kBigSlot;
        const int64_t row_limit = big ? kBigSlot - kBigSlot / 8 : kSlot - kSlot / 8;
        if (chosen[dir].max_row > row_limit / 2) continue;       // deeper groups would outgrow the slots: nothing to try
        const double uniform_cost = sweep_cost(chosen[dir], chosen[dir].grouping, big);
        int cap_try = std::min(depth_cap, 2 * chosen_m[dir] + 2);   // (deeper than that has never fitted a 512-entry slot)
        for (int attempt = 0; attempt < 4 && rc2 == AMGH_OK; ++attempt) {
          // per level (in sweep order) the depth it would like
          std::vector<int> want(g->nlev, 1);
          for (int sidx = 0; sidx < g->nlev; ++sidx) {
            const int l = backward ? g->nlev - 1 - sidx : sidx;
            const double rows = (double)(base.lvl_ptr[l + 1] - base.lvl_ptr[l]);
            double bestc = 1e300;
            for (int d = 1; d <= cap_try; ++d) {
              const double c = std::max(floor_s, t0 + bsec * rows * d * fl[d]) / d;
              if (c < bestc * (1.0 - 1e-9)) { bestc = c; want[sidx] = d; }
            }
            want[sidx] = std::max(want[sidx], std::min(chosen_m[dir], cap_try));   // never shallower than the uniform choice
          }
          std::vector<int32_t> cuts{0};
          for (int sidx = 0; sidx < g->nlev;) {
            int d = std::min(want[sidx], g->nlev - sidx);
            for (;;) {
              int tol = d;
              for (int t = sidx; t < sidx + d; ++t) tol = std::min(tol, want[t]);
              if (tol >= d || d <= 1) break;
              d = std::max(tol, 1);
            }
            sidx += d;
            cuts.push_back(sidx);
          }
          MergeGrouping Z;
          Z.backward = backward;
          Z.gb.resize(cuts.size());
          for (size_t k = 0; k < cuts.size(); ++k) Z.gb[k] = backward ? g->nlev - cuts[cuts.size() - 1 - k] : cuts[k];
          if (Z.depth() <= chosen_m[dir]) break;                 // nothing deeper anywhere: the uniform grouping stands
          MergeDev md;
          rc2 = merge_build_dev(g, d_lev_of, g->diag, Z, &md, chosen[dir].max_row > 440 ? 2 : chosen[dir].max_row > 110 ? 1 : 0);
          if (rc2 != AMGH_OK) break;
          const bool fits = !md.failed && md.max_row <= row_limit && !(md.growth > kMergeGrowthMax);
          const double zc = fits ? sweep_cost(md, Z, big) : 1e300;
          if (getenv("AMGH_VERBOSE"))
            fprintf(stderr, "[amghip] n=%lld %s zoned groups (depth <= %d): %d groups (uniform m=%d: %d), %.1f entries/row (max %lld)%s, "
                    "est. %.2f ms vs %.2f ms\n", (long long)n, backward ? "bwd" : "fwd", cap_try, Z.ngrp(), chosen_m[dir],
                    chosen[dir].ngrp, md.failed ? 0.0 : (double)md.total / n, (long long)md.max_row, fits ? "" : " [does not fit]",
                    fits ? 1e3 * zc : -1.0, 1e3 * uniform_cost);
          if (fits && zc < 0.97 * uniform_cost) {
            chosen[dir].free_dev();
            chosen[dir] = std::move(md);
            sampled[dir] = false;      // (a full build)
            md = MergeDev();
            break;
          }
          md.free_dev();
          if (fits) break;                                       // it fits and still does not win: deeper will not either
          cap_try = std::max(chosen_m[dir] + 1, (cap_try * 3) / 4);
          if (cap_try <= chosen_m[dir]) break;
        }
      }
      // where no zoned grouping replaced it, the chosen uniform depth in full (the candidates above were samples: all but
      // one of them are thrown away, and so is the uniform one when the zoned grouping wins); a row the sample did not
      // see may outgrow the slots: then one level fewer per group
      for (int dir = 0; dir < 2 && rc2 == AMGH_OK; ++dir) {
        if (!sampled[dir] || chosen_m[dir] <= 1) continue;
        const bool backward = dir == 1;
        const int64_t smax = chosen[dir].max_row;
        chosen[dir].free_dev();
        chosen[dir] = MergeDev();
        while (chosen_m[dir] > 1 && rc2 == AMGH_OK) {
          MergeDev md;
          rc2 = merge_build_dev(g, d_lev_of, g->diag, MergeGrouping::uniform(g->nlev, chosen_m[dir], backward), &md,
                                smax > 440 ? 2 : smax > 110 ? 1 : 0);
          if (rc2 != AMGH_OK) break;
          int cap = (md.max_row > kSlot || g_gs_bigslot == 2) ? kBigSlot : kSlot;
          // (a row the sample missed that needs the 2048-entry slots where the sample was priced on 512-entry ones:
          // one level fewer per group costs a few per cent, the long-row kernel on short rows +40 % — the 495 902-row
          // level of the smoothed-aggregation 160^3 hierarchy, 2.71 -> 2.2 ms per pass)
          const bool outgrew = cap == kBigSlot && smax <= kSlot && g_gs_bigslot != 2;
          const bool ok = !md.failed && md.max_row <= kBigSlot - kBigSlot / 8 && !(cap == kBigSlot && !g_gs_bigslot) &&
                          !(md.growth > kMergeGrowthMax) && !outgrew;
          if (ok) {
            if (getenv("AMGH_VERBOSE"))
              fprintf(stderr, "[amghip] n=%lld %s merge m=%d built in full: %.1f entries/row (max %lld)\n", (long long)n,
                      backward ? "bwd" : "fwd", chosen_m[dir], (double)md.total / n, (long long)md.max_row);
            fill[dir][chosen_m[dir]] = (double)md.total / (double)n;
            chosen_cap[dir] = cap;
            chosen[dir] = std::move(md);
            break;
          }
          md.free_dev();
          --chosen_m[dir];
        }
      }
      tm.lap("zoned groups (device)", n); dbg_pending("zoned groups (device)");
      for (int dir = 0; dir < 2 && rc2 == AMGH_OK; ++dir) {
        const bool backward = dir == 1;
        MergeDev& md = chosen[dir];
        if (!use || chosen_m[dir] <= 1) { md.free_dev(); continue; }
        const int cap = chosen_cap[dir];
        // levels per group as reported: the average depth of the grouping in use
        const int best_m = std::max(1, (int)std::lround((double)g->nlev / std::max(1, md.ngrp)));
        // the grouped system: groups as levels, composite rows gathered into one CSR on the device
        HostLevelCsr sys;
        struct { int ngrp; std::vector<int32_t> gptr; } G;
        G.ngrp = md.ngrp;
        G.gptr.resize(md.ngrp + 1);
        for (int q = 0; q <= md.ngrp; ++q) G.gptr[q] = base.lvl_ptr[md.grouping.gb[q]];
        sys.n = n;
        sys.pdiag = base.pdiag;
        sys.pdpos.assign(n, -1);
        sys.nlev = G.ngrp;
        sys.lvl_ptr.assign(G.gptr.begin(), G.gptr.end());
        sys.prow.assign(n + 1, 0);
        for (int64_t p2 = 0; p2 < n; ++p2) sys.prow[p2 + 1] = sys.prow[p2] + md.h_clen[p2];
        DevCsr src;
        rc2 = dev_upload(&src.rowptr, sys.prow.data(), n + 1);
        if (rc2 == AMGH_OK) rc2 = dev_alloc(&src.col, md.total);
        if (rc2 == AMGH_OK) rc2 = dev_alloc(&src.val, md.total);
        if (rc2 == AMGH_OK) {
          MergeArgs a{};
          a.n = (int)n; a.clen = md.clen; a.coff = md.coff; a.lev_of = d_lev_of; a.nlev = g->nlev;
          a.backward = backward ? 1 : 0; a.ngrp = md.ngrp;
          a.grp_of_lev = md.d_grp_of_lev; a.rnd_of_lev = md.d_rnd_of_lev;
          for (int q = 0; q < kMergeMaxRounds; ++q) { a.rcol[q] = md.rcol[q]; a.rval[q] = md.rval[q]; }
          hipLaunchKernelGGL(merge_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, a,
                             (const int32_t*)src.rowptr, src.col, src.val);
          if (hipDeviceSynchronize() != hipSuccess) rc2 = -1001;
        }
        const int64_t max_row = md.max_row;
        md.free_dev();
        if (rc2 == AMGH_OK) {
          GsSchedule* ch = new GsSchedule;
          (backward ? g->mb : g->mf) = ch;
          ch->ncols = g->ncols;
          rc2 = layout_upload(ch, sys, perm.data(), cap, gs_lean(), &src, g->diag);
          if (rc2 == AMGH_OK && g_gs_sell && !gs_lean()) rc2 = sell_build(ch, sys.prow);
          if (rc2 == AMGH_OK && gs_trim()) rc2 = child_compact_late(ch, sys.prow, perm.data());
          tm.lap("merged layout", n); dbg_pending("merged layout");
          if (rc2 == AMGH_OK && !(backward && gs_trim()))   // (trimmed: the backward triangle on first use, csr_gs_sweep)
            rc2 = tri_build_dev(backward ? &g->tri_b : &g->tri_f, g, backward, &g->bytes, 0.0,
                                backward ? &g->tri_nnz_b : &g->tri_nnz);
          tm.lap("other triangle", n); dbg_pending("other triangle");
          (backward ? g->merge_b : g->merge_f) = best_m;
          g->bytes += ch->bytes;
          if (getenv("AMGH_VERBOSE"))
            fprintf(stderr, "[amghip] n=%lld %s: groups of %d levels, %d-entry slots, longest row %lld\n", (long long)n,
                    backward ? "bwd" : "fwd", best_m, cap, (long long)max_row);
        }
        hipFree(src.rowptr); hipFree(src.col); hipFree(src.val);
      }
      for (MergeDev& md : chosen) md.free_dev();
      hipFree(d_lev_of);
      RC_TRY(rc2);
    } else {
    for (int dir = 0; dir < 2; ++dir) {
      const bool backward = dir == 1;
      double best = g->nblk > 0 ? block_cost : merge_cost(base.nlev, nnz);      int worse = 0;
      for (int m = 2; m <= g_gs_merge; ++m) {  // fill estimated on a sample of groups
        const std::pair<double, int64_t> est = merge_estimate(base, g->ncols, m, backward);
        int cap = kSlot;
        if (est.second > kSlot - kSlot / 8 || g_gs_bigslot == 2) cap = kBigSlot;  // (2 = always: test hook)
        if (est.second > kBigSlot - kBigSlot / 8 || (cap == kBigSlot && !g_gs_bigslot)) break;  // fill has exploded
        const int ngrp = (base.nlev + m - 1) / m;
        // + the pre-pass; a long-row launch costs ~2.5 us more (a wave per row, rows of very different lengths)
        const double c = merge_cost(ngrp, (int64_t)(est.first * n)) + 12.0 * (double)nnz / 2 / 4e12 +
                         (cap == kBigSlot ? ngrp * 2.5e-6 : 0.0);
        if (getenv("AMGH_VERBOSE"))
          fprintf(stderr, "[amghip] n=%lld %s merge m=%d: %d groups, ~%.1f entries/row (max %lld), est. %.2f ms vs %.2f ms\n",
                  (long long)n, backward ? "bwd" : "fwd", m, ngrp, est.first, (long long)est.second, 1e3 * c, 1e3 * best);
        if (c < 0.97 * best) { best = c; chosen_m[dir] = m; chosen_cap[dir] = cap; worse = 0; }
        else if (++worse >= 4) break;  // well past the minimum (the step to long-row slots is a bump, not the end)
        if (12.0 * est.first * n / 2.5e12 > best) break;  // streaming the composite rows alone costs more: fill only grows with m
      }
    }
    tm.lap("merge estimates", n);
    // an operator on the block path keeps it unless both directions are cheaper merged
    const bool use = g->nblk == 0 ? true : (chosen_m[0] > 1 && chosen_m[1] > 1);
    for (int dir = 0; dir < 2 && use; ++dir) {
      const bool backward = dir == 1;
      int best_m = chosen_m[dir], cap = chosen_cap[dir];
      MergeResult keep;
      while (best_m > 1) {
        keep = merge_build(base, g->ncols, best_m, backward);
        tm.lap("merge build", n);
        if (keep.max_row > cap && cap == kSlot && g_gs_bigslot && keep.max_row <= kBigSlot) cap = kBigSlot;
        if (keep.max_row <= cap && !(keep.growth > kMergeGrowthMax)) break;
        if (keep.growth > kMergeGrowthMax && getenv("AMGH_VERBOSE"))
          fprintf(stderr, "[amghip] n=%lld %s merge m=%d rejected: coefficient growth %.3g\n", (long long)n,
                  backward ? "bwd" : "fwd", best_m, keep.growth);
        --best_m;  // the sample missed a row that outgrows a slot: one level fewer per group
      }
      if (best_m > 1) {
        GsSchedule* ch = new GsSchedule;
        (backward ? g->mb : g->mf) = ch;
        ch->ncols = g->ncols;
        RC_TRY(layout_upload(ch, keep.sys, perm.data(), cap, gs_trim(), nullptr, g->diag));
        tm.lap("merged layout", n); dbg_pending("merged layout");
        if (!(backward && gs_trim()))
          RC_TRY(tri_build_dev(backward ? &g->tri_b : &g->tri_f, g, backward, &g->bytes, 0.0,
                               backward ? &g->tri_nnz_b : &g->tri_nnz));
        tm.lap("other triangle", n); dbg_pending("other triangle");
        (backward ? g->merge_b : g->merge_f) = best_m;
        g->bytes += ch->bytes;
        if (getenv("AMGH_VERBOSE"))
          fprintf(stderr, "[amghip] n=%lld %s: groups of %d levels, %d-entry slots, longest row %lld\n", (long long)n,
                  backward ? "bwd" : "fwd", best_m, cap, (long long)keep.max_row);
      }
    }
    }
    if (g->nblk > 0 && g->mf && g->mb) {  // merged groups replace the block-inverse sweeps
      for (GsSchedule::Outer* o : {&g->blk_f, &g->blk_b}) {
        hipFree(o->rowptr); hipFree(o->col); hipFree(o->val); hipFree(o->tinv);
        hipFree(o->near_ptr); hipFree(o->near_pi); hipFree(o->near_val);
        hipFree(o->nx_rowptr); hipFree(o->nx_col); hipFree(o->nx_val);
        hipFree(o->sp_rowptr); hipFree(o->sp_col); hipFree(o->sp_val);
        *o = GsSchedule::Outer();
      }
      g->bytes -= 2 * ((int64_t)g->nblk * kBlk * kBlk * kRealB + nnz * kEntB + (n + 1) * 4);
      g->nblk = 0;
    } else if (g->nblk > 0) {  // stays on the block path: drop half-built children
      for (GsSchedule** c : {&g->mf, &g->mb})
        if (*c) { g->bytes -= (*c)->bytes; (*c)->free_dev(); delete *c; *c = nullptr; }
    }
    if (g->mf || g->mb) g->xstride = g->ncols + n;
    if (gs_trim() && g->mf && g->mb && g->nblk == 0 && g->wcol) {
      // trimmed footprint: both directions always run merged groups, the un-merged slot copy would only serve a run with
      // merging switched off at run time — that run falls back to the CSR stream kernel
      hipFree(g->wcol); hipFree(g->wval); hipFree(g->slot_row); hipFree(g->wmeta);
      g->wcol = g->slot_row = nullptr; g->wval = nullptr; g->wmeta = nullptr;
      for (auto& sg : g->segs) if (!sg.chain) { sg.nslots = -1; sg.slot0 = 0; }
      g->bytes -= g->slot_bytes;
      g->slot_bytes = 0; g->slot_total = 0;
    }
  }
  g->diag_nonzero = true;
  for (int64_t p2 = 0; p2 < n && g->diag_nonzero; ++p2) g->diag_nonzero = base.pdiag[p2] != 0.0;
  RC_TRY(dev_alloc(&g->bp, n));
  RC_TRY(dev_alloc(&g->xp, g->xstride));
  g->bytes += 8 * (n + g->xstride);
  return AMGH_OK;
}

// x and s of a merged system share one vector [x (ncols) ; s (n)] per column: re-lay out xp the first time an operator
// gets merged children after its schedule was built (SOR children).  Frees xp: callers guarantee no x lives there.
int gs_grow_xp_for_merged(GsSchedule* g, int64_t* op_bytes) {
  if (g->xstride != g->ncols) return AMGH_OK;
  ++g_sched_epoch;
  hipFree(g->xp); g->xp = nullptr;
  g->xstride = g->ncols + g->n;
  RC_TRY(dev_alloc(&g->xp, g->xstride * g->cols_alloc));
  g->bytes += 8 * g->n * g->cols_alloc;
  if (op_bytes) *op_bytes += 8 * g->n * g->cols_alloc;
  return AMGH_OK;
}

// Merged children for SOR with relaxation factor omega, built on demand from the level-ordered matrix already on
// the device.  Returns the cache entry (children may be null: merging did not pay or was rejected).
GsSchedule::SorSet* sor_children(GsSchedule* g, real omega) {
  if (g->bw.on) return nullptr;   // the wavefront of blocks sweeps SOR itself (no merged children)
  for (GsSchedule::SorSet& ss : g->sor)
    if (ss.built && ss.omega == omega) return &ss;
  GsSchedule::SorSet& ss = g->sor[g->sor_next];
  g->sor_next ^= 1;
  if (ss.f || ss.b) ++g_sched_epoch;  // an evicted set may be referenced by captured graphs
  for (GsSchedule** c : {&ss.f, &ss.b})
    if (*c) { g->bytes -= (*c)->bytes; (*c)->free_dev(); delete *c; *c = nullptr; }
  for (GsSchedule::Tri* t : {&ss.tf, &ss.tb}) { hipFree(t->rowptr); hipFree(t->col); hipFree(t->val); *t = GsSchedule::Tri(); }
  ss.omega = omega;
  ss.built = true;
  const int64_t n = g->n;
  if (!(g_gs_merge > 1 && n >= 4096 && g->nlev >= 64) || omega == 0.0) return &ss;
  BuildTimer tm;
  HostLevelCsr base;
  base.n = n;
  base.nlev = g->nlev;
  base.lvl_ptr = g->lvl_ptr;
  base.prow.resize(n + 1);
  if (hipMemcpy(base.prow.data(), g->rowptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost) != hipSuccess) return &ss;
  const int64_t nnz = base.prow[n];
  base.pcol.resize(nnz); base.pval.resize(nnz); base.pdpos.resize(n); base.pdiag.resize(n);
  if (hipMemcpy(base.pcol.data(), g->col, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(base.pval.data(), g->val, sizeof(real) * nnz, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(base.pdpos.data(), g->dpos, sizeof(int32_t) * n, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(base.pdiag.data(), g->diag, sizeof(real) * n, hipMemcpyDeviceToHost) != hipSuccess)
    return &ss;
  HostLevelCsr scaled = base;               // the triangular system SOR solves has the diagonal D / omega
  for (real& dd : scaled.pdiag) dd /= omega;
  for (int dir = 0; dir < 2; ++dir) {
    const bool backward = dir == 1;
    double best = merge_cost(base.nlev, nnz);
    int best_m = 1, cap = kSlot, worse = 0;
    for (int m = 2; m <= g_gs_merge; ++m) {
      const std::pair<double, int64_t> est = merge_estimate(scaled, g->ncols, m, backward);
      int cap_m = kSlot;
      if (est.second > kSlot - kSlot / 8 || g_gs_bigslot == 2) cap_m = kBigSlot;
      if (est.second > kBigSlot - kBigSlot / 8 || (cap_m == kBigSlot && !g_gs_bigslot)) break;
      const int ngrp = (base.nlev + m - 1) / m;
      const double c = merge_cost(ngrp, (int64_t)(est.first * n)) + 12.0 * (double)nnz / 2 / 4e12 +
                       (cap_m == kBigSlot ? ngrp * 2.5e-6 : 0.0);
      if (c < 0.97 * best) { best = c; best_m = m; cap = cap_m; worse = 0; }
      else if (++worse >= 4) break;
      if (12.0 * est.first * n / 2.5e12 > best) break;
    }
    MergeResult keep;
    while (best_m > 1) {
      keep = merge_build(scaled, g->ncols, best_m, backward);
      if (keep.max_row > cap && cap == kSlot && g_gs_bigslot && keep.max_row <= kBigSlot) cap = kBigSlot;
      if (keep.max_row <= cap && !(keep.growth > kMergeGrowthMax)) break;
      --best_m;
    }
    if (best_m > 1) {
      GsSchedule* ch = new GsSchedule;
      ch->ncols = g->ncols;
      if (layout_upload(ch, keep.sys, nullptr, cap) != AMGH_OK ||
          tri_build_dev(backward ? &ss.tb : &ss.tf, g, backward, &g->bytes, (1.0 - omega) / omega, nullptr) != AMGH_OK) {
        ch->free_dev();
        delete ch;
        continue;
      }
      (backward ? ss.b : ss.f) = ch;
      g->bytes += ch->bytes;
      if (getenv("AMGH_VERBOSE"))
        fprintf(stderr, "[amghip] n=%lld SOR(%.3g) %s: groups of %d levels, %d-entry slots, longest row %lld\n", (long long)n,
                omega, backward ? "bwd" : "fwd", best_m, cap, (long long)keep.max_row);
    }
  }
  tm.lap("SOR merged children", n);
  return &ss;
}

int csr_ensure_gs(amgh_csr* op) {
  if (op->gs) return AMGH_OK;
  const int64_t n = op->nrows;
  std::vector<int32_t> rowptr(n + 1), col(op->nnz);
  std::vector<real> val(op->nnz);
  HIP_TRY(hipMemcpy(rowptr.data(), op->rowptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));
  if (op->nnz) {
    HIP_TRY(hipMemcpy(col.data(), op->col, sizeof(int32_t) * op->nnz, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(val.data(), op->val, sizeof(real) * op->nnz, hipMemcpyDeviceToHost));
  }
  GsSchedule* g = new GsSchedule;
  int rc = gs_build(g, n, op->ncols, rowptr.data(), col.data(), val.data(), op->gs_nrhs_hint);
  if (rc != AMGH_OK) { g->free_dev(); delete g; return rc; }
  op->gs = g;
  op->bytes += g->bytes;
  return AMGH_OK;
}

}  // namespace
