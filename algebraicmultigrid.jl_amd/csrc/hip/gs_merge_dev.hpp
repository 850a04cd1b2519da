// gs_merge_dev.hpp — composite rows of the merged dependency-level groups (gs_schedule.hpp, merge_build) built ON THE
// DEVICE from the schedule's level-ordered copy of the matrix.  Included by gs_schedule.hpp.
//
// The host construction substitutes row by row inside a group; here a group of m levels is m ROUNDS: round k forms
// the composite rows of every row whose level is the k-th of its group — all groups at once, they are independent —
// from the composite rows of the rounds before.  A lane group owns one row and an LDS hash table (column -> coefficient);
// the row's entries are taken IN ORDER, and for one entry the lanes add distinct columns (the substituted row's
// entries + its right-hand-side column), so every coefficient is accumulated in exactly the order of the host loop:
// the device result equals merge_build's bit for bit (tests compare the two).  Rows come out sorted by column.
// Two launch shapes: 16 lanes per row and a 512-slot table (composite rows up to ~440 entries); rows that outgrow it
// are redone by a whole 256-thread workgroup with a 4096-slot table (long-row slots hold up to 2048 entries).
#pragma once

namespace {

constexpr int kMergeMaxRounds = 32;

// AMGH_VERBOSE: say which HIP call of the builder failed (the caller only sees -1001)
inline int merge_fail(const char* what) {
  if (getenv("AMGH_VERBOSE")) fprintf(stderr, "[amghip] merged-group builder: %s failed: %s\n", what, hipGetErrorString(hipGetLastError()));
  return -1001;
}

// A grouping of the dependency levels of one sweep: group q = levels [gb[q], gb[q+1]) (ascending level boundaries,
// gb[0] = 0, gb[ngrp] = nlev).  A forward sweep runs the groups in ascending order and the levels of a group in
// ascending order (round k of group q = level gb[q] + k); a backward sweep both in descending order (round k = level
// gb[q+1] - 1 - k).  Groups need not have the same depth: deep where the levels are small (the launch is at its
// latency floor whatever it carries), shallow where they are large (substitution inflates the rows).
struct MergeGrouping {
  std::vector<int32_t> gb;
  bool backward = false;
  int ngrp() const { return (int)gb.size() - 1; }
  int depth() const { int d = 0; for (int q = 0; q + 1 < (int)gb.size(); ++q) d = std::max(d, gb[q + 1] - gb[q]); return d; }
  static MergeGrouping uniform(int nlev, int m, bool backward) {   // groups of m levels counted from the start of the sweep
    MergeGrouping G;
    G.backward = backward;
    const int ngrp = nlev ? (nlev + m - 1) / m : 0;
    G.gb.assign(ngrp + 1, 0);
    for (int q = 0; q <= ngrp; ++q)
      G.gb[q] = backward ? (q == 0 ? 0 : std::max(0, nlev - (ngrp - q) * m)) : std::min(nlev, q * m);
    return G;
  }
};

struct MergeDev {
  int m = 1, ngrp = 0;                // m = deepest group
  bool backward = false;
  MergeGrouping grouping;
  int32_t* d_grp_of_lev = nullptr;    // per dependency level: its group / its round inside the group (device)
  int32_t* d_rnd_of_lev = nullptr;
  int32_t* clen = nullptr;            // per level-ordered row: entries of its composite row
  int32_t* coff = nullptr;            // ... offset inside its round's pool
  int32_t* rcol[kMergeMaxRounds] = {};
  real* rval[kMergeMaxRounds] = {};
  int64_t total = 0, max_row = 0;
  int64_t sampled_rows = 0;           // rows whose composite row was built (= n unless the build sampled the groups)
  int sample_stride = 1;              // groups q with q % stride == stride / 2 were built (1: all)
  double growth = 0.0;
  bool failed = false;                // a row outgrew the long-row table (the caller gives up on this m)
  std::vector<int32_t> h_clen;        // host copy of clen
  void free_dev() {
    hipFree(clen); hipFree(coff); hipFree(d_grp_of_lev); hipFree(d_rnd_of_lev);
    clen = coff = nullptr; d_grp_of_lev = d_rnd_of_lev = nullptr;
    for (int k = 0; k < kMergeMaxRounds; ++k) { hipFree(rcol[k]); hipFree(rval[k]); rcol[k] = nullptr; rval[k] = nullptr; }
  }
};

struct MergeArgs {
  const int32_t* prow; const int32_t* pcol; const real* pval; const real* pdiag; const int32_t* lev_of;
  int n, ncols, nlev, ngrp, backward, round;
  const int32_t* grp_of_lev; const int32_t* rnd_of_lev;   // per dependency level
  const int32_t* rlev;                                    // per group: its level of the current round, or -1
  const int32_t* clen; const int32_t* coff;
  const int32_t* rcol[kMergeMaxRounds]; const real* rval[kMergeMaxRounds];
  // count pass: cnt / ovf out;  fill pass: off in, this round's pool + clen / coff out
  int32_t* cnt; unsigned char* ovf; int32_t* any_ovf;
  const int32_t* off; int32_t* out_col; real* out_val; int32_t* clen_out; int32_t* coff_out;
  unsigned long long* growth;  // max over rows of sum |coefficient| / |diagonal| (bit pattern of a non-negative real)
  int32_t* fail;
  int fill;       // 0 count, 1 fill
  int tier;       // 0: 128-slot tables (all rows), 1: 512 slots, 2: a workgroup per row with 4096 slots — rows flagged by the tier before
  // workgroup -> rows: only the levels of this round are launched.  blk_ptr[q] = first workgroup of group q's level
  // of the round (ROWS rows per workgroup), lvl_ptr = rows of every dependency level
  const int32_t* blk_ptr; const int32_t* lvl_ptr;
};

__device__ __forceinline__ int merge_group_of(int l, const MergeArgs& a) { return a.grp_of_lev[l]; }
__device__ __forceinline__ int merge_round_of(int l, const MergeArgs& a) { return a.rnd_of_lev[l]; }

// LANES lanes per row, ROWS rows per workgroup, CAP table slots per row.  Dynamic LDS: per row CAP x (key, value,
// dense key, dense value).
template <int LANES, int ROWS, int CAP>
__global__ __launch_bounds__(LANES * ROWS) void merge_rows_kernel(MergeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  real* s_valb = (real*)smem;                              // ROWS * CAP
  real* s_dvalb = s_valb + ROWS * CAP;                       // ROWS * CAP
  int32_t* s_keyb = (int32_t*)(s_dvalb + ROWS * CAP);          // ROWS * CAP
  int32_t* s_dkeyb = s_keyb + ROWS * CAP;                      // ROWS * CAP
  __shared__ int32_t s_cnt[ROWS], s_pos[ROWS], s_steps;
  __shared__ double s_sum[ROWS];
  const int grp = threadIdx.x / LANES, ln = threadIdx.x % LANES;
  // this workgroup's group q (binary search in blk_ptr), the level of the round inside it, and its rows
  int p = a.n;
  {
    int lo = 0, hi = a.ngrp;  // largest q with blk_ptr[q] <= blockIdx.x
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.blk_ptr[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const int l = a.rlev[lo];
    if (l >= 0 && l < a.nlev) {
      const int r = a.lvl_ptr[l] + ((int)blockIdx.x - a.blk_ptr[lo]) * ROWS + grp;
      if (r < a.lvl_ptr[l + 1]) p = r;
    }
  }
  int32_t* key = s_keyb + grp * CAP;
  real* val = s_valb + grp * CAP;
  int32_t* dkey = s_dkeyb + grp * CAP;
  real* dval = s_dvalb + grp * CAP;
  bool live = p < a.n;
  int lp = 0;
  if (live) {
    lp = a.lev_of[p];
    live = merge_round_of(lp, a) == a.round;
    if (live) live = a.ovf[p] == (unsigned char)a.tier;  // tier 0: every row; tier t: rows that outgrew tier t-1's table
  }
  if (!__syncthreads_or(live ? 1 : 0)) return;  // no row of this workgroup belongs to the round (rows of a level are contiguous)
  for (int t = ln; t < CAP; t += LANES) key[t] = -1;
  if (ln == 0) { s_cnt[grp] = 0; s_pos[grp] = 0; s_sum[grp] = 0.0; }
  if (threadIdx.x == 0) s_steps = 0;
  __syncthreads();
  const int32_t j0 = live ? a.prow[p] : 0, j1 = live ? a.prow[p + 1] : 0;
  if (ln == 0 && j1 > j0) atomicMax(&s_steps, j1 - j0);
  __syncthreads();
  const int steps = s_steps;
  const int gp = live ? merge_group_of(lp, a) : 0;
  bool overflow = false;
  auto add = [&](int32_t c, real v) {
    uint32_t hsh = ((uint32_t)c * 2654435761u) & (CAP - 1);
    int probes = 0;
    for (;;) {
      const int32_t old = atomicCAS(&key[hsh], -1, c);
      if (old == -1) { atomicAdd(&s_cnt[grp], 1); if (a.fill) val[hsh] = v; return; }
      if (old == c) { if (a.fill) val[hsh] += v; return; }
      hsh = (hsh + 1) & (CAP - 1);
      if (++probes >= CAP) { overflow = true; return; }
    }
  };
  for (int step = 0; step < steps; ++step) {
    const int32_t j = j0 + step;
    if (j < j1 && s_cnt[grp] <= CAP - CAP / 8) {
      const int32_t c = a.pcol[j];
      if (c != p && c < a.n) {  // diagonal; halo entries belong to the pre-pass
        const int lc = a.lev_of[c];
        const bool tri = a.backward ? lc > lp : lc < lp;
        if (tri) {              // the other triangle belongs to the pre-pass
          const real v = a.pval[j];
          const real dc = a.pdiag[c];
          if (merge_group_of(lc, a) == gp && dc != 0.0) {  // same group: substitute row c's formula
            const real f = v / dc;
            if (ln == 0) add(a.ncols + c, f);
            const int rc = merge_round_of(lc, a);
            const int32_t o = a.coff[c], len = a.clen[c];
            const int32_t* cc = a.rcol[rc] + o;
            const real* cv = a.rval[rc] + o;
            for (int32_t e = ln; e < len; e += LANES) add(cc[e], a.fill ? -f * cv[e] : 0.0);
          } else if (ln == 0) {
            add(c, v);          // an earlier group (final), or a row that keeps its x (zero diagonal)
          }
        }
      }
    }
    __syncthreads();
  }
  const int32_t cn = s_cnt[grp];
  const int any_over = __syncthreads_or((overflow || cn > CAP - CAP / 8) ? 1 : 0);
  const bool too_full = any_over != 0;
  // (the vote above is workgroup-wide: with several rows per workgroup a neighbour's overflow re-does this row in the
  //  long-row pass too — harmless, it recomputes the same row)
  if (!live) return;
  if (too_full) {
    if (CAP >= 4096) { if (ln == 0) *a.fail = 1; return; }
    if (ln == 0) { a.ovf[p] = (unsigned char)(a.tier + 1); *a.any_ovf = 1; if (!a.fill) a.cnt[p] = 0; }
    return;
  }
  if (!a.fill) { if (ln == 0) a.cnt[p] = cn; return; }
  // sorted output: table -> dense list -> rank of every column
  for (int t = ln; t < CAP; t += LANES)
    if (key[t] != -1) {
      const int32_t o = atomicAdd(&s_pos[grp], 1);
      dkey[o] = key[t];
      dval[o] = val[t];
    }
  __syncthreads();
  const int32_t base = a.off[p];
  double part = 0.0;
  for (int32_t e = ln; e < cn; e += LANES) {
    const int32_t ke = dkey[e];
    int32_t rank = 0;
    for (int32_t f = 0; f < cn; ++f) rank += (dkey[f] < ke);
    a.out_col[base + rank] = ke;
    a.out_val[base + rank] = dval[e];
    part += fabs((double)dval[e]);
  }
  atomicAdd(&s_sum[grp], part);  // (order of this sum does not matter: growth is a coarse guard, 1e4)
  __syncthreads();
  if (ln == 0) {
    a.clen_out[p] = cn;
    a.coff_out[p] = base;
    const real d = a.pdiag[p];
    if (d != 0.0) {
      const double gr = s_sum[grp] / fabs((double)d);
      atomicMax(a.growth, (unsigned long long)__double_as_longlong(gr));
    }
  }
}

// all composite rows of grouping (m, backward) of schedule g, on the device.  dscale: SOR solves with the diagonal
// D / omega (dscale = 1 / omega), Gauss-Seidel with D (dscale = 1).
// min_tier: the smallest table worth trying (0: 128 slots, 1: 512, 2: 4096 — the caller knows the longest row of the
// previous, shallower grouping; rows only grow with m)
// sample_stride > 1: a COST-MODEL build — only every sample_stride-th group is built (entries per row, longest row and
// coefficient growth of a grouping are bulk properties; the candidate group sizes that are thrown away need not be
// built in full); total / max_row / growth then describe the sampled rows (sampled_rows of them).
int merge_build_dev(const GsSchedule* g, const int32_t* d_lev_of, const real* d_diag, const MergeGrouping& G, MergeDev* out,
                    int min_tier = 0, int sample_stride = 1) {
  dbg_pending("entry of the merged-group builder");
  const int64_t n = g->n;
  MergeDev& R = *out;
  const bool backward = G.backward;
  const int m = G.depth();
  R.m = m; R.backward = backward; R.grouping = G; R.sample_stride = std::max(1, sample_stride);
  R.ngrp = G.ngrp();
  if (m > kMergeMaxRounds) return AMGH_EUNSUPPORTED;
  if (R.ngrp < 0 || (g->nlev > 0 && (G.gb.front() != 0 || G.gb.back() != g->nlev))) return AMGH_EINVAL;
  {
    std::vector<int32_t> gol(std::max(g->nlev, 1), 0), rol(std::max(g->nlev, 1), 0);
    for (int q = 0; q < R.ngrp; ++q)
      for (int l = G.gb[q]; l < G.gb[q + 1]; ++l) { gol[l] = q; rol[l] = backward ? G.gb[q + 1] - 1 - l : l - G.gb[q]; }
    RC_TRY(dev_upload(&R.d_grp_of_lev, gol.data(), (int64_t)gol.size()));
    RC_TRY(dev_upload(&R.d_rnd_of_lev, rol.data(), (int64_t)rol.size()));
  }
  RC_TRY(dev_alloc(&R.clen, n));
  RC_TRY(dev_alloc(&R.coff, n));
  HIP_TRY(hipMemsetAsync(R.clen, 0, sizeof(int32_t) * n, nullptr));
  int32_t *cnt = nullptr, *off = nullptr, *flags = nullptr;
  unsigned char* ovf = nullptr;
  unsigned long long* growth = nullptr;
  int rc = dev_alloc(&cnt, n + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&off, n + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&flags, 2);
  if (rc == AMGH_OK) rc = dev_alloc(&ovf, n);
  if (rc == AMGH_OK) rc = dev_alloc(&growth, 1);
  if (rc == AMGH_OK && hipMemsetAsync(growth, 0, 8, nullptr) != hipSuccess) rc = -1001;
  constexpr int LS = 16, RS = 4, CT = 128, CS = 512;  // 16 lanes per row, small / medium table
  constexpr int LB = 256, RB = 1, CB = 4096;          // a workgroup per row
  const size_t lds_t = (size_t)RS * CT * 24, lds_s = (size_t)RS * CS * 24, lds_b = (size_t)RB * CB * 24;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)merge_rows_kernel<LB, RB, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    (void)hipFuncSetAttribute((const void*)merge_rows_kernel<LS, RS, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
    attr_set = true;
  }
  MergeArgs a{};
  a.prow = g->rowptr; a.pcol = g->col; a.pval = g->val; a.pdiag = d_diag; a.lev_of = d_lev_of;
  a.n = (int)n; a.ncols = (int)g->ncols; a.nlev = g->nlev; a.ngrp = R.ngrp; a.backward = backward ? 1 : 0;
  a.grp_of_lev = R.d_grp_of_lev; a.rnd_of_lev = R.d_rnd_of_lev;
  a.clen = R.clen; a.coff = R.coff; a.cnt = cnt; a.ovf = ovf; a.any_ovf = flags; a.fail = flags + 1;
  a.tier = 0;
  a.off = off; a.clen_out = R.clen; a.coff_out = R.coff; a.growth = growth;
  a.lvl_ptr = g->d_lvl_ptr;
  int32_t *blk_s = nullptr, *blk_b = nullptr, *rlev = nullptr;
  if (rc == AMGH_OK) rc = dev_alloc(&blk_s, R.ngrp + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&blk_b, R.ngrp + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&rlev, R.ngrp + 1);
  a.rlev = rlev;
  std::vector<int32_t> hs(R.ngrp + 1), hb(R.ngrp + 1), hl(R.ngrp + 1, -1);
  for (int k = 0; k < m && rc == AMGH_OK; ++k) {
    for (int q = 0; q < kMergeMaxRounds; ++q) { a.rcol[q] = R.rcol[q]; a.rval[q] = R.rval[q]; }
    a.round = k;
    // only the levels of this round are launched: workgroups per group for both launch shapes
    hs[0] = hb[0] = 0;
    for (int q = 0; q < R.ngrp; ++q) {
      int l = k < G.gb[q + 1] - G.gb[q] ? (backward ? G.gb[q + 1] - 1 - k : G.gb[q] + k) : -1;
      if (sample_stride > 1 && q % sample_stride != sample_stride / 2) l = -1;   // not in the sample
      hl[q] = l;
      const int w = (l >= 0 && l < g->nlev) ? g->lvl_ptr[l + 1] - g->lvl_ptr[l] : 0;
      hs[q + 1] = hs[q] + (w + RS - 1) / RS;
      hb[q + 1] = hb[q] + w;
    }
    const unsigned grid_s = (unsigned)hs[R.ngrp], grid_b = (unsigned)hb[R.ngrp];
    if (grid_s == 0) continue;
    if (hipMemcpy(rlev, hl.data(), sizeof(int32_t) * (R.ngrp + 1), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(blk_s, hs.data(), sizeof(int32_t) * (R.ngrp + 1), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(blk_b, hb.data(), sizeof(int32_t) * (R.ngrp + 1), hipMemcpyHostToDevice) != hipSuccess) { rc = merge_fail("upload of the round's launch tables"); break; }
    if (hipMemsetAsync(cnt, 0, sizeof(int32_t) * (n + 1), nullptr) != hipSuccess ||
        hipMemsetAsync(ovf, min_tier, n, nullptr) != hipSuccess || hipMemsetAsync(flags, 0, 8, nullptr) != hipSuccess) { rc = merge_fail("reset of the round's counters"); break; }
    // three table sizes: most rows are short (128 slots, 13 waves per CU); rows that outgrow a table are flagged and
    // redone by the next tier
    auto run_tiers = [&](int fill, bool* any1, bool* any2) -> int {
      a.fill = fill;
      int32_t hf[2] = {0, 0};
      a.tier = 0; a.blk_ptr = blk_s;
      if (min_tier == 0) {
        hipLaunchKernelGGL((merge_rows_kernel<LS, RS, CT>), dim3(grid_s), dim3(LS * RS), lds_t, nullptr, a);
        if (!fill) {
          if (hipMemcpy(hf, flags, 8, hipMemcpyDeviceToHost) != hipSuccess) return merge_fail("tier-0 launch / flags");
          *any1 = hf[0] != 0;
        }
      } else if (!fill) {
        *any1 = true;
        *any2 = min_tier >= 2;
      }
      if (*any1) {
        if (!fill && hipMemsetAsync(flags, 0, 4, nullptr) != hipSuccess) return -1001;
        a.tier = 1;
        if (min_tier <= 1) {
          hipLaunchKernelGGL((merge_rows_kernel<LS, RS, CS>), dim3(grid_s), dim3(LS * RS), lds_s, nullptr, a);
          if (!fill) {
            if (hipMemcpy(hf, flags, 8, hipMemcpyDeviceToHost) != hipSuccess) return merge_fail("tier-1 launch / flags");
            *any2 = hf[0] != 0;
          }
        }
        if (*any2) {
          a.tier = 2; a.blk_ptr = blk_b;
          hipLaunchKernelGGL((merge_rows_kernel<LB, RB, CB>), dim3(grid_b), dim3(LB * RB), lds_b, nullptr, a);
        }
      }
      if (hipMemcpy(hf, flags, 8, hipMemcpyDeviceToHost) != hipSuccess) return merge_fail("tier-2 launch / flags");
      if (hf[1]) R.failed = true;
      return AMGH_OK;
    };
    bool any1 = false, any2 = false;
    rc = run_tiers(0, &any1, &any2);
    if (rc == AMGH_OK) {
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        if (getenv("AMGH_VERBOSE"))
          fprintf(stderr, "[amghip] merged-group builder: count pass of round %d: %s (grids %u / %u, tiers %d %d)\n", k, hipGetErrorString(e),
                  grid_s, grid_b, (int)any1, (int)any2);
        rc = -1001;
      }
    }
    if (rc != AMGH_OK || R.failed) break;
    int64_t total = 0;
    rc = dev_exclusive_scan(cnt, off, n, &total, nullptr);
    if (rc == AMGH_EUNSUPPORTED) { R.failed = true; rc = AMGH_OK; break; }
    if (rc != AMGH_OK) break;
    if (R.total + total >= (int64_t)INT32_MAX - 4096) { R.failed = true; break; }
    rc = dev_alloc(&R.rcol[k], total);
    if (rc == AMGH_OK) rc = dev_alloc(&R.rval[k], total);
    if (rc != AMGH_OK) break;
    R.total += total;
    a.out_col = R.rcol[k]; a.out_val = R.rval[k];
    // the fill passes flag the same rows again (same tables): the flags of the count pass stay valid, ovf is reset so
    // that tier 0 sees every row
    if (hipMemsetAsync(ovf, min_tier, n, nullptr) != hipSuccess) { rc = -1001; break; }
    rc = run_tiers(1, &any1, &any2);
    if (rc != AMGH_OK || R.failed) break;
    {
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        if (getenv("AMGH_VERBOSE"))
          fprintf(stderr, "[amghip] merged-group builder: fill pass of round %d: %s (grids %u / %u, total %lld, tiers %d %d)\n", k,
                  hipGetErrorString(e), grid_s, grid_b, (long long)total, (int)any1, (int)any2);
        rc = -1001;
      }
    }
  }
  if (rc == AMGH_OK && !R.failed) {
    R.h_clen.resize(n);
    unsigned long long gbits = 0;
    if (hipMemcpy(R.h_clen.data(), R.clen, sizeof(int32_t) * n, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&gbits, growth, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -1001;
    else {
      std::memcpy(&R.growth, &gbits, 8);
      for (int64_t p2 = 0; p2 < n; ++p2) R.max_row = std::max<int64_t>(R.max_row, R.h_clen[p2]);
      R.sampled_rows = n;
      if (sample_stride > 1) {
        R.sampled_rows = 0;
        for (int q = sample_stride / 2; q < R.ngrp; q += sample_stride) R.sampled_rows += g->lvl_ptr[G.gb[q + 1]] - g->lvl_ptr[G.gb[q]];
      }
    }
  }
  hipFree(cnt); hipFree(off); hipFree(flags); hipFree(ovf); hipFree(growth); hipFree(blk_s); hipFree(blk_b); hipFree(rlev);
  if (rc != AMGH_OK) R.free_dev();
  return rc;
}

// the composite rows as ONE contiguous CSR on the device (rows in level order): gather from the round pools
__global__ void merge_gather_kernel(MergeArgs a, const int32_t* prow_new, int32_t* col, real* val) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n) return;
  const int len = a.clen[p];
  if (len <= 0) return;
  const int rc = merge_round_of(a.lev_of[p], a);
  const int32_t* cc = a.rcol[rc] + a.coff[p];
  const real* cv = a.rval[rc] + a.coff[p];
  const int32_t dst = prow_new[p];
  for (int e = 0; e < len; ++e) { col[dst + e] = cc[e]; val[dst + e] = cv[e]; }
}

}  // namespace
