// gs_blocks.hpp — exact lexicographic Gauss-Seidel as a WAVEFRONT OF BLOCKS (no substitution, no recomputation).
// Included by gs_schedule.hpp (the schedule of single-right-hand-side hierarchies on stencil-like fine levels) and by
// tools/block_wave_bench.hip (the stand-alone measurement, profiles/r03_block_wave.log).
//
// smoother.jl:61-90 sweeps the rows in index order; any order that respects the dependency DAG (row i after every
// adjacent row c < i, before every adjacent row c > i) gives the same iterate.  The level schedules of gs_schedule.hpp
// pay a kernel boundary per dependency level (or per group of levels, after substitution).  Here the rows are
// partitioned into BLOCKS whose quotient graph is acyclic; blocks at the same depth of the quotient DAG are independent
// and form one launch; inside a block ONE wave walks the block's own dependency levels with the block's x in LDS —
// LDS operations of one wave execute in program order, so there is no barrier at all between the levels.  Every row is
// summed by one lane in the stored entry order with separately rounded products and divided by its diagonal: the
// iterate is the scalar loop's bit for bit, forward and backward from the same layout.
//
// Acyclic partition from MONOTONE POTENTIALS: a potential phi is any row function with phi(c) <= phi(i) along every
// dependency edge c -> i.  Cells of the product quantisation (phi_1 / h_1, phi_2 / h_2, phi_3 / h_3) depend only on cells
// that are <= in every coordinate, so the quotient graph is acyclic whatever the matrix.  The potentials are longest
// paths with 0/1 edge weights: phi_k counts the edges of offset class k, the classes being the clusters of the index
// offsets i - c on a log scale (on a lexicographically ordered grid: the x / y / z neighbours, and the skewed
// equivalents on the Ruge-Stueben coarse grids of such a grid).  Cells with more rows than a workgroup can hold are cut
// along the dependency level (one more monotone potential), a single oversized level into chunks of independent rows.
// The construction is always valid; whether it is USED is a cost-model decision (launches, depth, bytes).
//
// Execution: ONE launch per sweep (gs_bw_chain_kernel, the default for single-column sweeps) — workgroups draw tickets in
// the order of the quotient DAG's depths and a block waits for the flags of the few blocks it depends on instead of a
// kernel boundary, its record load running ahead of that chain —, or one launch per depth (gs_bw_packed_kernel: blocks of
// right-hand sides, tunable gs_bw_chain = 0).  Same records, same arithmetic, the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <thread>
#include <utility>
#include <vector>

namespace amgh {
namespace bw {

struct Desc {          // one block (32 bytes)
  int32_t row0, nrows; // rows of the block in block order (sorted by in-block dependency level)
  int32_t rec;         // the block's record in the record buffer, in units of 16 bytes
  int32_t npre;        // external columns that lie BEFORE the block in block order (the first npre of the sorted list): written by predecessor blocks
  int32_t ext0, next;  // external columns: positions of x gathered into LDS behind the block's own x
  int32_t nlev, maxk;  // STEPS of the block: pieces of <= 64 rows of one in-block dependency level, in order; off-diagonal entries per row (padded, a multiple of kChunk)
};
constexpr int kChunk = 6;       // entries summed per batch of LDS reads
constexpr int kThreads = 512;   // loading workgroup (the sweep itself is wave 0)
constexpr int kMaxSteps = 124;
#ifndef BW_PLAN_MAXK
#define BW_PLAN_MAXK 18
#endif
#ifndef BW_POLL_SLEEP
#define BW_POLL_SLEEP 8   // x 64 clocks between two polls of a flag (256^3 fine level, ms per sweep: 0: 0.932, 1: 0.920, 2: 0.921, 8: 0.910, 16: 0.914, 32: 0.924; tools/block_wave_bench -DBW_POLL_SLEEP=...)
#endif
constexpr int kPlanMaxK = BW_PLAN_MAXK;   // longest rows (off-diagonal entries, padded) a plan accepts = the kernels instantiated below  // steps per block (their row pointers live in two registers of wave 0)

// PACKED rows (the format of gs_bw_packed_kernel): a row is one run of 16-byte chunks
//     [ v0 v1 | v2 v3 | ... | (.. dg rc) | c0 .. c7 | c8 .. ]      values, diagonal, its reciprocal, then the columns as uint16
// BYTE offsets into the LDS x — so that a step loads a row's operands with a handful of ds_read_b128 off ONE address
// (a single wave is bound by its instruction count: ~3 ns per instruction, whatever it is).  The chunk count is odd: rows
// 16 k bytes apart with k odd are read by 64 lanes without bank conflicts.
template <typename R>
struct Packed {
  static constexpr int vpc = 16 / (int)sizeof(R);                       // values per chunk
  static int nvc(int maxk) { return (maxk + 2 + vpc - 1) / vpc; }       // value chunks: maxk values + diagonal + reciprocal
  static int ncc(int maxk) { return (maxk + 7) / 8; }                   // column chunks
  static int chunks(int maxk) { const int k = nvc(maxk) + ncc(maxk); return k | 1; }
  static size_t row_bytes(int maxk) { return (size_t)16 * chunks(maxk); }
  static size_t rec_bytes(int nrows, int maxk, int nsteps) { return ((size_t)nrows * row_bytes(maxk) + (size_t)(nsteps + 1) * 2 + 15) & ~(size_t)15; }
  // LDS: x (nrows + next + 1, padded to 16 bytes) | b (nrows, padded) | the record
  static size_t lds_bytes(const Desc& d) {
    const size_t nx = ((size_t)(d.nrows + d.next + 1) * sizeof(R) + 15) & ~(size_t)15, nb = ((size_t)d.nrows * sizeof(R) + 15) & ~(size_t)15;
    return nx + nb + rec_bytes(d.nrows, d.maxk, d.nlev);
  }
};

// (a byte buffer whose resize does not touch the pages: the records — gigabytes — are zeroed and filled block by block
// by the threads that build them instead of twice, first by one thread)
template <typename T>
struct NoInit {
  typedef T value_type;
  NoInit() = default;
  template <class U> NoInit(const NoInit<U>&) {}
  T* allocate(size_t k) { return static_cast<T*>(::operator new(k * sizeof(T))); }
  void deallocate(T* p, size_t) { ::operator delete(p); }
  template <class U, class... A> void construct(U* p, A&&... a) { if (sizeof...(A)) ::new ((void*)p) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const NoInit<U>&) const { return true; }
  template <class U> bool operator!=(const NoInit<U>&) const { return false; }
};
struct Plan {
  int64_t n = 0, nnz = 0;
  std::vector<int32_t> perm;        // block-order position -> natural row
  std::vector<Desc> blocks;         // in launch order
  std::vector<int32_t> launch_ptr;  // launch l = blocks [launch_ptr[l], launch_ptr[l + 1])
  std::vector<int64_t> launch_rec;  // byte offset of launch l's first record (launch_rec[nlaunch] = all records)
  std::vector<int32_t> ext_col;     // block-order positions
  std::vector<unsigned char, NoInit<unsigned char>> rec;   // the records, one after the other
  std::vector<int32_t> dep_ptr, dep;    // quotient graph on ordered blocks: the blocks a block waits for in the forward sweep ...
  std::vector<int32_t> sdep_ptr, sdep;  // ... and in the backward sweep (its successors); the chained kernel (one launch per sweep)
  size_t lds_max = 0;               // largest dynamic LDS request of a block
  // what the cost model looks at
  int nlevels = 0;                  // dependency levels of the operator
  int64_t sum_depth = 0;            // sum over launches of the deepest block's level count
  int64_t ext_total = 0;
  int max_rows = 0;
  double est_seconds = 0.0;         // modelled time of one directional sweep, one launch per depth ...
  double est_chain_seconds = 0.0;   // ... and as one launch, blocks chained by flags
  double est_flow_seconds = 0.0;    // ... and as a dataflow with relayed walks (gs_flow.hpp / gs_relay.hpp)
  bool late_ok = false;             // every row's entries lie split around the padding with at most maxk / 2 on either side (the records' fill)
  int cuts[2] = {99, 99};
  int32_t range[3] = {1, 1, 1};
  int32_t cells[3] = {1, 1, 1};
};

// (an exception in a worker — the plans allocate gigabytes — is carried to the calling thread, which lets the caller fall
// back to another schedule instead of std::terminate taking the process down)
template <class F>
inline void parallel_for(int T, F fn) {
  if (T <= 1) { fn(0, 1); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)T);
  try {
    for (int t = 1; t < T; ++t)
      th.emplace_back([=, &err] { try { fn(t, T); } catch (...) { err[(size_t)t] = std::current_exception(); } });
  } catch (...) {   // (a thread could not be created: the ones that run are joined before the exception travels on — a
    for (auto& x : th) x.join();   // joinable std::thread destroyed would be std::terminate)
    throw;
  }
  try { fn(0, T); } catch (...) { err[0] = std::current_exception(); }
  for (auto& x : th) x.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
}

struct Params {
  int target_rows = 512;     // rows per block aimed at
  int max_rows = 1024;       // hard cap (uint16 local indices, LDS)
  size_t lds_limit = 150 * 1024;
  int threads = 8;
  bool require_three = false;   // give up right after the offset histogram unless it shows three classes (a caller that would
                                // reject such a plan anyway: saves the passes over the matrix)
  bool require_two = false;     // ... or at least two (operators large enough for a line-shaped wavefront to pay)
  bool flow_only = false;       // the plan will only be executed by the dataflow kernel (gs_flow.hpp): a block's LDS holds its x,
                                // not its record — the record's bytes do not bound the rows of a block (long rows: levels 2+)
};

// Returns false when the operator cannot be laid out (rows too long for the LDS budget, local index overflow).
template <typename R>
bool plan(int64_t n, const int32_t* rowptr, const int32_t* col, const R* val, const Params& prm, Plan* out) {
  Plan& P = *out;
  P = Plan();
  P.n = n; P.nnz = rowptr[n];
  if (n <= 0) return false;
  const bool timing = getenv("BW_PLAN_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[bw plan] %-34s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  // ---- offset classes ----
  int64_t hist[32] = {0};
  int maxlen = 0;
  int64_t cmax = n - 1;   // largest column (halo columns of a row-sharded operator lie behind the rows)
  {
    const int TH = std::max(1, prm.threads);
    std::vector<std::array<int64_t, 34>> part((size_t)TH);   // per thread: the histogram, the longest row, the largest column
    parallel_for(TH, [&](int t, int TT) {
      std::array<int64_t, 34>& h = part[(size_t)t];
      h.fill(0); h[33] = n - 1;
      for (int64_t i = n * t / TT; i < n * (t + 1) / TT; ++i) {
        int off = 0;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
          h[33] = std::max<int64_t>(h[33], col[j]);
          const int64_t d = (int64_t)i - col[j];
          if (d != 0) ++off;                                   // (halo columns of a row-sharded operator count as entries of the row)
          if (d != 0 && col[j] < n) h[31 - __builtin_clz((unsigned)(d < 0 ? -d : d))]++;
        }
        h[32] = std::max<int64_t>(h[32], off);   // off-diagonal entries of the longest row
      }
    });
    for (const auto& h : part) {
      for (int b = 0; b < 32; ++b) hist[b] += h[b];
      maxlen = std::max(maxlen, (int)h[32]); cmax = std::max(cmax, h[33]);
    }
  }
  const int maxk = ((std::max(1, maxlen) + kChunk - 1) / kChunk) * kChunk;
  if (maxk > kPlanMaxK) return false;   // (kMaxK: the kernels' register sets; longer rows make the walk instruction-bound, profiles/r03_block_wave.log)
  {
    std::vector<int> occ;
    for (int b = 0; b < 32; ++b) if (hist[b]) occ.push_back(b);
    std::vector<std::pair<int, int>> gaps;
    for (size_t k = 0; k + 1 < occ.size(); ++k) gaps.push_back({occ[k + 1] - occ[k], occ[k]});
    std::stable_sort(gaps.begin(), gaps.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
    for (int g = 0; g < 2 && g < (int)gaps.size(); ++g) if (gaps[g].first >= 2) P.cuts[g] = gaps[g].second;
    if (P.cuts[0] > P.cuts[1]) std::swap(P.cuts[0], P.cuts[1]);
    if (const char* e = getenv("BW_PLAN_CUTS")) { int a = 99, b = 99; if (sscanf(e, "%d,%d", &a, &b) >= 1) { P.cuts[0] = a; P.cuts[1] = b; } }   // (measurement hook: the offset classes by hand)
  }
  if (prm.require_three && (P.cuts[0] == 99 || P.cuts[1] == 99)) return false;
  if (prm.require_two && P.cuts[0] == 99 && P.cuts[1] == 99) return false;   // (one class: a chain of slabs)
  lap("offset histogram");
  const int cut0 = P.cuts[0], cut1 = P.cuts[1];
  auto cls = [cut0, cut1](int64_t d) { const int b = 31 - __builtin_clz((unsigned)d); return b <= cut0 ? 0 : b <= cut1 ? 1 : 2; };
  // ---- potentials and dependency levels of the symmetrised pattern (one pass in index order) ----
  std::vector<int32_t> phi[3], lev(n, 0);
  for (int k = 0; k < 3; ++k) phi[k].assign(n, 0);
  for (int64_t i = 0; i < n; ++i) {
    int32_t p0 = phi[0][i], p1 = phi[1][i], p2 = phi[2][i], l = lev[i];
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      const int32_t c = col[j];
      if (c >= i) continue;
      const int k = cls(i - c);
      p0 = std::max(p0, phi[0][c] + (k == 0)); p1 = std::max(p1, phi[1][c] + (k == 1)); p2 = std::max(p2, phi[2][c] + (k == 2));
      l = std::max(l, lev[c] + 1);
    }
    phi[0][i] = p0; phi[1][i] = p1; phi[2][i] = p2; lev[i] = l;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {   // an upper entry of row i is a dependency i -> c as well
      const int32_t c = col[j];
      if (c <= i || c >= n) continue;
      const int k = cls(c - i);
      phi[0][c] = std::max(phi[0][c], p0 + (k == 0)); phi[1][c] = std::max(phi[1][c], p1 + (k == 1)); phi[2][c] = std::max(phi[2][c], p2 + (k == 2));
      lev[c] = std::max(lev[c], l + 1);
    }
  }
  for (int k = 0; k < 3; ++k) P.range[k] = *std::max_element(phi[k].begin(), phi[k].end()) + 1;
  lap("potentials + levels");
  P.nlevels = *std::max_element(lev.begin(), lev.end()) + 1;
  // ---- rows per block from the LDS budget ----
  const size_t per_row = (prm.flow_only ? 0 : Packed<R>::row_bytes(maxk)) + 2 * sizeof(R) + 2 * sizeof(R);   // packed row + b + x + ~2 external values
  int cap = (int)std::min<size_t>(prm.max_rows, prm.lds_limit / per_row);
  cap = (cap / 64) * 64;
  if (cap < 64) return false;
  const int target = std::min(prm.target_rows, cap);
  cap = std::min(cap, std::max(64, ((target * 5 / 4 + 63) / 64) * 64));   // blocks stay near the size asked for
  // ---- cell widths: the same number of cells m along every (non-degenerate) potential, m raised until few rows sit in oversized cells ----
  double m = std::cbrt((double)n / target);
  std::vector<int32_t> cell(n);
  int32_t h[3], nb[3];
  std::vector<int32_t> count;
  for (int it = 0; it < 24; ++it) {
    int64_t ncell = 1;
    int32_t ht[3], nt[3];
    for (int k = 0; k < 3; ++k) {
      ht[k] = std::max<int32_t>(1, (int32_t)std::floor(P.range[k] / m + 0.5));
      nt[k] = (P.range[k] + ht[k] - 1) / ht[k];
      ncell *= nt[k];
    }
    if (ncell > ((int64_t)1 << 28)) { if (it == 0) return false; break; }
    for (int k = 0; k < 3; ++k) { h[k] = ht[k]; nb[k] = nt[k]; }
    count.assign((size_t)ncell, 0);
    {
      const int TH = std::max(1, prm.threads);
      std::vector<std::vector<int32_t>> pc((size_t)TH);
      parallel_for(TH, [&](int t, int TT) {
        std::vector<int32_t>& c = pc[(size_t)t];
        c.assign((size_t)ncell, 0);
        for (int64_t i = n * t / TT; i < n * (t + 1) / TT; ++i) {
          const int32_t q = (phi[0][i] / h[0]) + nb[0] * ((phi[1][i] / h[1]) + nb[1] * (phi[2][i] / h[2]));
          cell[i] = q;
          c[q]++;
        }
      });
      for (const auto& c : pc) for (int64_t q = 0; q < ncell; ++q) count[(size_t)q] += c[(size_t)q];
    }
    int64_t over = 0;
    for (int32_t cnt : count) if (cnt > cap) over += cnt;
    if (over * 10 <= n || (h[0] == 1 && h[1] == 1 && h[2] == 1)) break;
    m *= 1.12;
  }
  for (int k = 0; k < 3; ++k) P.cells[k] = nb[k];
  lap("cell widths");
  const int64_t ncell = (int64_t)nb[0] * nb[1] * nb[2];
  // ---- rows of every cell in (level, index) order; oversized cells cut along the level ----
  std::vector<int64_t> cptr(ncell + 1, 0);
  for (int64_t q = 0; q < ncell; ++q) cptr[q + 1] = cptr[q] + count[q];
  std::vector<int32_t> crow(n);
  {
    std::vector<int64_t> next(cptr.begin(), cptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) crow[next[cell[i]]++] = (int32_t)i;
  }
  const int T = std::max(1, prm.threads);
  parallel_for(T, [&](int t, int TT) {
    for (int64_t q = ncell * t / TT; q < ncell * (t + 1) / TT; ++q)
      std::stable_sort(crow.begin() + cptr[q], crow.begin() + cptr[q + 1], [&](int32_t a, int32_t b) { return lev[a] < lev[b]; });
  });
  lap("rows of cells, sorted");
  // blocks (pre-order: by cell, then by level window)
  std::vector<int32_t> blk(n);
  std::vector<int64_t> bptr;        // rows of pre-order block b: crow[bptr[b] .. bptr[b+1])
  std::vector<int32_t> bsum;        // coordinate sum of the block's cell (a topological key of the quotient graph)
  bptr.push_back(0);
  for (int64_t q = 0; q < ncell; ++q) {
    int64_t a = cptr[q];
    const int64_t e = cptr[q + 1];
    if (a == e) continue;
    const int32_t cs = (int32_t)(q % nb[0] + (q / nb[0]) % nb[1] + q / ((int64_t)nb[0] * nb[1]));
    while (a < e) {
      int64_t z = std::min<int64_t>(e, a + cap);
      if (z < e) {   // cut at a level boundary; a single level longer than the cap is cut into chunks of independent rows
        int64_t zz = z;
        while (zz > a && lev[crow[zz - 1]] == lev[crow[z]]) --zz;
        if (zz > a) z = zz;
        else { const int32_t l0 = lev[crow[a]]; z = a; while (z < e && z < a + cap && lev[crow[z]] == l0) ++z; }
      }
      for (int64_t r = a; r < z; ++r) blk[crow[r]] = (int32_t)(bptr.size() - 1);
      bptr.push_back(z);
      bsum.push_back(cs);
      a = z;
    }
  }
  const int32_t B = (int32_t)bsum.size();
  std::vector<int32_t>().swap(cell);
  for (int k = 0; k < 3; ++k) std::vector<int32_t>().swap(phi[k]);
  lap("blocks");
  // ---- launches: longest path in the quotient graph ----
  std::vector<int32_t> blev(B, 0);
  std::vector<int64_t> eptr(B + 1, 0);   // edges (block of c) -> (block of i) for c < i adjacent, collected per target block (pre-order)
  std::vector<int32_t> esrc;
  {
    // (counted and filled by all threads at once: the order of a block's list is whatever the threads make it — only its
    // maximum and, sorted and made unique below, its set are used)
    const int TH = std::max(1, prm.threads);
    parallel_for(TH, [&](int t, int TT) {
      for (int64_t i = n * t / TT; i < n * (t + 1) / TT; ++i)
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
          const int32_t c = col[j];
          if (c >= n || c == i || blk[c] == blk[i]) continue;
          __atomic_fetch_add(&eptr[(c < i ? blk[i] : blk[c]) + 1], (int64_t)1, __ATOMIC_RELAXED);
        }
    });
    for (int32_t b = 0; b < B; ++b) eptr[b + 1] += eptr[b];
    esrc.resize(eptr[B]);
    std::vector<int64_t> next(eptr.begin(), eptr.end() - 1);
    parallel_for(TH, [&](int t, int TT) {
      for (int64_t i = n * t / TT; i < n * (t + 1) / TT; ++i)
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
          const int32_t c = col[j];
          if (c >= n || c == i || blk[c] == blk[i]) continue;
          const int32_t tgt = c < i ? blk[i] : blk[c], src = c < i ? blk[c] : blk[i];
          esrc[__atomic_fetch_add(&next[tgt], (int64_t)1, __ATOMIC_RELAXED)] = src;
        }
    });
    // pre-order is a topological order: cells by coordinate sum would be one, and so is (cell index, window) — a cell
    // only depends on cells with smaller-or-equal coordinates, i.e. smaller cell index, and inside a cell on earlier windows
    for (int32_t b = 0; b < B; ++b) {
      int32_t l = 0;
      for (int64_t e = eptr[b]; e < eptr[b + 1]; ++e) l = std::max(l, blev[esrc[e]] + 1);
      blev[b] = l;
    }
  }
  const int nlaunch = *std::max_element(blev.begin(), blev.end()) + 1;
  lap("quotient edges + depths");
  // ---- block order: by launch, pre-order inside a launch ----
  std::vector<int32_t> order(B);
  P.launch_ptr.assign(nlaunch + 1, 0);
  for (int32_t b = 0; b < B; ++b) P.launch_ptr[blev[b] + 1]++;
  for (int l = 0; l < nlaunch; ++l) P.launch_ptr[l + 1] += P.launch_ptr[l];
  {
    std::vector<int32_t> next(P.launch_ptr.begin(), P.launch_ptr.end() - 1);
    for (int32_t b = 0; b < B; ++b) order[next[blev[b]]++] = b;
  }
  // ---- in-block dependency levels (index order is a topological order) ----
  std::vector<int32_t> ilev(n, 0);
  parallel_for(T, [&](int t, int TT) {   // (block by block: the recurrence only couples rows of one block)
    std::vector<int32_t> rows;
    for (int32_t b = (int32_t)(B * (int64_t)t / TT); b < (int32_t)(B * (int64_t)(t + 1) / TT); ++b) {
      rows.assign(crow.begin() + bptr[b], crow.begin() + bptr[b + 1]);
      std::sort(rows.begin(), rows.end());
      for (int32_t i : rows) {
        int32_t l = ilev[i];
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) { const int32_t c = col[j]; if (c < i && blk[c] == b) l = std::max(l, ilev[c] + 1); }
        ilev[i] = l;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) { const int32_t c = col[j]; if (c > i && c < n && blk[c] == b) ilev[c] = std::max(ilev[c], l + 1); }
      }
    }
  });
  lap("in-block levels");
  // ---- rows in block order ----
  P.blocks.resize(B);
  P.perm.resize(n);
  std::vector<int32_t> inv(n);
  {
    int64_t pos = 0;
    for (int32_t ob = 0; ob < B; ++ob) {
      const int32_t b = order[ob];
      Desc& d = P.blocks[ob];
      d.row0 = (int32_t)pos; d.nrows = (int32_t)(bptr[b + 1] - bptr[b]);
      for (int64_t r = bptr[b]; r < bptr[b + 1]; ++r) P.perm[pos++] = crow[r];
    }
  }
  parallel_for(T, [&](int t, int TT) {
    for (int32_t ob = B * (int64_t)t / TT; ob < B * (int64_t)(t + 1) / TT; ++ob) {
      const Desc& d = P.blocks[ob];
      std::stable_sort(P.perm.begin() + d.row0, P.perm.begin() + d.row0 + d.nrows, [&](int32_t a, int32_t b) { return ilev[a] != ilev[b] ? ilev[a] < ilev[b] : a < b; });
      for (int32_t p = 0; p < d.nrows; ++p) inv[P.perm[d.row0 + p]] = d.row0 + p;
    }
  });
  std::vector<int32_t> obk(B);   // pre-order block -> ordered block
  for (int32_t ob = 0; ob < B; ++ob) obk[order[ob]] = ob;
  lap("block order");
  // ---- the quotient graph on ordered blocks (distinct predecessors / successors of every block) ----
  {
    P.dep_ptr.assign(B + 1, 0);
    std::vector<std::vector<int32_t>> pre(B);
    parallel_for(T, [&](int t, int TT) {
      for (int32_t ob = B * (int64_t)t / TT; ob < B * (int64_t)(t + 1) / TT; ++ob) {
        const int32_t b = order[ob];
        std::vector<int32_t>& v = pre[ob];
        for (int64_t e = eptr[b]; e < eptr[b + 1]; ++e) v.push_back(obk[esrc[e]]);
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
      }
    });
    P.sdep_ptr.assign(B + 1, 0);
    for (int32_t ob = 0; ob < B; ++ob) { P.dep_ptr[ob + 1] = P.dep_ptr[ob] + (int32_t)pre[ob].size(); for (int32_t q : pre[ob]) P.sdep_ptr[q + 1]++; }
    for (int32_t ob = 0; ob < B; ++ob) P.sdep_ptr[ob + 1] += P.sdep_ptr[ob];
    P.dep.resize(P.dep_ptr[B]); P.sdep.resize(P.sdep_ptr[B]);
    std::vector<int32_t> nx(P.sdep_ptr.begin(), P.sdep_ptr.end() - 1);
    for (int32_t ob = 0; ob < B; ++ob) {
      std::copy(pre[ob].begin(), pre[ob].end(), P.dep.begin() + P.dep_ptr[ob]);
      for (int32_t q : pre[ob]) P.sdep[nx[q]++] = ob;
    }
    std::vector<int64_t>().swap(eptr); std::vector<int32_t>().swap(esrc);
  }
  lap("dependency lists");
  // ---- external columns, record offsets ----
  std::vector<int64_t> ext_ptr(B + 1, 0);
  std::vector<std::vector<int32_t>> exts(B);
  parallel_for(T, [&](int t, int TT) {
    std::vector<int32_t> seen((size_t)cmax + 1, -1);   // per thread: the block that listed a position last (each position once per block)
    for (int32_t ob = B * (int64_t)t / TT; ob < B * (int64_t)(t + 1) / TT; ++ob) {
      const Desc& d = P.blocks[ob];
      std::vector<int32_t>& ex = exts[ob];
      for (int32_t p = 0; p < d.nrows; ++p) {
        const int32_t i = P.perm[d.row0 + p];
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
          const int32_t c = col[j];
          if (c == i) continue;
          const int32_t q = c >= n ? c : inv[c];            // halo columns keep their place behind the rows
          if ((q < d.row0 || q >= d.row0 + d.nrows) && seen[q] != ob) { seen[q] = ob; ex.push_back(q); }
        }
      }
      std::sort(ex.begin(), ex.end());
    }
  });
  lap("external columns");
  size_t rec_total = 0;
  bool fits = true;
  for (int32_t ob = 0; ob < B; ++ob) {
    Desc& d = P.blocks[ob];
    d.npre = (int32_t)(std::lower_bound(exts[ob].begin(), exts[ob].end(), d.row0) - exts[ob].begin());
    d.maxk = maxk;
    int32_t nl = 0, run = 0, last = -1;   // steps: a level's rows in pieces of 64 (rows are sorted by level)
    for (int32_t p = 0; p < d.nrows; ++p) {
      const int32_t l = ilev[P.perm[d.row0 + p]];
      if (l != last || run == 64) { ++nl; run = 0; last = l; }
      ++run;
    }
    d.nlev = nl;
    if (nl > kMaxSteps) fits = false;
    d.ext0 = (int32_t)ext_ptr[ob]; d.next = (int32_t)exts[ob].size();
    ext_ptr[ob + 1] = ext_ptr[ob] + d.next;
    if (rec_total / 16 > (size_t)INT32_MAX) fits = false;
    d.rec = (int32_t)(rec_total / 16);
    rec_total += Packed<R>::rec_bytes(d.nrows, d.maxk, d.nlev);
    const size_t l = prm.flow_only ? (((size_t)(d.nrows + d.next + 1) * sizeof(R) + 15) & ~(size_t)15) + 16 : Packed<R>::lds_bytes(d);
    P.lds_max = std::max(P.lds_max, l);
    if (d.nrows + d.next + 1 > 65535 || d.nlev > 65534 || l > 160 * 1024 - 64) fits = false;   // (- 64: the chained kernel's static LDS)
    if ((size_t)(d.nrows + d.next + 1) * sizeof(R) > 65535) fits = false;   // byte offsets in 16 bits
    P.max_rows = std::max(P.max_rows, d.nrows);
  }
  if (!fits || ext_ptr[B] > INT32_MAX) return false;
  P.launch_rec.assign(nlaunch + 1, (int64_t)rec_total);
  for (int l = 0; l < nlaunch; ++l) P.launch_rec[l] = (int64_t)P.blocks[P.launch_ptr[l]].rec * 16;
  P.ext_total = ext_ptr[B];
  P.ext_col.resize(ext_ptr[B]);
  lap("descriptors");
  P.rec.resize(rec_total);
  std::atomic<int> late_bad{0};
  parallel_for(T, [&](int t, int TT) {
    std::vector<int32_t> where((size_t)cmax + 1);      // per thread: index of a position in the current block's external list
    for (int32_t ob = B * (int64_t)t / TT; ob < B * (int64_t)(t + 1) / TT; ++ob) {
      const Desc& d = P.blocks[ob];
      const std::vector<int32_t>& ex = exts[ob];
      std::copy(ex.begin(), ex.end(), P.ext_col.begin() + d.ext0);
      for (size_t e = 0; e < ex.size(); ++e) where[ex[e]] = (int32_t)e;
      unsigned char* rec = P.rec.data() + (size_t)d.rec * 16;
      std::memset(rec, 0, Packed<R>::rec_bytes(d.nrows, d.maxk, d.nlev));
      const size_t rs = Packed<R>::row_bytes(d.maxk);
      const int nvc = Packed<R>::nvc(d.maxk);
      const uint16_t zoff = (uint16_t)((size_t)(d.nrows + d.next) * sizeof(R));   // the LDS slot that holds 0
      uint16_t* stp = (uint16_t*)(rec + (size_t)d.nrows * rs);
      int32_t nl = 0, run = 0, last = -1;
      for (int32_t p = 0; p < d.nrows; ++p) {
        const int32_t i = P.perm[d.row0 + p];
        const int32_t l = ilev[i];
        if (l != last || run == 64) { stp[nl++] = (uint16_t)p; run = 0; last = l; }
        ++run;
        R* v = (R*)(rec + (size_t)p * rs);
        uint16_t* cc = (uint16_t*)(rec + (size_t)p * rs + (size_t)16 * nvc);
        for (int k = 0; k < 8 * Packed<R>::ncc(d.maxk); ++k) cc[k] = zoff;
        // Entries in stored order, the padding in the MIDDLE: the columns that precede the row (new values of a forward sweep)
        // from slot 0 up, the columns behind it (new values of a backward sweep) ending at slot maxk - 1.  A sum over all
        // slots in order is the scalar loop's, bit for bit, wherever the zeros sit (an accumulator that starts at +0 never
        // becomes -0); what the split buys is that the entries whose x may still be on its way lie in one HALF of the
        // slots per direction (Plan::late_ok: every row has at most maxk / 2 entries on either side — gs_relay.hpp sums the
        // other half above the hand-over).  Rows of a row-sharded operator (halo columns) keep the plain order.
        int nlow = 0, nupp = 0;
        bool halo = false;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
          const int32_t c = col[j];
          if (c == i) continue;
          if (c >= n) halo = true;
          if (c < i) ++nlow; else ++nupp;
        }
        const bool split = !halo;
        if (!split || 2 * nlow > d.maxk || 2 * nupp > d.maxk) late_bad.store(1, std::memory_order_relaxed);
        int kl = 0, ku = split ? d.maxk - nupp : 0;
        R dg = 0;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
          const int32_t c = col[j];
          if (c == i) { dg = val[j]; continue; }
          const int32_t q = c >= n ? c : inv[c];
          size_t lc;
          if (q >= d.row0 && q < d.row0 + d.nrows) lc = (size_t)(q - d.row0);
          else lc = (size_t)d.nrows + (size_t)where[q];
          const int k = (!split || c < i) ? kl++ : ku++;
          v[k] = val[j];
          cc[k] = (uint16_t)(lc * sizeof(R));
        }
        v[d.maxk] = dg;
        // reciprocal for the division-free quotient; 0 = "divide" (diagonals whose reciprocal or products may leave the normal range)
        const double ad = std::fabs((double)dg);
        const bool safe = sizeof(R) == 8 ? (ad > 1e-100 && ad < 1e100) : (ad > 1e-12 && ad < 1e12);
        v[d.maxk + 1] = safe ? (R)1 / dg : (R)0;
      }
      stp[nl] = (uint16_t)d.nrows;
    }
  });
  P.late_ok = late_bad.load() == 0;
  lap("records");
  // ---- modelled time of one sweep: per launch a boundary + the latency chain of a block + its bytes + its deepest block ----
  {
    double tsec = 0.0;
    for (int l = 0; l < nlaunch; ++l) {
      int depth = 0;
      double bytes = 0.0;
      for (int32_t ob = P.launch_ptr[l]; ob < P.launch_ptr[l + 1]; ++ob) {
        const Desc& d = P.blocks[ob];
        depth = std::max(depth, d.nlev);
        bytes += (double)Packed<R>::rec_bytes(d.nrows, d.maxk, d.nlev) + d.nrows * 3.0 * sizeof(R) + d.next * (4.0 + sizeof(R)) * 2;
      }
      P.sum_depth += depth;
      // measured on MI355X (profiles/r03_block_wave.log): boundary 1.6 us, write-back 0.4, a block's own load chain 1.4 (or the launch's
      // bytes at ~4.5 TB/s), 0.21 + 0.0095 maxk us per step of the walk (a single wave: ~3 ns per instruction)
      const double step = 0.21e-6 + 0.0095e-6 * maxk;
      tsec += 1.6e-6 + 0.4e-6 + std::max(1.4e-6, bytes / 4.5e12 + 0.8e-6) + depth * step;
    }
    P.est_seconds = tsec;
    // chained by flags (one launch per sweep): along the critical path a depth costs its deepest block's walk plus the hand-off
    // (write-through, flag, the fetch behind it: 2.3 us), the record loads run ahead of it; in the wide middle of the wavefront the
    // LDS slots bound the rate instead (a block holds its slot for load + wait + walk).  256^3: modelled 0.90, measured 0.93 ms
    {
      const double step = 0.21e-6 + 0.0095e-6 * maxk;
      const double slots = 256.0 * std::max<size_t>(1, (160 * 1024) / std::max<size_t>(1, P.lds_max));
      double walk_all = 0.0;
      for (const Desc& d : P.blocks) walk_all += 4.8e-6 + d.nlev * step;
      P.est_chain_seconds = (double)P.sum_depth * step + nlaunch * 2.3e-6 + 0.3 * walk_all / slots;
      // as a dataflow (rows published as they are computed, a block's walk relayed between three waves; round 5): ONE chain through
      // the operator's dependency levels — 0.31 + 0.0087 maxk us per level: hand-over + dependent tail — plus a hand-off per depth
      // of the block graph (1.1 us: write-through cell, coherent poll), plus, where the wavefront is wide, half of what the resident
      // workgroups (5 / 3 per CU) need for all blocks (6 us of load phase + the walk).  Measured (profiles/r05_dictionary_layout.log,
      // tools/bwforce_sweep.py): 256^3 level 0 0.557 ms (model 0.567), level 1 0.78 (0.87); 128^3 level 0 0.19 (0.22), level 1 0.34 (0.35)
      const double fstep = 0.31e-6 + 0.0087e-6 * maxk;
      double occupy = 0.0;
      for (const Desc& d : P.blocks) occupy += 6.0e-6 + d.nlev * fstep;
      P.est_flow_seconds = P.nlevels * fstep + nlaunch * 1.1e-6 + 0.5 * occupy / (256.0 * (maxk <= 6 ? 5 : 3)) + 8.0e-6;
    }
  }
  return true;
}

// ---- device side -----------------------------------------------------------------------------------------------------
template <typename R>
struct Args {
  const Desc* blocks; const unsigned char* rec; const int32_t* ext_col;
  const R* b; R* x; int64_t ldb, ldx; R omega; int32_t block0;
  long long* tim;   // measurement hook (tools/block_wave_bench): 4 wall-clock stamps per block, or null
};

// ---- the walk on PACKED rows ----------------------------------------------------------------------------------------
template <typename R> struct Vec16;
template <> struct Vec16<double> { typedef double2 type; };
template <> struct Vec16<float> { typedef float4 type; };

template <typename R, int MAXK>
struct PackedOps {
  static constexpr int VPC = 16 / (int)sizeof(R), NVC = (MAXK + 2 + VPC - 1) / VPC, NCC = (MAXK + 7) / 8;
  R v[NVC * VPC];        // values, then diagonal (slot MAXK) and its reciprocal (slot MAXK + 1)
  uint32_t c[NCC * 4];   // byte offsets of the columns into the LDS x, two per word
  R bb; int p;
};
template <typename R, int MAXK>
__device__ __forceinline__ void packed_load(PackedOps<R, MAXK>& o, int p, bool act, const unsigned char* rec, const R* bl) {
  typedef PackedOps<R, MAXK> O;
  constexpr int KCH = (O::NVC + O::NCC) | 1;
  o.p = act ? p : -1;
  const int q = act ? p : 0;
  const unsigned char* row = rec + __umul24((unsigned)q, 16u * KCH);   // (24-bit multiply: full rate; rows < 2^16, bytes < 2^24)
  typedef typename Vec16<R>::type V;
#pragma unroll
  for (int k = 0; k < O::NVC; ++k) *(V*)&o.v[k * O::VPC] = *(const V*)(row + 16 * k);
#pragma unroll
  for (int k = 0; k < O::NCC; ++k) *(uint4*)&o.c[4 * k] = *(const uint4*)(row + 16 * (O::NVC + k));
  o.bb = bl[q];
}
template <typename R, int MAXK, bool SOR>
__device__ __forceinline__ void packed_row(const PackedOps<R, MAXK>& o, R* xl, R omega) {
  R xv[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const uint32_t w = o.c[k >> 1];
    const uint32_t off = (k & 1) ? (w >> 16) : (w & 0xffffu);
    xv[k] = *(const R*)((const char*)xl + off);
  }
  R acc = (R)0;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) acc += o.v[k] * xv[k];
  const R dg = o.v[MAXK], rc = o.v[MAXK + 1];
  if (o.p >= 0 && dg != (R)0) {
    if (SOR) { xl[o.p] = ((R)1 - omega) * xl[o.p] + (omega / dg) * (o.bb - acc); return; }
    // (bb - acc) / dg from rc = RN(1 / dg): q0 = RN(nn rc), the exact remainder nn - dg q0 (one fma), RN(q0 + rem rc) is the
    // correctly rounded quotient (Markstein) while nothing leaves the normal range; otherwise divide
    const R nn = o.bb - acc;
    R q = nn * rc;
    const R rem = __builtin_fma(-dg, q, nn);
    q = __builtin_fma(rem, rc, q);
    const R an = __builtin_fabs(nn);
    const bool safe = sizeof(R) == 8 ? (an > (R)1e-200 && an < (R)1e200) : (an > (R)1e-25 && an < (R)1e25);
    // (a wave-uniform branch: as a plain select the compiler evaluates the whole division sequence in every step)
    if (__builtin_amdgcn_ballot_w64(!(rc != (R)0 && safe)) != 0) {
      asm volatile("; rows outside the normal range: the division itself" ::: "memory");   // (keeps the branch a branch)
      if (!(rc != (R)0 && safe)) q = nn / dg;
    }
    xl[o.p] = q;
  }
}
// A block's inputs into LDS with every global load of a thread in flight at once: a load / LDS-store loop costs one memory
// round trip per iteration (the compiler waits for each load before its store), and the external x values are two
// dependent round trips away (their positions first) — so the positions are requested first, then b, x and the record
// in batches of 8 x 16 bytes per thread, then the gathers, and only then anything is waited for.
// External columns [e0, e1) only (the chained kernel fetches the others after its flags).
// (keeps a loaded value — and so its load — above this point: without it the compiler sinks every load into the
// conditional LDS store that uses it, one memory round trip after the other)
__device__ __forceinline__ void pin(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(int32_t& v) { asm volatile("" : "+v"(v)); }
template <typename R>
__device__ __forceinline__ void block_load(const Args<R>& a, const Desc& d, const R* __restrict__ b, const R* x, R* xl, R* bl,
                                           unsigned char* rec, int recb, int e0, int e1, int tid) {
  constexpr int EU = 2, RU = 8;
  int32_t ec[EU];
#pragma unroll
  for (int k = 0; k < EU; ++k) { const int e = e0 + tid + k * kThreads; ec[k] = a.ext_col[e < e1 ? d.ext0 + e : 0]; }   // (idle lanes: entry 0, a valid position)
  R bv[2], xv[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) { const int p = tid + k * kThreads; const int q = d.row0 + (p < d.nrows ? p : 0); bv[k] = b[q]; xv[k] = x[q]; }
  const uint4* src = (const uint4*)(a.rec + (size_t)(uint32_t)d.rec * 16);
  uint4* dst = (uint4*)rec;
  const int nq = recb >> 4;
  uint4 r[RU];
#pragma unroll
  for (int k = 0; k < RU; ++k) { const int e = tid + k * kThreads; r[k] = src[e < nq ? e : 0]; }
  R xe[EU];
#pragma unroll
  for (int k = 0; k < EU; ++k) xe[k] = x[ec[k]];
#pragma unroll
  for (int k = 0; k < 2; ++k) { pin(bv[k]); pin(xv[k]); }
#pragma unroll
  for (int k = 0; k < RU; ++k) pin(r[k]);
#pragma unroll
  for (int k = 0; k < EU; ++k) pin(xe[k]);
#pragma unroll
  for (int k = 0; k < 2; ++k) { const int p = tid + k * kThreads; if (p < d.nrows) { bl[p] = bv[k]; xl[p] = xv[k]; } }
#pragma unroll
  for (int k = 0; k < RU; ++k) { const int e = tid + k * kThreads; if (e < nq) dst[e] = r[k]; }
  for (int e8 = tid + RU * kThreads; e8 < nq; e8 += RU * kThreads) {   // (records beyond 64 KB)
#pragma unroll
    for (int k = 0; k < RU; ++k) { const int e = e8 + k * kThreads; r[k] = src[e < nq ? e : 0]; }
#pragma unroll
    for (int k = 0; k < RU; ++k) pin(r[k]);
#pragma unroll
    for (int k = 0; k < RU; ++k) { const int e = e8 + k * kThreads; if (e < nq) dst[e] = r[k]; }
  }
#pragma unroll
  for (int k = 0; k < EU; ++k) { const int e = e0 + tid + k * kThreads; if (e < e1) xl[d.nrows + e] = xe[k]; }
  for (int p = tid + 2 * kThreads; p < d.nrows; p += kThreads) { bl[p] = b[d.row0 + p]; xl[p] = x[d.row0 + p]; }        // (never: blocks hold <= 1024 rows)
  for (int e = e0 + tid + EU * kThreads; e < e1; e += kThreads) xl[d.nrows + e] = x[a.ext_col[d.ext0 + e]];            // (more than 1024 external columns)
  if (tid == 0) xl[d.nrows + d.next] = (R)0;
}
// wave 0 walks the block's steps (rows of one in-block dependency level, at most 64): the next step's operands are loaded
// while the current one is summed; LDS operations of one wave execute in program order — no barrier
template <typename R, bool SOR, bool BWD, int MAXK>
__device__ __forceinline__ void packed_walk(const unsigned char* rec, const R* bl, R* xl, const Desc& d, int tid, R omega) {
  typedef PackedOps<R, MAXK> O;
  constexpr int KCH = (O::NVC + O::NCC) | 1;
  const int ns = d.nlev;
  const uint16_t* stp = (const uint16_t*)(rec + (size_t)d.nrows * (16 * KCH));
  const int lp0 = tid <= ns ? (int)stp[tid] : d.nrows;
  const int lp1 = tid + 64 <= ns ? (int)stp[tid + 64] : d.nrows;
#define BW_SP(i, out)                                                          \
  {                                                                            \
    const int i_ = (i);                                                        \
    const int u0_ = __builtin_amdgcn_readlane(lp0, i_ & 63);                   \
    const int u1_ = __builtin_amdgcn_readlane(lp1, i_ & 63);                   \
    out = i_ < 64 ? u0_ : u1_;                                                 \
  }
#define BW_RANGE(k, r0, r1)                                                    \
  {                                                                            \
    const int k_ = (k);                                                        \
    const int st_ = BWD ? ns - 1 - k_ : k_;                                    \
    const bool in_ = k_ < ns;                                                  \
    int q0_, q1_;                                                              \
    BW_SP(in_ ? st_ : 0, q0_);                                                 \
    BW_SP(in_ ? st_ + 1 : 0, q1_);                                             \
    r0 = q0_; r1 = in_ ? q1_ : q0_;                                            \
  }
  PackedOps<R, MAXK> A, B;
  int r0, r1;
  BW_RANGE(0, r0, r1);
  packed_load<R, MAXK>(A, r0 + tid, r0 + tid < r1, rec, bl);
  for (int k = 0; k < ns; k += 2) {
    BW_RANGE(k + 1, r0, r1);
    packed_load<R, MAXK>(B, r0 + tid, r0 + tid < r1, rec, bl);
    packed_row<R, MAXK, SOR>(A, xl, omega);
    BW_RANGE(k + 2, r0, r1);
    packed_load<R, MAXK>(A, r0 + tid, r0 + tid < r1, rec, bl);
    packed_row<R, MAXK, SOR>(B, xl, omega);
  }
#undef BW_RANGE
#undef BW_SP
}
template <typename R, bool SOR, bool BWD, int MAXK>
__global__ __launch_bounds__(kThreads) void gs_bw_packed_kernel(Args<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const long long t_start = a.tim ? wall_clock64() : 0;
  const Desc d = a.blocks[a.block0 + blockIdx.x];
  const int tid = threadIdx.x;
  const R* __restrict__ b = a.b + (int64_t)blockIdx.y * a.ldb;
  R* __restrict__ x = a.x + (int64_t)blockIdx.y * a.ldx;
  typedef PackedOps<R, MAXK> O;
  constexpr int KCH = (O::NVC + O::NCC) | 1;
  const int nxb = (int)(((size_t)(d.nrows + d.next + 1) * sizeof(R) + 15) & ~(size_t)15), nbb = (int)(((size_t)d.nrows * sizeof(R) + 15) & ~(size_t)15);
  R* xl = (R*)lds;
  R* bl = (R*)(lds + nxb);
  unsigned char* rec = lds + nxb + nbb;
  const int ns = d.nlev;
  const int recb = (int)(((size_t)d.nrows * (16 * KCH) + (size_t)(ns + 1) * 2 + 15) & ~(size_t)15);
  block_load<R>(a, d, b, x, xl, bl, rec, recb, 0, d.next, tid);
  __syncthreads();
  if (tid >= 64) return;
  const long long t_loaded = a.tim ? wall_clock64() : 0;
  packed_walk<R, SOR, BWD, MAXK>(rec, bl, xl, d, tid, a.omega);
  const long long t_swept = a.tim ? wall_clock64() : 0;
  for (int p = tid; p < d.nrows; p += 64) x[d.row0 + p] = xl[p];
  if (a.tim && tid == 0) {
    long long* t = a.tim + 4 * (int64_t)(a.block0 + blockIdx.x);
    t[0] = t_start; t[1] = t_loaded; t[2] = t_swept; t[3] = wall_clock64();
  }
}

// ---- the whole sweep in ONE launch: blocks chained by flags ------------------------------------------------------------
// The launches above pay, per depth of the quotient graph, a kernel boundary AND the blocks' own load (the record has to
// arrive before the walk starts) AND the deepest block of the launch.  Here every workgroup draws a ticket (its block,
// in the order of the launches: a block only waits for blocks with smaller tickets, which are running or done whatever
// the dispatch order — no deadlock), loads its record, b and x without waiting for anybody, then wave 0 polls the flags
// of the blocks it depends on, fetches the x values they wrote (agent-scope loads: past this CU's L1 and this XCD's L2),
// walks, writes its x through (agent-scope stores), drains them and raises its own flag.  Flags hold the sweep's epoch
// (the ticket counter divided by the number of blocks: no reset between sweeps, graph-replay safe).
template <typename R>
struct ChainArgs {
  Args<R> a;
  const int32_t* dep_ptr; const int32_t* dep;   // the blocks a block waits for (of this direction)
  unsigned int* flags;                          // per block: epoch of the last sweep that finished it
  unsigned long long* head;                     // ticket counter (never reset)
  int32_t nblocks;
  int32_t* err;                                 // set when a poll gave up (cannot happen; bounds a hang to seconds)
};
template <typename R> __device__ __forceinline__ R agent_load(const R* p);
template <> __device__ __forceinline__ double agent_load<double>(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <> __device__ __forceinline__ float agent_load<float>(const float* p) {
  return __uint_as_float(__hip_atomic_load((const unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void agent_store(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_store(float* p, float v) {
  __hip_atomic_store((unsigned int*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename R, bool SOR, bool BWD, int MAXK>
__global__ __launch_bounds__(kThreads) void gs_bw_chain_kernel(ChainArgs<R> c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  __shared__ unsigned long long s_ticket;
  const Args<R>& a = c.a;
  const int tid = threadIdx.x;
  const long long t_start = a.tim ? wall_clock64() : 0;
  if (tid == 0) s_ticket = atomicAdd(c.head, 1ull);
  __syncthreads();
  const unsigned long long ticket = s_ticket;
  const unsigned int epoch = (unsigned int)(ticket / (unsigned long long)c.nblocks) + 1u;
  const int t = (int)(ticket % (unsigned long long)c.nblocks);
  const int ob = BWD ? c.nblocks - 1 - t : t;
  const Desc d = a.blocks[ob];
  const R* __restrict__ b = a.b;
  R* x = a.x;
  typedef PackedOps<R, MAXK> O;
  constexpr int KCH = (O::NVC + O::NCC) | 1;
  const int nxb = (int)(((size_t)(d.nrows + d.next + 1) * sizeof(R) + 15) & ~(size_t)15), nbb = (int)(((size_t)d.nrows * sizeof(R) + 15) & ~(size_t)15);
  R* xl = (R*)lds;
  R* bl = (R*)(lds + nxb);
  unsigned char* rec = lds + nxb + nbb;
  const int ns = d.nlev;
  const int recb = (int)(((size_t)d.nrows * (16 * KCH) + (size_t)(ns + 1) * 2 + 15) & ~(size_t)15);
  // external x that nobody writes before this block has run: the far side of the sweep (and halo columns)
  const int e0 = BWD ? 0 : d.npre, e1 = BWD ? d.npre : d.next;
  block_load<R>(a, d, b, x, xl, bl, rec, recb, e0, e1, tid);
  __syncthreads();
  if (tid >= 64) return;
  const long long t_loaded = a.tim ? wall_clock64() : 0;
  // the near side: positions now (static data), values once the flags are up — one round trip behind the last flag
  const int f0 = BWD ? d.npre : 0, f1 = BWD ? d.next : d.npre;
  constexpr int FU = 6;   // 384 values by one wave in one batch (a 512-row block of a 7-point grid has 192 per side)
  int32_t fc[FU];
#pragma unroll
  for (int k = 0; k < FU; ++k) { const int e = f0 + tid + 64 * k; fc[k] = a.ext_col[e < f1 ? d.ext0 + e : 0]; }
#pragma unroll
  for (int k = 0; k < FU; ++k) pin(fc[k]);
  // the blocks this one depends on have raised their flags
  for (int e = c.dep_ptr[ob] + tid; e < c.dep_ptr[ob + 1]; e += 64) {
    const unsigned int* f = c.flags + c.dep[e];
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
      __builtin_amdgcn_s_sleep(BW_POLL_SLEEP);
      if (++spins > (1 << 24)) { *c.err = 1; break; }
    }
  }
  // (the polls above are complete for every lane before any lane goes on: the loop is left by the whole wave)
  asm volatile("" ::: "memory");   // the fetches below stay below
  {
    R fv[FU];
#pragma unroll
    for (int k = 0; k < FU; ++k) fv[k] = agent_load<R>(x + fc[k]);
#pragma unroll
    for (int k = 0; k < FU; ++k) pin(fv[k]);
#pragma unroll
    for (int k = 0; k < FU; ++k) { const int e = f0 + tid + 64 * k; if (e < f1) xl[d.nrows + e] = fv[k]; }
    for (int e = f0 + tid + 64 * FU; e < f1; e += 64) xl[d.nrows + e] = agent_load<R>(x + a.ext_col[d.ext0 + e]);
  }
  const long long t_ready = a.tim ? wall_clock64() : 0;
  packed_walk<R, SOR, BWD, MAXK>(rec, bl, xl, d, tid, a.omega);
  const long long t_swept = a.tim ? wall_clock64() : 0;
  {   // (the values out of LDS in one batch, then the stores: a read / wait / store loop pays an LDS round trip per iteration)
    constexpr int WU = 10;
    R wv[WU];
#pragma unroll
    for (int k = 0; k < WU; ++k) { const int p = tid + 64 * k; wv[k] = xl[p < d.nrows ? p : 0]; }
#pragma unroll
    for (int k = 0; k < WU; ++k) pin(wv[k]);
#pragma unroll
    for (int k = 0; k < WU; ++k) { const int p = tid + 64 * k; if (p < d.nrows) agent_store(x + d.row0 + p, wv[k]); }
    for (int p = tid + 64 * WU; p < d.nrows; p += 64) agent_store(x + d.row0 + p, xl[p]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) __hip_atomic_store(c.flags + ob, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.tim && tid == 0) {
    long long* tt = a.tim + 5 * (int64_t)ob;
    tt[0] = t_start; tt[1] = t_loaded; tt[2] = t_ready; tt[3] = t_swept; tt[4] = wall_clock64();
  }
}
template <typename R, int MAXK>
inline hipError_t sweep_chain_k(const ChainArgs<R>& c, size_t lds_max, bool sor, bool backward, hipStream_t st) {
  static std::atomic<uint64_t> attr_set{0};   // (the kernel's static 8 bytes — the ticket — count against the 160 KB as well)
  int dev = 0;
  if (lds_max > 64 * 1024 && hipGetDevice(&dev) == hipSuccess && !((attr_set.load() >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute((const void*)gs_bw_chain_kernel<R, false, false, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    (void)hipFuncSetAttribute((const void*)gs_bw_chain_kernel<R, false, true, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    (void)hipFuncSetAttribute((const void*)gs_bw_chain_kernel<R, true, false, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    (void)hipFuncSetAttribute((const void*)gs_bw_chain_kernel<R, true, true, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    attr_set.fetch_or((uint64_t)1 << (dev & 63));
  }
  const dim3 grid((unsigned)c.nblocks);
  if (sor) { if (backward) hipLaunchKernelGGL((gs_bw_chain_kernel<R, true, true, MAXK>), grid, dim3(kThreads), lds_max, st, c); else hipLaunchKernelGGL((gs_bw_chain_kernel<R, true, false, MAXK>), grid, dim3(kThreads), lds_max, st, c); }
  else { if (backward) hipLaunchKernelGGL((gs_bw_chain_kernel<R, false, true, MAXK>), grid, dim3(kThreads), lds_max, st, c); else hipLaunchKernelGGL((gs_bw_chain_kernel<R, false, false, MAXK>), grid, dim3(kThreads), lds_max, st, c); }
  return hipGetLastError();
}
// (c.dep_ptr / c.dep: the caller passes the predecessor lists for a forward sweep, the successor lists for a backward one)
template <typename R>
inline hipError_t sweep_chain(const ChainArgs<R>& c, int maxk, size_t lds_max, bool sor, bool backward, hipStream_t st) {
  switch (maxk) {
    case 6: return sweep_chain_k<R, 6>(c, lds_max, sor, backward, st);
    case 12: return sweep_chain_k<R, 12>(c, lds_max, sor, backward, st);
#if BW_PLAN_MAXK >= 18
    case 18: return sweep_chain_k<R, 18>(c, lds_max, sor, backward, st);
#endif
  }
  return hipErrorInvalidValue;
}

// launches of one directional sweep: blocks of launch l = [launch_ptr[l], launch_ptr[l + 1])
template <typename R, int MAXK>
inline hipError_t sweep_k(const Args<R>& a0, const std::vector<int32_t>& launch_ptr, size_t lds_max, bool sor, bool backward, int ncols,
                          hipStream_t st) {
  // more than the default 64 KB of dynamic LDS needs the attribute: once per kernel and device
  static std::atomic<uint64_t> attr_set{0};
  int dev = 0;
  if (lds_max > 64 * 1024 && hipGetDevice(&dev) == hipSuccess && !((attr_set.load() >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute((const void*)gs_bw_packed_kernel<R, false, false, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gs_bw_packed_kernel<R, false, true, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gs_bw_packed_kernel<R, true, false, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gs_bw_packed_kernel<R, true, true, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.fetch_or((uint64_t)1 << (dev & 63));
  }
  const int nl = (int)launch_ptr.size() - 1;
  Args<R> a = a0;
  for (int s = 0; s < nl; ++s) {
    const int l = backward ? nl - 1 - s : s;
    a.block0 = launch_ptr[l];
    const dim3 grid((unsigned)(launch_ptr[l + 1] - launch_ptr[l]), (unsigned)ncols);
    if (sor) { if (backward) hipLaunchKernelGGL((gs_bw_packed_kernel<R, true, true, MAXK>), grid, dim3(kThreads), lds_max, st, a); else hipLaunchKernelGGL((gs_bw_packed_kernel<R, true, false, MAXK>), grid, dim3(kThreads), lds_max, st, a); }
    else { if (backward) hipLaunchKernelGGL((gs_bw_packed_kernel<R, false, true, MAXK>), grid, dim3(kThreads), lds_max, st, a); else hipLaunchKernelGGL((gs_bw_packed_kernel<R, false, false, MAXK>), grid, dim3(kThreads), lds_max, st, a); }
  }
  return hipGetLastError();
}
constexpr int kMaxK = BW_PLAN_MAXK;   // longest rows (off-diagonal entries, padded to a multiple of kChunk) the kernels are instantiated for
template <typename R>
inline hipError_t sweep(const Args<R>& a, int maxk, const std::vector<int32_t>& launch_ptr, size_t lds_max, bool sor, bool backward, int ncols,
                        hipStream_t st) {
  switch (maxk) {
    case 6: return sweep_k<R, 6>(a, launch_ptr, lds_max, sor, backward, ncols, st);
    case 12: return sweep_k<R, 12>(a, launch_ptr, lds_max, sor, backward, ncols, st);
#if BW_PLAN_MAXK >= 18
    case 18: return sweep_k<R, 18>(a, launch_ptr, lds_max, sor, backward, ncols, st);
#endif
  }
  return hipErrorInvalidValue;
}

}  // namespace bw
}  // namespace amgh
