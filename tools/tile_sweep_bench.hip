// tile_sweep_bench — an exact lexicographic Gauss-Seidel forward sweep where a launch covers a GROUP of m dependency
// levels WITHOUT substitution: the rows of the group are split into tiles (a partition of the matrix graph), one
// workgroup per tile walks the m levels one after the other with a workgroup barrier in between, keeping the values
// it produces in LDS; the in-group ancestors a tile needs from its neighbours are RECOMPUTED by the tile itself
// (overlapping "trapezoid" tiles), so workgroups never wait for each other inside a launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/tile_sweep_bench tools/tile_sweep_bench.hip
// Compared with the shipped merged groups (rows of a group made independent by substituting their in-group
// ancestors: composite rows inflate with depth, so groups stay 3-4 levels deep), redundancy here grows with
// depth / tile width instead of exponentially with depth: deeper groups, fewer launches.
// usage: tile_sweep_bench poisson N m mode W      mode = block (W x W columns of the grid) | blob (BFS-grown parts of W rows)
//        tile_sweep_bench file  PATH m blob W     PATH: int64 n, int64 nnz, int32 rowptr[n+1], int32 col[nnz], double val[nnz]
// Prints redundancy, tiles per launch, LDS need, microseconds per launch / per dependency level / per sweep, and the
// difference to the scalar sweep on the host.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Csr { int64_t n = 0; std::vector<int32_t> rp, ci; std::vector<double> va; };

static Csr poisson3(int N) {
  Csr A; A.n = (int64_t)N * N * N; A.rp.assign(A.n + 1, 0);
  A.ci.reserve(7 * A.n); A.va.reserve(7 * A.n);
  for (int k = 0; k < N; ++k) for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) {
    const int64_t r = i + (int64_t)N * (j + (int64_t)N * k);
    if (k > 0) { A.ci.push_back((int32_t)(r - (int64_t)N * N)); A.va.push_back(-1.0); }
    if (j > 0) { A.ci.push_back((int32_t)(r - N)); A.va.push_back(-1.0); }
    if (i > 0) { A.ci.push_back((int32_t)(r - 1)); A.va.push_back(-1.0); }
    A.ci.push_back((int32_t)r); A.va.push_back(6.0);
    if (i < N - 1) { A.ci.push_back((int32_t)(r + 1)); A.va.push_back(-1.0); }
    if (j < N - 1) { A.ci.push_back((int32_t)(r + N)); A.va.push_back(-1.0); }
    if (k < N - 1) { A.ci.push_back((int32_t)(r + (int64_t)N * N)); A.va.push_back(-1.0); }
    A.rp[r + 1] = (int32_t)A.ci.size();
  }
  return A;
}
static Csr load(const char* path) {
  Csr A; FILE* f = fopen(path, "rb"); if (!f) { printf("cannot open %s\n", path); exit(1); }
  int64_t hdr[2]; if (fread(hdr, 8, 2, f) != 2) exit(1);
  A.n = hdr[0]; A.rp.resize(A.n + 1); A.ci.resize(hdr[1]); A.va.resize(hdr[1]);
  if (fread(A.rp.data(), 4, A.n + 1, f) != (size_t)A.n + 1 || fread(A.ci.data(), 4, hdr[1], f) != (size_t)hdr[1] ||
      fread(A.va.data(), 8, hdr[1], f) != (size_t)hdr[1]) exit(1);
  fclose(f); return A;
}

// ---- device side ---------------------------------------------------------------------------------------------------
struct Tile {            // one workgroup of one launch
  int32_t row0;          // first local row in grow / diag (local rows are sorted by level)
  int32_t nloc;          // owned + recomputed rows
  int32_t ext0, next;    // external columns (final values of earlier groups) gathered into LDS at the start
  int32_t sub0;          // index into sub_ptr / sub_ent (m + 1 entries / m entries)
};
struct Args {
  const Tile* tiles; const int32_t* sub_ptr; const int64_t* sub_ent; const int32_t* sub_k;
  const int32_t* grow; const double* diag; const int32_t* ext_col; const int32_t* eidx; const double* eval;
  const double* s; double* x; int m;
};
constexpr int kT = 512;

// s = b - U x (the strictly upper triangle, old values): the pre-pass the shipped merged sweeps use too
__global__ void prepass_kernel(const int32_t* rp, const int32_t* ci, const double* va, const double* b, const double* x,
                               int64_t n, double* s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = b[i];
  for (int32_t j = rp[i]; j < rp[i + 1]; ++j) if (ci[j] > i) acc -= va[j] * x[ci[j]];
  s[i] = acc;
}

// the pre-pass in tile-local order: s_loc[local row] = b - U x for every local row of every tile (recomputed copies too)
__global__ void prepass_local_kernel(const int32_t* rp, const int32_t* ci, const double* va, const double* b, const double* x,
                                     const int32_t* grow, int64_t nlocal, double* s_loc) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nlocal) return;
  const int64_t i = grow[k] & 0x7fffffff;
  double acc = b[i];
  for (int32_t j = rp[i]; j < rp[i + 1]; ++j) if (ci[j] > i) acc -= va[j] * x[ci[j]];
  s_loc[k] = acc;
}

// One workgroup = one tile of one group.  The global loads run AHEAD of the sub-levels: the external x values go to
// LDS first, and — one row per thread and sub-level — the rows' operands (global row, diagonal or its reciprocal, s,
// entries) are loaded into registers a CHUNK of MC sub-levels ahead: while chunk c is computed from LDS and registers
// (one workgroup barrier per sub-level), the loads of chunk c + 1 are in flight.  The tile descriptor carries the
// sub-level table, so the chain of dependent round trips is: descriptor -> operands -> s / x gathers.
constexpr int kMaxDepth = 32;
struct TileD {
  int32_t row0, nloc, ext0, next;
  int32_t sub_ptr[kMaxDepth + 1];
  int32_t sub_k[kMaxDepth];
  int64_t sub_ent[kMaxDepth];
};
struct ArgsD {
  const TileD* tiles; const int32_t* grow; const double* diag; const int32_t* ext_col; const int32_t* eidx; const double* eval;
  const int4* pk_i; const double4* pk_d;   // K <= 3: one row = {idx0, idx1, idx2, global row} + {val0, val1, val2, diag}: three 16-byte loads
  const double* s; double* x; int m; int use_rcp; int packed;
  const double* s_loc; int local_s;   // s in tile-local order (written by a tile-shaped pre-pass): no dependent load behind the row id   // stage: 0 full | 1 operand loads only | 2 sub-level loop only
};
// a workgroup barrier that orders LDS only: it does not wait for global loads or stores in flight (__syncthreads()
// waits for vmcnt(0): every sub-level would pay the write latency of its own x stores, and the prefetch its loads)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int MC, int KMAX>
struct Ops { int32_t g[MC], r0[MC]; double d[MC], sv[MC]; int32_t idx[MC][KMAX]; double val[MC][KMAX]; };

template <int MC, int KMAX>
__device__ __forceinline__ void fetch_chunk(const ArgsD& a, const TileD& t, int q0, int tid, int zero_slot, Ops<MC, KMAX>& o) {
#pragma unroll
  for (int q = 0; q < MC; ++q) {
    o.g[q] = -2; o.d[q] = 1.0; o.sv[q] = 0.0; o.r0[q] = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { o.idx[q][k] = zero_slot; o.val[q][k] = 0.0; }
    const int ql = q0 + q;
    if (ql < a.m) {
      const int r0 = t.sub_ptr[ql], nr = t.sub_ptr[ql + 1] - r0;
      o.r0[q] = r0;
      if (tid < nr) {
        if (KMAX == 3 && a.packed) {
          const int4 pi = a.pk_i[t.row0 + r0 + tid];
          const double4 pd = a.pk_d[t.row0 + r0 + tid];
          o.idx[q][0] = pi.x; o.idx[q][1] = pi.y; o.idx[q][KMAX - 1] = pi.z; o.g[q] = pi.w;
          o.val[q][0] = pd.x; o.val[q][1] = pd.y; o.val[q][KMAX - 1] = pd.z; o.d[q] = pd.w;
        } else {
          const int K = t.sub_k[ql];
          const int64_t base = t.sub_ent[ql];
          o.g[q] = a.grow[t.row0 + r0 + tid];
          o.d[q] = a.diag[t.row0 + r0 + tid];
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            if (k < K) { o.idx[q][k] = a.eidx[base + (int64_t)k * nr + tid]; o.val[q][k] = a.eval[base + (int64_t)k * nr + tid]; }
        }
      }
    }
  }
  if (a.local_s) {
#pragma unroll
    for (int q = 0; q < MC; ++q) {
      const int ql = q0 + q;
      if (ql < a.m) { const int r0 = t.sub_ptr[ql], nr = t.sub_ptr[ql + 1] - r0; if (tid < nr) o.sv[q] = a.s_loc[t.row0 + r0 + tid]; }
    }
  } else {
#pragma unroll
    for (int q = 0; q < MC; ++q) if (o.g[q] != -2) o.sv[q] = a.s[o.g[q] & 0x7fffffff];
  }
}

template <int MC, int KMAX, int THREADS, int STAGE>
__global__ __launch_bounds__(THREADS) void tile_kernel(ArgsD a, int tile0) {
  extern __shared__ __attribute__((aligned(16))) double xl[];
  const TileD& t = a.tiles[tile0 + blockIdx.x];
  const int tid = threadIdx.x;
  const int zero_slot = t.nloc + t.next;
  for (int e = tid; e < t.next; e += THREADS) xl[t.nloc + e] = a.x[a.ext_col[t.ext0 + e]];
  if (tid == 0) xl[zero_slot] = 0.0;
  Ops<MC, KMAX> cur, nxt;
  if (STAGE == 2) {       // ablation: no operand loads (every thread a dummy row on the zero slot)
#pragma unroll
    for (int q = 0; q < MC; ++q) {
      cur.g[q] = -1; cur.d[q] = 1.0; cur.sv[q] = 1.0; cur.r0[q] = 0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) { cur.idx[q][k] = zero_slot; cur.val[q][k] = 0.5; }
    }
    nxt = cur;
  } else {
    fetch_chunk<MC, KMAX>(a, t, 0, tid, zero_slot, cur);
  }
  __syncthreads();
  if (STAGE == 1) {       // ablation: operand loads only
    double acc = 0.0;
    for (int q0 = 0; q0 < a.m; q0 += MC) {
      if (q0 + MC < a.m) fetch_chunk<MC, KMAX>(a, t, q0 + MC, tid, zero_slot, nxt);
#pragma unroll
      for (int q = 0; q < MC; ++q) {
        acc += cur.sv[q] * cur.d[q];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) acc += cur.val[q][k] * (double)cur.idx[q][k];
      }
      cur = nxt;
    }
    if (acc == 1.2345e300) a.x[0] = acc;
    return;
  }
  for (int q0 = 0; q0 < a.m; q0 += MC) {
    if (STAGE == 0 && q0 + MC < a.m) fetch_chunk<MC, KMAX>(a, t, q0 + MC, tid, zero_slot, nxt);
#pragma unroll
    for (int q = 0; q < MC; ++q) {
      if (q0 + q < a.m) {
        if (cur.g[q] != -2) {
          double acc = cur.sv[q];
#pragma unroll
          for (int k = 0; k < KMAX; ++k) acc -= cur.val[q][k] * xl[cur.idx[q][k]];
          const double xn = a.use_rcp ? acc * cur.d[q] : acc / cur.d[q];
          xl[cur.r0[q] + tid] = xn;
          if (cur.g[q] >= 0) a.x[cur.g[q]] = xn;   // owned rows only (bit 31 = recomputed copy of a neighbour's row)
          else if (STAGE == 2 && xn == 1.2345e300) a.x[0] = xn;
        }
        lds_barrier();
      }
    }
    cur = nxt;
  }
}

// ---- host construction -----------------------------------------------------------------------------------------------
struct Built {
  std::vector<Tile> tiles; std::vector<int32_t> tile_ptr;   // tiles of launch g: [tile_ptr[g], tile_ptr[g+1])
  std::vector<int32_t> sub_ptr, sub_k; std::vector<int64_t> sub_ent;
  std::vector<int32_t> grow, ext_col, eidx; std::vector<double> diag, eval;
  int64_t owned = 0, local = 0, ext = 0, ents = 0, pad = 0; int max_lds = 0;
};

int main(int argc, char** argv) {
  if (argc < 6) { printf("usage: %s poisson N m block|blob W [threads=512] [rcp=0] [chunk=4] [packed=1] [local_s=0]   |   %s file PATH m blob W [threads] [rcp] [chunk]\n", argv[0], argv[0]); return 1; }
  const bool is_poisson = !strcmp(argv[1], "poisson");
  const int N = is_poisson ? atoi(argv[2]) : 0;
  Csr A = is_poisson ? poisson3(N) : load(argv[2]);
  const int m = atoi(argv[3]);
  const bool block_mode = !strcmp(argv[4], "block");
  const int W = atoi(argv[5]);
  const int64_t n = A.n;
  printf("n = %lld, nnz = %lld, group depth m = %d, %s W = %d\n", (long long)n, (long long)A.rp[n], m, argv[4], W);
  // dependency levels of the forward sweep
  std::vector<int32_t> lev(n, 0);
  int nlev = 0;
  for (int64_t i = 0; i < n; ++i) {
    int l = 0;
    for (int32_t j = A.rp[i]; j < A.rp[i + 1]; ++j) if (A.ci[j] < i) l = std::max(l, lev[A.ci[j]] + 1);
    lev[i] = l; nlev = std::max(nlev, l + 1);
  }
  printf("dependency levels: %d\n", nlev);
  // partition of the rows
  std::vector<int32_t> part(n, -1);
  int nparts = 0;
  if (block_mode) {
    if (!is_poisson) { printf("block mode needs the grid\n"); return 1; }
    const int nb = (N + W - 1) / W;
    for (int k = 0; k < N; ++k) for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i)
      part[i + (int64_t)N * (j + (int64_t)N * k)] = (j / W) + nb * (k / W);
    nparts = nb * nb;
  } else {   // BFS-grown parts of about W rows on the symmetric graph, seeds in natural order
    std::vector<int32_t> queue;
    for (int64_t seed = 0; seed < n; ++seed) {
      if (part[seed] >= 0) continue;
      queue.clear(); queue.push_back((int32_t)seed); part[seed] = nparts;
      size_t head = 0;
      while (head < queue.size() && (int)queue.size() < W) {
        const int32_t r = queue[head++];
        for (int32_t j = A.rp[r]; j < A.rp[r + 1] && (int)queue.size() < W; ++j) {
          const int32_t c = A.ci[j];
          if (c != r && part[c] < 0) { part[c] = nparts; queue.push_back(c); }
        }
      }
      ++nparts;
    }
  }
  printf("parts: %d (%.0f rows each)\n", nparts, (double)n / nparts);
  // rows of every level, and the groups
  const int ngrp = (nlev + m - 1) / m;
  std::vector<int64_t> gcount(ngrp + 1, 0);
  for (int64_t i = 0; i < n; ++i) gcount[lev[i] / m + 1]++;
  for (int g = 0; g < ngrp; ++g) gcount[g + 1] += gcount[g];
  std::vector<int32_t> grows(n);
  { std::vector<int64_t> w(gcount.begin(), gcount.end() - 1); for (int64_t i = 0; i < n; ++i) grows[w[lev[i] / m]++] = (int32_t)i; }
  // build the tiles, groups in parallel over host threads
  const int HT = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  std::vector<Built> parts_built(HT);
  std::vector<std::vector<int>> grp_of_thread(HT);
  for (int g = 0; g < ngrp; ++g) grp_of_thread[g % HT].push_back(g);
  auto work = [&](int th) {
    Built& B = parts_built[th];
    std::vector<int32_t> stamp(n, -1), locid(n, 0);
    std::vector<int32_t> rows, stack, exts, order;
    std::vector<std::pair<int32_t, int32_t>> byp;
    int32_t tick = 0;
    for (int g : grp_of_thread[th]) {
      const int l0 = g * m;
      byp.clear();
      for (int64_t q = gcount[g]; q < gcount[g + 1]; ++q) byp.push_back({part[grows[q]], grows[q]});
      std::sort(byp.begin(), byp.end());
      B.tile_ptr.push_back((int32_t)B.tiles.size());   // (per-thread; merged later in group order)
      size_t a = 0;
      while (a < byp.size()) {
        size_t b = a;
        while (b < byp.size() && byp[b].first == byp[a].first) ++b;
        // closure of the owned rows inside the group
        ++tick; rows.clear(); stack.clear();
        for (size_t q = a; q < b; ++q) { const int32_t r = byp[q].second; stamp[r] = tick; rows.push_back(r); stack.push_back(r); }
        const size_t nown = rows.size();
        while (!stack.empty()) {
          const int32_t r = stack.back(); stack.pop_back();
          for (int32_t j = A.rp[r]; j < A.rp[r + 1]; ++j) {
            const int32_t c = A.ci[j];
            if (c < r && lev[c] >= l0 && stamp[c] != tick) { stamp[c] = tick; rows.push_back(c); stack.push_back(c); }
          }
        }
        // local order: by level, then row index; owned flag = came from the first nown entries
        order.resize(rows.size());
        for (size_t q = 0; q < rows.size(); ++q) order[q] = (int32_t)q;
        std::sort(order.begin(), order.end(), [&](int32_t p, int32_t q2) {
          return lev[rows[p]] != lev[rows[q2]] ? lev[rows[p]] < lev[rows[q2]] : rows[p] < rows[q2]; });
        Tile T{};
        T.row0 = (int32_t)B.grow.size(); T.nloc = (int32_t)rows.size(); T.ext0 = (int32_t)B.ext_col.size();
        T.sub0 = (int32_t)B.sub_ptr.size();
        for (size_t q = 0; q < order.size(); ++q) locid[rows[order[q]]] = (int32_t)q;
        // external columns
        exts.clear();
        const int32_t tick_ext = ++tick;   // a second stamp value for "is an external column of this tile"
        std::vector<int32_t>& ext_stamp = stamp;  // rows of the tile keep the previous tick: distinguish by level
        for (size_t q = 0; q < order.size(); ++q) {
          const int32_t r = rows[order[q]];
          for (int32_t j = A.rp[r]; j < A.rp[r + 1]; ++j) {
            const int32_t c = A.ci[j];
            if (c < r && lev[c] < l0 && ext_stamp[c] != tick_ext) { ext_stamp[c] = tick_ext; locid[c] = T.nloc + (int32_t)exts.size(); exts.push_back(c); }
          }
        }
        T.next = (int32_t)exts.size();
        B.ext_col.insert(B.ext_col.end(), exts.begin(), exts.end());
        // sub-levels
        size_t q = 0;
        for (int sl = 0; sl < m; ++sl) {
          B.sub_ptr.push_back((int32_t)q);
          size_t e = q;
          while (e < order.size() && lev[rows[order[e]]] == l0 + sl) ++e;
          const int nr = (int)(e - q);
          int K = 0;
          for (size_t z = q; z < e; ++z) {
            const int32_t r = rows[order[z]]; int len = 0;
            for (int32_t j = A.rp[r]; j < A.rp[r + 1]; ++j) if (A.ci[j] < r) ++len;
            K = std::max(K, len);
          }
          B.sub_k.push_back(K);
          B.sub_ent.push_back((int64_t)B.eidx.size());
          const size_t base = B.eidx.size();
          B.eidx.resize(base + (size_t)K * nr, 0);
          B.eval.resize(base + (size_t)K * nr, 0.0);
          for (size_t z = q; z < e; ++z) {
            const int32_t r = rows[order[z]]; int k = 0; double d = 0.0;
            for (int32_t j = A.rp[r]; j < A.rp[r + 1]; ++j) {
              const int32_t c = A.ci[j];
              if (c < r) { B.eidx[base + (size_t)k * nr + (z - q)] = locid[c]; B.eval[base + (size_t)k * nr + (z - q)] = A.va[j]; ++k; }
              else if (c == r) d = A.va[j];
            }
            B.ents += k; B.pad += K - k;
            const bool owned = (size_t)order[z] < nown;
            B.grow.push_back(owned ? r : (int32_t)(r | 0x80000000));
            B.diag.push_back(d);
          }
          q = e;
        }
        B.sub_ptr.push_back((int32_t)q);
        B.tiles.push_back(T);
        B.owned += (int64_t)nown; B.local += T.nloc; B.ext += T.next;
        B.max_lds = std::max(B.max_lds, T.nloc + T.next + 1);
        a = b;
      }
    }
    B.tile_ptr.push_back((int32_t)B.tiles.size());
  };
  {
    std::vector<std::thread> th;
    for (int t = 0; t < HT; ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  // merge the per-thread pieces in group order (thread t built groups t, t + HT, ...)
  Built B;
  std::vector<int32_t> launch_tile0(ngrp + 1, 0);
  {
    std::vector<size_t> cursor(HT, 0);
    for (int g = 0; g < ngrp; ++g) {
      const int t = g % HT; Built& P = parts_built[t];
      const size_t k = cursor[t]++;
      launch_tile0[g] = (int32_t)B.tiles.size();
      for (int32_t z = P.tile_ptr[k]; z < P.tile_ptr[k + 1]; ++z) {
        Tile T = P.tiles[z];
        const int32_t row0 = T.row0, ext0 = T.ext0, sub0 = T.sub0;
        const int tix = sub0 / (m + 1);
        T.row0 = (int32_t)B.grow.size(); T.ext0 = (int32_t)B.ext_col.size(); T.sub0 = (int32_t)B.sub_ptr.size();
        B.grow.insert(B.grow.end(), P.grow.begin() + row0, P.grow.begin() + row0 + T.nloc);
        B.diag.insert(B.diag.end(), P.diag.begin() + row0, P.diag.begin() + row0 + T.nloc);
        B.ext_col.insert(B.ext_col.end(), P.ext_col.begin() + ext0, P.ext_col.begin() + ext0 + T.next);
        for (int sl = 0; sl <= m; ++sl) B.sub_ptr.push_back(P.sub_ptr[sub0 + sl]);
        for (int sl = 0; sl < m; ++sl) {
          const int64_t e0 = P.sub_ent[(size_t)tix * m + sl];
          const int nr = P.sub_ptr[sub0 + sl + 1] - P.sub_ptr[sub0 + sl];
          const int K = P.sub_k[(size_t)tix * m + sl];
          B.sub_k.push_back(K);
          B.sub_ent.push_back((int64_t)B.eidx.size());
          B.eidx.insert(B.eidx.end(), P.eidx.begin() + e0, P.eidx.begin() + e0 + (int64_t)K * nr);
          B.eval.insert(B.eval.end(), P.eval.begin() + e0, P.eval.begin() + e0 + (int64_t)K * nr);
        }
        B.tiles.push_back(T);
      }
      B.owned += 0;
    }
    launch_tile0[ngrp] = (int32_t)B.tiles.size();
    for (int t = 0; t < HT; ++t) {
      B.owned += parts_built[t].owned; B.local += parts_built[t].local; B.ext += parts_built[t].ext;
      B.ents += parts_built[t].ents; B.pad += parts_built[t].pad; B.max_lds = std::max(B.max_lds, parts_built[t].max_lds);
    }
  }
  int64_t low = 0;
  for (int64_t i = 0; i < n; ++i) for (int32_t j = A.rp[i]; j < A.rp[i + 1]; ++j) low += A.ci[j] < i;
  printf("tiles: %zu over %d launches (%.0f per launch); rows computed / rows owned = %.3f; lower entries streamed / lower "
         "entries = %.3f (+ %.1f %% padding); external gathers per owned row %.2f; largest tile needs %d doubles of LDS\n",
         B.tiles.size(), ngrp, (double)B.tiles.size() / ngrp, (double)B.local / B.owned, (double)B.ents / low,
         100.0 * B.pad / std::max<int64_t>(1, B.ents), (double)B.ext / B.owned, B.max_lds);
  const double bytes = 12.0 * (B.ents + B.pad) + 12.0 * B.local + 4.0 * B.ext + 8.0 * B.ext + 8.0 * B.owned + 8.0 * B.local;
  printf("bytes per sweep: %.2f GB (the level-by-level sweep streams %.2f GB)\n", bytes / 1e9, (12.0 * low + 28.0 * n) / 1e9);
  const int threads = argc > 6 ? atoi(argv[6]) : 512;
  const int use_rcp = argc > 7 ? atoi(argv[7]) : 0;
  const size_t lds_bytes = (size_t)(B.max_lds + 8) * 8;
  if (lds_bytes > 160 * 1024) { printf("largest tile needs %zu bytes of LDS: smaller W or m\n", lds_bytes); return 1; }
  if (m > kMaxDepth) { printf("m > %d\n", kMaxDepth); return 1; }
  // ---- device ----
  auto up = [](const void* h, size_t bytes_) { void* d; CHECK(hipMalloc(&d, std::max<size_t>(bytes_, 8))); CHECK(hipMemcpy(d, h, bytes_, hipMemcpyHostToDevice)); return d; };
  std::vector<double> b(n), x0(n);
  uint64_t sd = 88172645463325252ull;
  for (int64_t i = 0; i < n; ++i) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; b[i] = (double)(sd >> 11) / 9007199254740992.0; sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; x0[i] = (double)(sd >> 11) / 9007199254740992.0 - 0.5; }
  std::vector<TileD> td(B.tiles.size());
  for (size_t z = 0; z < B.tiles.size(); ++z) {
    const Tile& T = B.tiles[z]; TileD& D = td[z];
    memset(&D, 0, sizeof(D));
    D.row0 = T.row0; D.nloc = T.nloc; D.ext0 = T.ext0; D.next = T.next;
    const int tix = T.sub0 / (m + 1);
    for (int q = 0; q <= m; ++q) D.sub_ptr[q] = B.sub_ptr[T.sub0 + q];
    for (int q = 0; q < m; ++q) { D.sub_k[q] = B.sub_k[(size_t)tix * m + q]; D.sub_ent[q] = B.sub_ent[(size_t)tix * m + q]; }
  }
  std::vector<double> dg(B.diag);
  if (use_rcp) for (double& v : dg) v = 1.0 / v;
  const int packed = argc > 9 ? atoi(argv[9]) : 1;
  std::vector<int4> pki; std::vector<double4> pkd;
  int kmax0 = 0;
  for (int32_t k : B.sub_k) kmax0 = std::max(kmax0, k);
  if (packed && kmax0 <= 3) {
    pki.resize(B.grow.size()); pkd.resize(B.grow.size());
    for (size_t z = 0; z < B.tiles.size(); ++z) {
      const Tile& T = B.tiles[z]; const int tix = T.sub0 / (m + 1);
      const int zero_slot = T.nloc + T.next;
      for (int q = 0; q < m; ++q) {
        const int r0 = B.sub_ptr[T.sub0 + q], nr = B.sub_ptr[T.sub0 + q + 1] - r0, K = B.sub_k[(size_t)tix * m + q];
        const int64_t base = B.sub_ent[(size_t)tix * m + q];
        for (int r = 0; r < nr; ++r) {
          int ix[3] = {zero_slot, zero_slot, zero_slot}; double vv[3] = {0.0, 0.0, 0.0};
          for (int k = 0; k < K; ++k) { ix[k] = B.eidx[base + (int64_t)k * nr + r]; vv[k] = B.eval[base + (int64_t)k * nr + r]; }
          pki[T.row0 + r0 + r] = make_int4(ix[0], ix[1], ix[2], B.grow[T.row0 + r0 + r]);
          pkd[T.row0 + r0 + r] = make_double4(vv[0], vv[1], vv[2], dg[T.row0 + r0 + r]);
        }
      }
    }
  }
  ArgsD a{};
  a.pk_i = (const int4*)up(pki.data(), pki.size() * sizeof(int4)); a.pk_d = (const double4*)up(pkd.data(), pkd.size() * sizeof(double4));
  a.packed = (packed && kmax0 <= 3) ? 1 : 0;
  a.tiles = (const TileD*)up(td.data(), td.size() * sizeof(TileD));
  a.grow = (const int32_t*)up(B.grow.data(), B.grow.size() * 4);
  a.diag = (const double*)up(dg.data(), dg.size() * 8);
  a.ext_col = (const int32_t*)up(B.ext_col.data(), B.ext_col.size() * 4);
  a.eidx = (const int32_t*)up(B.eidx.data(), B.eidx.size() * 4);
  a.eval = (const double*)up(B.eval.data(), B.eval.size() * 8);
  int32_t* d_rp = (int32_t*)up(A.rp.data(), (n + 1) * 4); int32_t* d_ci = (int32_t*)up(A.ci.data(), A.ci.size() * 4);
  double* d_va = (double*)up(A.va.data(), A.va.size() * 8);
  double* d_b = (double*)up(b.data(), n * 8); double* d_x = (double*)up(x0.data(), n * 8);
  double* d_s; CHECK(hipMalloc(&d_s, n * 8));
  const int local_s = argc > 10 ? atoi(argv[10]) : 0;
  double* d_sloc; CHECK(hipMalloc(&d_sloc, std::max<size_t>(8, B.grow.size() * 8)));
  a.s = d_s; a.x = d_x; a.m = m; a.use_rcp = use_rcp; a.s_loc = d_sloc; a.local_s = local_s;
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
  int kmax = 0, max_nr = 0;
  for (int32_t k : B.sub_k) kmax = std::max(kmax, k);
  for (size_t z = 0; z + 1 < B.sub_ptr.size(); ++z) if ((z + 1) % (m + 1) != 0) max_nr = std::max(max_nr, B.sub_ptr[z + 1] - B.sub_ptr[z]);
  printf("longest lower row: %d entries; most rows of one tile in one sub-level: %d (one thread each, %d threads); LDS per tile <= %zu B; "
         "x = acc %s; %s\n", kmax, max_nr, threads, lds_bytes, use_rcp ? "* (1 / diag)" : "/ diag", a.packed ? "rows packed for 16-byte loads" : "ELL arrays");
  if (max_nr > threads) { printf("a sub-level of a tile has more rows than threads: smaller W or more threads\n"); return 1; }
  if (kmax > 16) { printf("no kernel instance for %d entries per row\n", kmax); return 1; }
  const int mc = argc > 8 ? atoi(argv[8]) : 4;    // sub-levels per register chunk
#define SETATTR(MC, KK, TT) \
  CHECK(hipFuncSetAttribute((const void*)tile_kernel<MC, KK, TT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  CHECK(hipFuncSetAttribute((const void*)tile_kernel<MC, KK, TT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  CHECK(hipFuncSetAttribute((const void*)tile_kernel<MC, KK, TT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  SETATTR(8, 3, 512); SETATTR(4, 3, 512); SETATTR(2, 3, 512); SETATTR(4, 3, 1024); SETATTR(2, 3, 1024);
  SETATTR(4, 8, 512); SETATTR(2, 8, 512); SETATTR(2, 16, 512); SETATTR(1, 16, 512); SETATTR(2, 8, 1024); SETATTR(1, 16, 1024);
  int stage = 0;
  auto launch = [&](int nt, int t0) {
#define TL3(MC, KK, TT) do { if (stage == 0) hipLaunchKernelGGL((tile_kernel<MC, KK, TT, 0>), dim3(nt), dim3(TT), lds_bytes, st, a, t0); \
                             else if (stage == 1) hipLaunchKernelGGL((tile_kernel<MC, KK, TT, 1>), dim3(nt), dim3(TT), lds_bytes, st, a, t0); \
                             else hipLaunchKernelGGL((tile_kernel<MC, KK, TT, 2>), dim3(nt), dim3(TT), lds_bytes, st, a, t0); } while (0)
    if (threads <= 512) {
      if (kmax <= 3) { if (mc >= 8) TL3(8, 3, 512); else if (mc >= 4) TL3(4, 3, 512); else TL3(2, 3, 512); }
      else if (kmax <= 8) { if (mc >= 4) TL3(4, 8, 512); else TL3(2, 8, 512); }
      else { if (mc >= 2) TL3(2, 16, 512); else TL3(1, 16, 512); }
    } else {
      if (kmax <= 3) { if (mc >= 4) TL3(4, 3, 1024); else TL3(2, 3, 1024); }
      else if (kmax <= 8) TL3(2, 8, 1024);
      else TL3(1, 16, 1024);
    }
#undef TL3
  };
  auto sweep = [&]() {
    if (local_s) {
      const int64_t nl = (int64_t)B.grow.size();
      hipLaunchKernelGGL(prepass_local_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, d_rp, d_ci, d_va, d_b, d_x, a.grow, nl, d_sloc);
    } else {
      hipLaunchKernelGGL(prepass_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_rp, d_ci, d_va, d_b, d_x, n, d_s);
    }
    CHECK(hipEventRecord(e1, st));
    for (int g = 0; g < ngrp; ++g) {
      const int nt = launch_tile0[g + 1] - launch_tile0[g];
      if (nt > 0) launch(nt, launch_tile0[g]);
    }
  };
  // correctness: one sweep from x0 against the scalar loop
  sweep(); CHECK(hipStreamSynchronize(st)); CHECK(hipGetLastError());
  std::vector<double> xd(n), xr(x0);
  CHECK(hipMemcpy(xd.data(), d_x, n * 8, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; ++i) {
    double rs = 0.0, d = 0.0;
    for (int32_t j = A.rp[i]; j < A.rp[i + 1]; ++j) { if (A.ci[j] == i) d = A.va[j]; else rs += A.va[j] * xr[A.ci[j]]; }
    xr[i] = (b[i] - rs) / d;
  }
  double num = 0.0, den = 0.0;
  for (int64_t i = 0; i < n; ++i) { num += (xd[i] - xr[i]) * (xd[i] - xr[i]); den += xr[i] * xr[i]; }
  printf("||x_tiles - x_scalar|| / ||x_scalar|| = %.3e\n", std::sqrt(num / den));
  // timing
  for (int it = 0; it < 2; ++it) sweep();
  CHECK(hipStreamSynchronize(st));
  const int reps = 5; float tp = 0.f, tt = 0.f;
  for (int it = 0; it < reps; ++it) {
    CHECK(hipEventRecord(e0, st)); sweep(); CHECK(hipEventRecord(e2, st)); CHECK(hipEventSynchronize(e2));
    float f1, f2; CHECK(hipEventElapsedTime(&f1, e0, e1)); CHECK(hipEventElapsedTime(&f2, e1, e2)); tp += f1; tt += f2;
  }
  printf("forward sweep: pre-pass %.3f ms + %d tile launches %.3f ms = %.2f us per launch, %.3f us per dependency level\n",
         tp / reps, ngrp, tt / reps, 1e3 * tt / reps / ngrp, 1e3 * tt / reps / nlev);
  for (int stg = 1; stg <= 2; ++stg) {
    stage = stg;
    sweep(); CHECK(hipStreamSynchronize(st));
    float ts = 0.f;
    for (int it = 0; it < reps; ++it) {
      CHECK(hipEventRecord(e0, st)); sweep(); CHECK(hipEventRecord(e2, st)); CHECK(hipEventSynchronize(e2));
      float f2; CHECK(hipEventElapsedTime(&f2, e1, e2)); ts += f2;
    }
    printf("   ablation, %s: %.2f us per launch\n", stg == 1 ? "operand loads only (no sub-level loop)" : "sub-level loop only (no operand loads)",
           1e3 * ts / reps / ngrp);
  }
  return 0;
}
