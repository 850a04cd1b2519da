// gs_step_bench — what one merged-group launch of the Gauss-Seidel sweep costs, and why.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gs_step_bench tools/gs_step_bench.hip
// A synthetic chain of K dependent "group" launches with the shape of the 256^3 fine-level sweep: launch k updates
// rows [kR, (k+1)R) from slot-packed composite rows (LEN entries each) whose columns point into the rows of launch
// k-1 (same relative position +- a window), exactly the access pattern of gs_slot_kernel.  Variants:
//   T / EPT   workgroup size and entries per thread (slot = T * EPT entries)
//   STAGE     0 full kernel | 1 no x gather | 2 matrix loads only (no gather, no LDS, one store per row)
//   eager launches vs ONE hipGraph replay of the whole chain (is the boundary host- or device-bound?)
// Prints microseconds per launch.  Every variant checks its result against the T512/EPT1 run (same arithmetic order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <type_traits>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));
typedef int i4_t __attribute__((ext_vector_type(4)));

struct Args {
  const int32_t* wcol; const double* wval; const int32_t* slot_row; const i4_t* wmeta; const double* diag; const double* bp;
  double* x; long long ent0; int nslots; int slot0; int xcd;
  const i4_t* smeta; const double* sdiag; int rmax;   // in-slot row descriptors (round 5): per slot rmax x {start, end, row, 0} and the diagonals, found from blockIdx alone
};

__device__ __forceinline__ int xcd_block(int b, int nb) { const int per = (nb + 7) / 8; return (b % 8) * per + b / 8; }

template <int T, int EPT, int STAGE>
__global__ __launch_bounds__(T) void slot_kernel(Args a) {
  constexpr int S = T * EPT;
  __shared__ double s_prod[S];
  const int tid = threadIdx.x;
  int lb = blockIdx.x;
  if (a.xcd) lb = xcd_block(lb, a.nslots);
  if (lb >= a.nslots) return;
  const int s = a.slot0 + lb;
  const long long base = a.ent0 + (long long)lb * S;
  double v[EPT]; int c[EPT];
  if (EPT == 1) { v[0] = a.wval[base + tid]; c[0] = a.wcol[base + tid]; }
  else if (EPT == 2) {
    const d2_t vv = *(const d2_t*)(a.wval + base + 2 * tid); const i2_t cc = *(const i2_t*)(a.wcol + base + 2 * tid);
    v[0] = vv.x; v[1] = vv.y; c[0] = cc.x; c[1] = cc.y;
  } else {
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) {
      const long long o = base + 4 * (tid + q * T);
      const d2_t v0 = *(const d2_t*)(a.wval + o), v1 = *(const d2_t*)(a.wval + o + 2); const i4_t cc = *(const i4_t*)(a.wcol + o);
      v[4 * q] = v0.x; v[4 * q + 1] = v0.y; v[4 * q + 2] = v1.x; v[4 * q + 3] = v1.y;
      c[4 * q] = cc.x; c[4 * q + 1] = cc.y; c[4 * q + 2] = cc.z; c[4 * q + 3] = cc.w;
    }
  }
  const int r0 = a.slot_row[2 * s], r1 = a.slot_row[2 * s + 1];
  double xv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) xv[e] = (STAGE == 0) ? a.x[c[e]] : 1.0;
  const int nrows = r1 - r0;
  i4_t m = i4_t{0, 0, -1, 0}; double d = 0.0, bb = 0.0;
  if (tid < nrows) { m = a.wmeta[r0 + tid]; d = a.diag[r0 + tid]; bb = a.bp[r0 + tid]; }
  if (STAGE == 2) {
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc += v[e] * (double)c[e];
    if (tid < nrows) a.x[r0 + tid] = (bb - acc) / d;
    return;
  }
  // entry position of element e of this thread inside the slot
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int pos = (EPT <= 2) ? EPT * tid + e : 4 * (tid + (e / 4) * T) + (e & 3);
    s_prod[pos] = v[e] * xv[e];
  }
  __syncthreads();
  for (int t = tid; t < nrows; t += T) {
    if (t != tid) { m = a.wmeta[r0 + t]; d = a.diag[r0 + t]; bb = a.bp[r0 + t]; }
    double acc = 0.0;
    const int lo = (int)(m.x - base), hi = (int)(m.y - base);
    for (int j = lo; j < hi; ++j) acc += s_prod[j];
    if (d != 0.0) a.x[r0 + t] = (bb - acc) / d;
  }
}

// The slot kernel with the row descriptors INSIDE the slot (round-5 verdict, task 1a): the launch's first round trip carries
// the (col, val) entries AND the rows' {start, end, row} + diagonals (their addresses follow from blockIdx), the second one
// only the x gathers and b — instead of slot_row first, then wmeta / diag / b behind it.
template <int T, int EPT>
__global__ __launch_bounds__(T) void slot_inslot_kernel(Args a) {
  constexpr int S = T * EPT;
  __shared__ double s_prod[S];
  const int tid = threadIdx.x;
  int lb = blockIdx.x;
  if (a.xcd) lb = xcd_block(lb, a.nslots);
  if (lb >= a.nslots) return;
  const int s = a.slot0 + lb;
  const long long base = a.ent0 + (long long)lb * S;
  double v[EPT]; int c[EPT];
  if (EPT == 1) { v[0] = a.wval[base + tid]; c[0] = a.wcol[base + tid]; }
  else {
    const d2_t vv = *(const d2_t*)(a.wval + base + 2 * tid); const i2_t cc = *(const i2_t*)(a.wcol + base + 2 * tid);
    v[0] = vv.x; v[EPT - 1] = vv.y; c[0] = cc.x; c[EPT - 1] = cc.y;
  }
  i4_t m = i4_t{0, 0, -1, 0}; double d = 0.0;
  if (tid < a.rmax) { m = a.smeta[(long long)s * a.rmax + tid]; d = a.sdiag[(long long)s * a.rmax + tid]; }
  double xv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) xv[e] = a.x[c[e]];
  const double bb = m.z >= 0 ? a.bp[m.z] : 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) s_prod[EPT * tid + e] = v[e] * xv[e];
  __syncthreads();
  if (m.z >= 0) {
    double acc = 0.0;
    const int lo = (int)(m.x - base), hi = (int)(m.y - base);
    for (int j = lo; j < hi; ++j) acc += s_prod[j];
    if (d != 0.0) a.x[m.z] = (bb - acc) / d;
  }
}

struct Layout { std::vector<int32_t> wcol, slot_row; std::vector<double> wval; std::vector<i4_t> wmeta; std::vector<int> slot0, nslots; std::vector<long long> ent0; };
// SELL-like layout: a wave owns 64 / KL consecutive rows, KL lanes per row; iteration t, lane l reads entry t * KL + (l % KL)
// of row l / KL from position chunk_base + t * 64 + l (fully coalesced); padding entries have col = -1.
struct SellArgs { const int32_t* col; const double* val; const i2_t* chunk; const double* diag; const double* bp; double* x; int row0; int nrows; int chunk0; int nchunks; int xcd; };
template <int KL>
__global__ __launch_bounds__(256) void sell_kernel(SellArgs a) {
  constexpr int C = 64 / KL;
  int wb = blockIdx.x;
  const int nwg = (a.nchunks + 3) / 4;
  if (a.xcd) wb = xcd_block(wb, nwg);
  if (wb >= nwg) return;
  const int ch = wb * 4 + (threadIdx.x >> 6);
  if (ch >= a.nchunks) return;
  const int lane = threadIdx.x & 63;
  const i2_t cd = a.chunk[a.chunk0 + ch];   // {offset in units of 64 entries, iterations}
  const long long base = (long long)cd.x * 64 + lane;
  const int r = ch * C + lane / KL;
  const bool live = r < a.nrows;
  double d = 1.0, bb = 0.0;
  if (live) { d = a.diag[a.row0 + r]; bb = a.bp[a.row0 + r]; }
  double acc = 0.0;
  int t = 0;
  for (; t + 4 <= cd.y; t += 4) {
    double v[4]; int c[4]; double xv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a.val[base + (long long)(t + e) * 64]; c[e] = a.col[base + (long long)(t + e) * 64]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) xv[e] = c[e] >= 0 ? a.x[c[e]] : 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (c[e] >= 0) acc += v[e] * xv[e];
  }
  for (; t < cd.y; ++t) {
    const double v = a.val[base + (long long)t * 64]; const int c = a.col[base + (long long)t * 64];
    if (c >= 0) acc += v * a.x[c];
  }
#pragma unroll
  for (int w = 1; w < KL; w <<= 1) acc += __shfl_xor(acc, w, 64);
  if (live && (lane % KL) == 0 && d != 0.0) a.x[a.row0 + r] = (bb - acc) / d;
}
// the same with EVERY load of the chunk in flight at once (T <= TMAX iterations, fully unrolled): one round trip for
// col/val, one for the gathers, whatever the row length.  UNI: the iteration count comes from the kernel arguments
// (uniform per launch) instead of the per-chunk descriptor: no dependent scalar load in front of the stream.
template <int KL, int TMAX, bool UNI>
__global__ __launch_bounds__(256) void sell_full_kernel(SellArgs a, int Tuni) {
  constexpr int C = 64 / KL;
  int wb = blockIdx.x;
  const int nwg = (a.nchunks + 3) / 4;
  if (a.xcd) wb = xcd_block(wb, nwg);
  if (wb >= nwg) return;
  const int ch = wb * 4 + (threadIdx.x >> 6);
  if (ch >= a.nchunks) return;
  const int lane = threadIdx.x & 63;
  long long base; int T;
  if (UNI) { T = Tuni; base = ((long long)a.chunk0 + ch) * Tuni * 64 + lane; }  // chunk0 = first chunk of the launch, all chunks T wide
  else { const i2_t cd = a.chunk[a.chunk0 + ch]; base = (long long)cd.x * 64 + lane; T = cd.y; }
  const int r = ch * C + lane / KL;
  const bool live = r < a.nrows;
  double v[TMAX]; int c[TMAX]; double xv[TMAX];
#pragma unroll
  for (int e = 0; e < TMAX; ++e) { c[e] = -1; if (e < T) c[e] = a.col[base + (long long)e * 64]; }
#pragma unroll
  for (int e = 0; e < TMAX; ++e) { v[e] = 0.0; if (e < T) v[e] = a.val[base + (long long)e * 64]; }
  double d = 1.0, bb = 0.0;
  if (live) { d = a.diag[a.row0 + r]; bb = a.bp[a.row0 + r]; }
#pragma unroll
  for (int e = 0; e < TMAX; ++e) xv[e] = c[e] >= 0 ? a.x[c[e]] : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int e = 0; e < TMAX; ++e) if (c[e] >= 0) acc += v[e] * xv[e];
#pragma unroll
  for (int w = 1; w < KL; w <<= 1) acc += __shfl_xor(acc, w, 64);
  if (live && (lane % KL) == 0 && d != 0.0) a.x[a.row0 + r] = (bb - acc) / d;
}
struct Sell { std::vector<int32_t> col; std::vector<double> val; std::vector<i2_t> chunk; std::vector<int> chunk0, nchunks; };
// from the S=512 slot layout (rows of LEN entries, launch k = rows [kR, (k+1)R))
Sell build_sell(const Layout& L, int K, int R, int LEN, int KL) {
  Sell S; const int C = 64 / KL; const int T = (LEN + KL - 1) / KL;
  for (int k = 0; k < K; ++k) {
    S.chunk0.push_back((int)S.chunk.size());
    const int nch = (R + C - 1) / C; S.nchunks.push_back(nch);
    for (int ch = 0; ch < nch; ++ch) {
      const size_t off = S.col.size(); S.chunk.push_back(i2_t{(int)(off / 64), T});
      S.col.resize(off + (size_t)T * 64, -1); S.val.resize(off + (size_t)T * 64, 0.0);
      for (int l = 0; l < 64; ++l) {
        const int r = ch * C + l / KL; if (r >= R) continue;
        const i4_t m = L.wmeta[(size_t)k * R + r];
        for (int t = 0; t < T; ++t) { const int e = t * KL + (l % KL); if (e < LEN) { S.col[off + (size_t)t * 64 + l] = L.wcol[m.x + e]; S.val[off + (size_t)t * 64 + l] = L.wval[m.x + e]; } }
      }
    }
  }
  return S;
}

// ---- overlapped launches: consecutive groups on TWO streams, ordered by device-side counters instead of the kernel
// boundary.  Group k's workgroups load their matrix entries and row data (independent of x) as soon as they are
// resident — while group k-1 is still running — then wait until every workgroup of group k-1 has arrived on its
// counters, gather x with agent-scope (sc1) loads, finish, store x write-through (sc1) and arrive.  Both grids are
// capped so that two groups are always co-resident (no deadlock whatever the dispatch order); spins are bounded.
constexpr int kNCnt = 64;
struct DepArgs { Args a; unsigned* cnt_prev; unsigned* cnt_cur; unsigned target_prev; int* err; };
__device__ __forceinline__ double ld_agent(const double* p) {
  unsigned long long u = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int T, int EPT>
__global__ __launch_bounds__(T) void slot_dep_kernel(DepArgs d) {
  constexpr int S = T * EPT;
  __shared__ double s_prod[S];
  __shared__ int s_ok;
  const Args& a = d.a;
  const int tid = threadIdx.x;
  bool waited = false;
  for (int lb0 = blockIdx.x; lb0 < a.nslots; lb0 += gridDim.x) {
    const int lb = lb0;
    const int s = a.slot0 + lb;
    const long long base = a.ent0 + (long long)lb * S;
    double v[EPT]; int c[EPT];
    if (EPT == 1) { v[0] = a.wval[base + tid]; c[0] = a.wcol[base + tid]; }
    else if (EPT == 2) {
      const d2_t vv = *(const d2_t*)(a.wval + base + 2 * tid); const i2_t cc = *(const i2_t*)(a.wcol + base + 2 * tid);
      v[0] = vv.x; v[1] = vv.y; c[0] = cc.x; c[1] = cc.y;
    } else {
#pragma unroll
      for (int q = 0; q < EPT / 4; ++q) {
        const long long o = base + 4 * (tid + q * T);
        const d2_t v0 = *(const d2_t*)(a.wval + o), v1 = *(const d2_t*)(a.wval + o + 2); const i4_t cc = *(const i4_t*)(a.wcol + o);
        v[4 * q] = v0.x; v[4 * q + 1] = v0.y; v[4 * q + 2] = v1.x; v[4 * q + 3] = v1.y;
        c[4 * q] = cc.x; c[4 * q + 1] = cc.y; c[4 * q + 2] = cc.z; c[4 * q + 3] = cc.w;
      }
    }
    const int r0 = a.slot_row[2 * s], r1 = a.slot_row[2 * s + 1];
    const int nrows = r1 - r0;
    i4_t m = i4_t{0, 0, -1, 0}; double dg = 0.0, bb = 0.0;
    if (tid < nrows) { m = a.wmeta[r0 + tid]; dg = a.diag[r0 + tid]; bb = a.bp[r0 + tid]; }
    if (!waited) {  // everything above is in flight while we wait for the previous group
      waited = true;
      if (d.cnt_prev) {
        if (tid < 64) {
          int spins = 0; bool ok = true;
          for (;;) {
            unsigned cval = tid < kNCnt ? __hip_atomic_load(d.cnt_prev + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
            for (int w = 1; w < 64; w <<= 1) cval += __shfl_xor(cval, w, 64);
            if (cval >= d.target_prev) break;
            if (++spins > (1 << 18)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
          }
          if (tid == 0) { s_ok = ok; if (!ok) *d.err = 1; }
        }
        __syncthreads();
      }
    }
    double xv[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) xv[e] = ld_agent(a.x + c[e]);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int pos = (EPT <= 2) ? EPT * tid + e : 4 * (tid + (e / 4) * T) + (e & 3);
      s_prod[pos] = v[e] * xv[e];
    }
    __syncthreads();
    for (int t = tid; t < nrows; t += T) {
      if (t != tid) { m = a.wmeta[r0 + t]; dg = a.diag[r0 + t]; bb = a.bp[r0 + t]; }
      double acc = 0.0;
      const int lo = (int)(m.x - base), hi = (int)(m.y - base);
      for (int j = lo; j < hi; ++j) acc += s_prod[j];
      if (dg != 0.0) st_agent(a.x + r0 + t, (bb - acc) / dg);
    }
    __syncthreads();  // s_prod is reused by the next slot of this workgroup
  }
  // arrive: every storing wave drains its write-through stores, then ONE lane counts the workgroup in
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(d.cnt_cur + (blockIdx.x % kNCnt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class Tv> Tv* up(const std::vector<Tv>& v);
// ---- ONE persistent launch for the whole chain, groups separated by an XCD-hierarchical grid barrier (the primitive the
// guide prices at 4.1 / 5.9 / 9.7 us for 256 / 512 / 1024 workgroups): arrivals are counted per XCC (s_getreg XCC_ID, no
// placement assumption), the last arriver of an XCC releases that XCD's L2 once and arrives on the top counter, sees all
// XCCs there, acquires and publishes the XCC's generation; everybody else polls its XCC's generation word (relaxed,
// s_sleep) and then invalidates its L1 once.  The matrix entries of the NEXT group's first slot are loaded before the
// barrier (they do not depend on x).  Every spin is bounded.
struct XcdBar { unsigned xcc_cnt[8][16]; unsigned xcc_gen[8][16]; unsigned top[16]; unsigned census[8][16]; unsigned start[16]; };
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u; }
__device__ __forceinline__ bool spin_until(unsigned* p, unsigned want, int* err) {
  int spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 1023) == 1023 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;  // somebody timed out: unwind
    if (++spins > (1 << 19)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
  }
  return true;
}
__device__ __forceinline__ void xcd_barrier(XcdBar* b, unsigned xcc, unsigned nwg_xcc, unsigned nxcc, unsigned step, int* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's x stores are in its XCD's L2
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(&b->xcc_cnt[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nwg_xcc * (step + 1) - 1) {   // last arriver of this XCC: the XCD leader of this step
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      spin_until(&b->top[0], nxcc * (step + 1), err);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&b->xcc_gen[xcc][0], step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      spin_until(&b->xcc_gen[xcc][0], step + 1, err);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
}
struct PersistArgs { Args a; const long long* ent0; const int* nslots; const int* slot0; int K; XcdBar* bar; int* err; };
template <int T>
__global__ __launch_bounds__(T) void slot_persistent_kernel(PersistArgs pa) {
  __shared__ double s_prod[T];
  __shared__ unsigned s_info[2];
  const Args& a = pa.a;
  const int tid = threadIdx.x;
  XcdBar* bar = pa.bar;
  const unsigned xcc = xcc_id();
  // census: how many workgroups run on each XCC (dispatch placement is not ours to assume), then one plain barrier
  if (tid == 0) {
    __hip_atomic_fetch_add(&bar->census[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&bar->start[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin_until(&bar->start[0], gridDim.x, pa.err);
    unsigned nx = 0;
    for (int q = 0; q < 8; ++q) nx += __hip_atomic_load(&bar->census[q][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    s_info[0] = __hip_atomic_load(&bar->census[xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_info[1] = nx;
  }
  __syncthreads();
  const unsigned nwg_xcc = s_info[0], nxcc = s_info[1];
  // matrix entries of my first slot of group 0
  double v = 0.0; int c = 0;
  if ((int)blockIdx.x < pa.nslots[0]) { const long long base = pa.ent0[0] + (long long)blockIdx.x * T; v = a.wval[base + tid]; c = a.wcol[base + tid]; }
  for (int k = 0; k < pa.K; ++k) {
    const int ns = pa.nslots[k];
    for (int lb = blockIdx.x; lb < ns; lb += gridDim.x) {
      const int s = pa.slot0[k] + lb;
      const long long base = pa.ent0[k] + (long long)lb * T;
      if (lb != (int)blockIdx.x) { v = a.wval[base + tid]; c = a.wcol[base + tid]; }   // (first slot: prefetched before the barrier)
      const int r0 = a.slot_row[2 * s], nrows = a.slot_row[2 * s + 1] - r0;
      const double xv = a.x[c];
      i4_t m = i4_t{0, 0, -1, 0}; double dg = 0.0, bb = 0.0;
      if (tid < nrows) { m = a.wmeta[r0 + tid]; dg = a.diag[r0 + tid]; bb = a.bp[r0 + tid]; }
      s_prod[tid] = v * xv;
      __syncthreads();
      if (tid < nrows) {
        double acc = 0.0;
        for (int j = (int)(m.x - base); j < (int)(m.y - base); ++j) acc += s_prod[j];
        if (dg != 0.0) a.x[r0 + tid] = (bb - acc) / dg;
      }
      __syncthreads();
    }
    if (k + 1 < pa.K) {
      // next group's first slot: independent of x, in flight across the barrier
      if ((int)blockIdx.x < pa.nslots[k + 1]) {
        const long long base = pa.ent0[k + 1] + (long long)blockIdx.x * T;
        v = a.wval[base + tid]; c = a.wcol[base + tid];
      }
      xcd_barrier(bar, xcc, nwg_xcc, nxcc, (unsigned)k, pa.err);
    }
  }
}
double run_persistent(const Layout& L, Args a, int K, int grid, hipStream_t st, std::vector<double>* out, long long n,
                      const std::vector<double>& x0, int* herr) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  XcdBar* bar; int* err;
  CHECK(hipMalloc(&bar, sizeof(XcdBar))); CHECK(hipMalloc(&err, 4));
  long long* d_ent0 = up(L.ent0); int* d_ns = up(L.nslots); int* d_s0 = up(L.slot0);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemcpyAsync(a.x, x0.data(), 8 * n, hipMemcpyHostToDevice, st));
    CHECK(hipMemsetAsync(bar, 0, sizeof(XcdBar), st)); CHECK(hipMemsetAsync(err, 0, 4, st));
    CHECK(hipStreamSynchronize(st));
    PersistArgs pa{}; pa.a = a; pa.a.xcd = 0; pa.ent0 = d_ent0; pa.nslots = d_ns; pa.slot0 = d_s0; pa.K = K; pa.bar = bar; pa.err = err;
    CHECK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((slot_persistent_kernel<512>), dim3(grid), dim3(512), 0, st, pa);
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  CHECK(hipMemcpy(herr, err, 4, hipMemcpyDeviceToHost));
  if (out) { out->resize(n); CHECK(hipMemcpy(out->data(), a.x, 8 * n, hipMemcpyDeviceToHost)); }
  CHECK(hipFree(bar)); CHECK(hipFree(err)); CHECK(hipFree(d_ent0)); CHECK(hipFree(d_ns)); CHECK(hipFree(d_s0));
  return 1e3 * best / K;
}

template <int T, int EPT>
double run_dep(const Layout& L, Args a, int K, int cap, bool two_streams, hipStream_t sa, hipStream_t sb, std::vector<double>* out, long long n,
               const std::vector<double>& x0, int* herr) {
  hipEvent_t e0, e1, eb; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&eb));
  unsigned* cnt; int* err;
  CHECK(hipMalloc(&cnt, sizeof(unsigned) * kNCnt * (K + 1))); CHECK(hipMalloc(&err, 4));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemcpyAsync(a.x, x0.data(), 8 * n, hipMemcpyHostToDevice, sa));
    CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned) * kNCnt * (K + 1), sa)); CHECK(hipMemsetAsync(err, 0, 4, sa));
    CHECK(hipStreamSynchronize(sa)); CHECK(hipStreamSynchronize(sb));
    CHECK(hipEventRecord(e0, sa));
    unsigned prev_grid = 0;
    for (int k = 0; k < K; ++k) {
      DepArgs d{}; d.a = a; d.a.ent0 = L.ent0[k]; d.a.nslots = L.nslots[k]; d.a.slot0 = L.slot0[k]; d.a.xcd = 0;
      const int grid = std::min(L.nslots[k], cap);
      d.cnt_prev = k ? cnt + (size_t)kNCnt * (k - 1) : nullptr; d.cnt_cur = cnt + (size_t)kNCnt * k; d.target_prev = prev_grid; d.err = err;
      hipLaunchKernelGGL((slot_dep_kernel<T, EPT>), dim3(grid), dim3(T), 0, (two_streams && (k & 1)) ? sb : sa, d);
      prev_grid = (unsigned)grid;
    }
    CHECK(hipEventRecord(eb, sb)); CHECK(hipStreamWaitEvent(sa, eb, 0));
    CHECK(hipEventRecord(e1, sa)); CHECK(hipStreamSynchronize(sa)); CHECK(hipStreamSynchronize(sb));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  CHECK(hipMemcpy(herr, err, 4, hipMemcpyDeviceToHost));
  if (out) { out->resize(n); CHECK(hipMemcpy(out->data(), a.x, 8 * n, hipMemcpyDeviceToHost)); }
  CHECK(hipFree(cnt)); CHECK(hipFree(err));
  return 1e3 * best / K;
}


// ---- software-pipelined slot walk (round-3 verdict, task 1A): the grid is the RESIDENT capacity (G workgroups), workgroup
// b walks slots b, b + G, b + 2G, ... of ONE launch with register double buffering: the col / val loads of slot k+2 and the
// x gather of slot k+1 are issued before the LDS row sums of slot k (two s_prod buffers: one barrier per slot).
template <int T>
__global__ __launch_bounds__(T) void slot_pipe_kernel(Args a) {
  __shared__ double s_prod[2][T];
  const int tid = threadIdx.x;
  const int G = gridDim.x;
  int lb = blockIdx.x;
  if (lb >= a.nslots) return;
  // prologue: slot 0's entries, slot 1's entries, slot 0's gather
  long long base0 = a.ent0 + (long long)lb * T;
  double v0 = a.wval[base0 + tid]; int c0 = a.wcol[base0 + tid];
  int r00 = a.slot_row[2 * (a.slot0 + lb)], r01 = a.slot_row[2 * (a.slot0 + lb) + 1];
  double v1 = 0.0; int c1 = 0; int r10 = 0, r11 = 0; long long base1 = 0;
  const bool has1 = lb + G < a.nslots;
  if (has1) { base1 = a.ent0 + (long long)(lb + G) * T; v1 = a.wval[base1 + tid]; c1 = a.wcol[base1 + tid];
              r10 = a.slot_row[2 * (a.slot0 + lb + G)]; r11 = a.slot_row[2 * (a.slot0 + lb + G) + 1]; }
  double x0 = a.x[c0];
  i4_t m0 = i4_t{0, 0, -1, 0}; double d0 = 0.0, b0 = 0.0;
  if (tid < r01 - r00) { m0 = a.wmeta[r00 + tid]; d0 = a.diag[r00 + tid]; b0 = a.bp[r00 + tid]; }
  int buf = 0;
  for (;; lb += G) {
    // issue: entries of slot lb + 2G, gather + row data of slot lb + G
    double v2 = 0.0; int c2 = 0; int r20 = 0, r21 = 0; long long base2 = 0;
    const bool cur1 = lb + G < a.nslots, has2 = lb + 2 * G < a.nslots;
    if (has2) { base2 = a.ent0 + (long long)(lb + 2 * G) * T; v2 = a.wval[base2 + tid]; c2 = a.wcol[base2 + tid];
                r20 = a.slot_row[2 * (a.slot0 + lb + 2 * G)]; r21 = a.slot_row[2 * (a.slot0 + lb + 2 * G) + 1]; }
    double x1 = 0.0; i4_t m1 = i4_t{0, 0, -1, 0}; double d1 = 0.0, b1 = 0.0;
    if (cur1) { x1 = a.x[c1]; if (tid < r11 - r10) { m1 = a.wmeta[r10 + tid]; d1 = a.diag[r10 + tid]; b1 = a.bp[r10 + tid]; } }
    // finish slot lb
    s_prod[buf][tid] = v0 * x0;
    __syncthreads();
    const int nrows = r01 - r00;
    if (tid < nrows) {
      double acc = 0.0;
      const int lo = (int)(m0.x - base0), hi = (int)(m0.y - base0);
      for (int j = lo; j < hi; ++j) acc += s_prod[buf][j];
      if (d0 != 0.0) a.x[r00 + tid] = (b0 - acc) / d0;
    }
    if (!cur1) break;
    buf ^= 1;
    v0 = v1; c0 = c1; x0 = x1; m0 = m1; d0 = d1; b0 = b1; r00 = r10; r01 = r11; base0 = base1;
    v1 = v2; c1 = c2; r10 = r20; r11 = r21; base1 = base2;
  }
}
template <int T>
double run_pipe(const Layout& L, Args a, int K, int G, hipStream_t st, std::vector<double>* out, long long n, const std::vector<double>& x0) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemcpyAsync(a.x, x0.data(), 8 * n, hipMemcpyHostToDevice, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int k = 0; k < K; ++k) {
      Args b = a; b.ent0 = L.ent0[k]; b.nslots = L.nslots[k]; b.slot0 = L.slot0[k]; b.xcd = 0;
      hipLaunchKernelGGL((slot_pipe_kernel<T>), dim3(std::min(G, b.nslots)), dim3(T), 0, st, b);
    }
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  if (out) { out->resize(n); CHECK(hipMemcpy(out->data(), a.x, 8 * n, hipMemcpyDeviceToHost)); }
  return 1e3 * best / K;
}

__global__ void empty_kernel(double* x) { if (x == nullptr) x[0] = 1.0; }


// K launches of R rows of LEN entries each, slot size S
int g_pattern = 0;  // 0: random +-1024 window, 1: stencil (fixed offsets, shifted by one per row)
Layout build(int K, int R, int LEN, int S, long long n) {
  Layout L;
  const int rps = S / LEN;  // rows per slot
  const int spl = (R + rps - 1) / rps;
  L.wcol.assign((size_t)K * spl * S, 0); L.wval.assign((size_t)K * spl * S, 0.0); L.wmeta.resize((size_t)K * R);
  uint64_t st = 12345;
  auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 33); };
  for (int k = 0; k < K; ++k) {
    L.slot0.push_back((int)(L.slot_row.size() / 2)); L.nslots.push_back(spl); L.ent0.push_back((long long)k * spl * S);
    for (int q = 0; q < spl; ++q) {
      const int ra = q * rps, rb = std::min(R, ra + rps);
      L.slot_row.push_back(k * R + ra); L.slot_row.push_back(k * R + rb);
      for (int r = ra; r < rb; ++r) {
        const long long start = L.ent0[k] + (long long)q * S + (long long)(r - ra) * LEN;
        L.wmeta[(size_t)k * R + r] = i4_t{(int)start, (int)(start + LEN), -1, 0};
        for (int j = 0; j < LEN; ++j) {
          long long c = (long long)(k == 0 ? K : k - 1) * R + r + (long long)(rnd() % 2048) - 1024;  // previous launch's rows, near r
          if (g_pattern == 1) c = (long long)(k - 1) * R + r + (long long)(j - LEN / 2) * 257 + (j % 3) - 1;
          if (k == 0) c = (long long)K * R + (rnd() % R);
          if (c < 0) c = 0; if (c >= n) c = n - 1;
          if (k > 0) { const long long lo = (long long)(k - 1) * R, hi = (long long)k * R - 1; if (c < lo) c = lo; if (c > hi) c = hi; }
          L.wcol[start + j] = (int32_t)c; L.wval[start + j] = -1.0 / (LEN + 1) * (1.0 + 1e-3 * (rnd() % 7));
        }
      }
    }
  }
  return L;
}

template <int T, int EPT, int STAGE>
double run(const Layout& L, Args a, int K, bool graph, int xcd, hipStream_t st, std::vector<double>* out, long long n, const std::vector<double>& x0, int reps = 3) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  hipGraphExec_t exec = nullptr;
  auto enqueue = [&]() {
    for (int k = 0; k < K; ++k) {
      Args b = a; b.ent0 = L.ent0[k]; b.nslots = L.nslots[k]; b.slot0 = L.slot0[k]; b.xcd = xcd;
      const int grid = xcd ? ((b.nslots + 7) / 8) * 8 : b.nslots;
      hipLaunchKernelGGL((slot_kernel<T, EPT, STAGE>), dim3(grid), dim3(T), 0, st, b);
    }
  };
  if (graph) {
    hipGraph_t g; CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enqueue(); CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0)); CHECK(hipGraphDestroy(g));
  }
  for (int rep = 0; rep < reps; ++rep) {
    CHECK(hipMemcpyAsync(a.x, x0.data(), 8 * n, hipMemcpyHostToDevice, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    if (graph) CHECK(hipGraphLaunch(exec, st)); else enqueue();
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  if (exec) CHECK(hipGraphExecDestroy(exec));
  if (out) { out->resize(n); CHECK(hipMemcpy(out->data(), a.x, 8 * n, hipMemcpyDeviceToHost)); }
  return 1e3 * best / K;
}

template <int T, int EPT>
double run_inslot(const Layout& L, Args a, int K, bool graph, hipStream_t st, std::vector<double>* out, long long n, const std::vector<double>& x0, const std::vector<double>& diag) {
  // the in-slot descriptors from the layout's per-row ones
  const size_t nslot = L.slot_row.size() / 2;
  int rmax = 1;
  for (size_t q = 0; q < nslot; ++q) rmax = std::max(rmax, L.slot_row[2 * q + 1] - L.slot_row[2 * q]);
  if (rmax > T) { printf("  in-slot: %d rows in a slot > %d threads\n", rmax, T); return 0.0; }
  std::vector<i4_t> sm(nslot * rmax, i4_t{0, 0, -1, 0}); std::vector<double> sd(nslot * rmax, 0.0);
  for (size_t q = 0; q < nslot; ++q)
    for (int r = L.slot_row[2 * q]; r < L.slot_row[2 * q + 1]; ++r) {
      const i4_t w = L.wmeta[r];
      sm[q * rmax + (r - L.slot_row[2 * q])] = i4_t{w.x, w.y, r, 0}; sd[q * rmax + (r - L.slot_row[2 * q])] = diag[r];
    }
  i4_t* dsm; double* dsd; CHECK(hipMalloc(&dsm, sizeof(i4_t) * sm.size())); CHECK(hipMalloc(&dsd, 8 * sd.size()));
  CHECK(hipMemcpy(dsm, sm.data(), sizeof(i4_t) * sm.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsd, sd.data(), 8 * sd.size(), hipMemcpyHostToDevice));
  a.smeta = dsm; a.sdiag = dsd; a.rmax = rmax;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  hipGraphExec_t exec = nullptr;
  auto enqueue = [&]() {
    for (int k = 0; k < K; ++k) {
      Args b = a; b.ent0 = L.ent0[k]; b.nslots = L.nslots[k]; b.slot0 = L.slot0[k]; b.xcd = 1;
      hipLaunchKernelGGL((slot_inslot_kernel<T, EPT>), dim3(((b.nslots + 7) / 8) * 8), dim3(T), 0, st, b);
    }
  };
  if (graph) { hipGraph_t g; CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enqueue(); CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0)); CHECK(hipGraphDestroy(g)); }
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemcpyAsync(a.x, x0.data(), 8 * n, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st)); if (graph) CHECK(hipGraphLaunch(exec, st)); else enqueue(); CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  if (exec) CHECK(hipGraphExecDestroy(exec));
  if (out) { out->resize(n); CHECK(hipMemcpy(out->data(), a.x, 8 * n, hipMemcpyDeviceToHost)); }
  CHECK(hipFree(dsm)); CHECK(hipFree(dsd));
  return 1e3 * best / K;
}

template <int KL, int TMAX = 0, bool UNI = false>
double run_sell(const Sell& S, SellArgs a, int K, int R, bool graph, hipStream_t st, std::vector<double>* out, long long n, const std::vector<double>& x0) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  auto enqueue = [&]() {
    for (int k = 0; k < K; ++k) {
      SellArgs b = a; b.row0 = k * R; b.nrows = R; b.chunk0 = S.chunk0[k]; b.nchunks = S.nchunks[k]; b.xcd = 1;
      const int nwg = (b.nchunks + 3) / 4;
      if constexpr (TMAX == 0) hipLaunchKernelGGL((sell_kernel<KL>), dim3(((nwg + 7) / 8) * 8), dim3(256), 0, st, b);
      else hipLaunchKernelGGL((sell_full_kernel<KL, TMAX, UNI>), dim3(((nwg + 7) / 8) * 8), dim3(256), 0, st, b, (int)S.chunk[0].y);
    }
  };
  hipGraphExec_t exec = nullptr;
  if (graph) { hipGraph_t g; CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enqueue(); CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0)); CHECK(hipGraphDestroy(g)); }
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemcpyAsync(a.x, x0.data(), 8 * n, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st)); if (graph) CHECK(hipGraphLaunch(exec, st)); else enqueue(); CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  if (exec) CHECK(hipGraphExecDestroy(exec));
  if (out) { out->resize(n); CHECK(hipMemcpy(out->data(), a.x, 8 * n, hipMemcpyDeviceToHost)); }
  return 1e3 * best / K;
}

template <class Tv> Tv* up(const std::vector<Tv>& v) { Tv* p; CHECK(hipMalloc(&p, sizeof(Tv) * std::max<size_t>(1, v.size()))); CHECK(hipMemcpy(p, v.data(), sizeof(Tv) * v.size(), hipMemcpyHostToDevice)); return p; }

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 200;
  hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipStream_t st2; CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  // boundary alone: empty kernels, eager vs graph
  {
    double* dummy; CHECK(hipMalloc(&dummy, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wg : {1, 256, 1024}) {
      for (int graph = 0; graph < 2; ++graph) {
        hipGraphExec_t exec = nullptr;
        if (graph) { hipGraph_t g; CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
          for (int k = 0; k < 2000; ++k) hipLaunchKernelGGL(empty_kernel, dim3(wg), dim3(256), 0, st, dummy);
          CHECK(hipStreamEndCapture(st, &g)); CHECK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0)); CHECK(hipGraphDestroy(g)); }
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CHECK(hipStreamSynchronize(st)); CHECK(hipEventRecord(e0, st));
          if (graph) CHECK(hipGraphLaunch(exec, st)); else for (int k = 0; k < 2000; ++k) hipLaunchKernelGGL(empty_kernel, dim3(wg), dim3(256), 0, st, dummy);
          CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        printf("empty kernel x2000, %4d WGs, %s: %.2f us per launch\n", wg, graph ? "graph" : "eager", 1e3 * best / 2000);
        if (exec) CHECK(hipGraphExecDestroy(exec));
      }
    }
  }
  for (int pat = 1; pat >= 1; --pat)
  for (int LEN : {10, 27, 60}) {
    g_pattern = pat;
    for (int R : {6400, 22000, 45000, 65000, 130000}) {
      if (LEN == 60 && R > 22000) continue;
      if (LEN == 100 && R > 6400) continue;
      if (pat == 0 && !(R == 65000 && LEN == 10) && !(R == 22000 && LEN == 27)) continue;
      printf("==== pattern %s\n", pat ? "stencil" : "random window");   // rows per launch: x LEN / 512 = 293 / 508 / 1270 / 2540 slots of 512
      const long long n = (long long)(K + 1) * R;
      std::vector<double> x0(n), diag(n, 1.0), bp(n);
      for (long long i = 0; i < n; ++i) { x0[i] = 1.0 + 1e-3 * (i % 97); bp[i] = 0.5 + 1e-3 * (i % 89); }
      double* dx = up(x0); double* dd = up(diag); double* db = up(bp);
      std::vector<double> ref;
      printf("---- LEN %d, %d rows/launch (%d x 512-entry slots, %.1f MB of col/val per launch), K=%d launches\n", LEN, R, (R + 512 / LEN - 1) / (512 / LEN),
             12.0 * R * LEN / 1e6, K);
      for (int S : {512, 1024}) {
        Layout L = build(K, R, LEN, S, n);
        Args a{}; a.wcol = up(L.wcol); a.wval = up(L.wval); a.slot_row = up(L.slot_row); a.wmeta = up(L.wmeta); a.diag = dd; a.bp = db; a.x = dx;
        auto report = [&](const char* name, double us, const std::vector<double>* got) {
          long long bad = 0;
          if (got && !ref.empty()) for (long long i = 0; i < n; ++i) bad += ((*got)[i] != ref[i]);
          printf("  S=%4d %-22s %6.2f us/launch%s\n", S, name, us, got && !ref.empty() ? (bad ? "  MISMATCH" : "  (= reference)") : "");
        };
        std::vector<double> got;
        if (S == 512) {
          const double t = run<512, 1, 0>(L, a, K, false, 1, st, &ref, n, x0); report("T512 E1 eager xcd", t, nullptr);
          report("T512 E1 graph xcd", run<512, 1, 0>(L, a, K, true, 1, st, &got, n, x0), &got);
          report("T256 E2 eager xcd", run<256, 2, 0>(L, a, K, false, 1, st, &got, n, x0), &got);
          report("T512 E1 IN-SLOT descr.", run_inslot<512, 1>(L, a, K, false, st, &got, n, x0, diag), &got);
          report("T256 E2 IN-SLOT descr.", run_inslot<256, 2>(L, a, K, false, st, &got, n, x0, diag), &got);
          if (getenv("INSLOT_ONLY")) { hipFree((void*)a.wcol); hipFree((void*)a.wval); hipFree((void*)a.slot_row); hipFree((void*)a.wmeta); continue; }
          report("T512 E1 no-gather", run<512, 1, 1>(L, a, K, false, 1, st, nullptr, n, x0), nullptr);
          for (int G : {512, 768, 1024, 2048}) {
            char nm[64]; snprintf(nm, sizeof nm, "PIPE T512 grid %d", G);
            report(nm, run_pipe<512>(L, a, K, G, st, &got, n, x0), &got);
          }
          for (int G : {1024, 2048}) {
            char nm[64]; snprintf(nm, sizeof nm, "PIPE T256 grid %d (S=512!)", G);
            (void)nm; (void)G;
          }
          if (!getenv("FULL")) { report("T512 E1 loads-only", run<512, 1, 2>(L, a, K, false, 1, st, nullptr, n, x0), nullptr); hipFree((void*)a.wcol); hipFree((void*)a.wval); hipFree((void*)a.slot_row); hipFree((void*)a.wmeta); continue; }
          {
            int herr = 0;
            double t1 = run_dep<512, 1>(L, a, K, 480, false, st, st2, &got, n, x0, &herr);
            printf("  S= 512 dep T512 E1 ONE stream (counters + boundary)   %6.2f us/launch  err %d\n", t1, herr);
            for (int grid : {256, 512, 1024}) {
              char nm[64]; snprintf(nm, sizeof nm, "persistent xcd-barrier %d WG", grid);
              const double tp = run_persistent(L, a, K, grid, st, &got, n, x0, &herr);
              report(nm, tp, &got);
              printf("       (spin timeout flag %d)\n", herr);
            }
            report("dep T512E1 2 streams cap480", run_dep<512, 1>(L, a, K, 480, true, st, st2, &got, n, x0, &herr), &got);
            printf("       (spin timeout flag %d)\n", herr);
            report("dep T256E2 2 streams cap960", run_dep<256, 2>(L, a, K, 960, true, st, st2, &got, n, x0, &herr), &got);
            printf("       (spin timeout flag %d)\n", herr);
            report("dep T256E2 2 streams cap480", run_dep<256, 2>(L, a, K, 480, true, st, st2, &got, n, x0, &herr), &got);
            printf("       (spin timeout flag %d)\n", herr);
          }
          report("T512 E1 loads-only", run<512, 1, 2>(L, a, K, false, 1, st, nullptr, n, x0), nullptr);
          auto sell = [&](auto klc, const char* name) {
            constexpr int KL = decltype(klc)::value;
            Sell Sl = build_sell(L, K, R, LEN, KL);
            SellArgs sa{}; sa.col = up(Sl.col); sa.val = up(Sl.val); sa.chunk = up(Sl.chunk); sa.diag = dd; sa.bp = db; sa.x = dx;
            std::vector<double> g2;
            const double t = run_sell<KL>(Sl, sa, K, R, false, st, &g2, n, x0);
            double err = 0; for (long long i = 0; i < n; ++i) err = std::max(err, std::abs(g2[i] - ref[i]) / (1e-300 + std::abs(ref[i])));
            printf("  SELL %-24s %6.2f us/launch  (max rel diff vs slots %.1e; %.2f x entries stored)\n", name, t, err, (double)Sl.col.size() / ((double)K * R * LEN));
            const int T = (LEN + KL - 1) / KL;
            if (T <= 8) {
              const double t8 = run_sell<KL, 8, false>(Sl, sa, K, R, false, st, &g2, n, x0);
              double e8 = 0; for (long long i = 0; i < n; ++i) e8 = std::max(e8, std::abs(g2[i] - ref[i]) / (1e-300 + std::abs(ref[i])));
              const double t8g = run_sell<KL, 8, false>(Sl, sa, K, R, true, st, nullptr, n, x0);
              const double t8u = run_sell<KL, 8, true>(Sl, sa, K, R, false, st, &g2, n, x0);
              double e8u = 0; for (long long i = 0; i < n; ++i) e8u = std::max(e8u, std::abs(g2[i] - ref[i]) / (1e-300 + std::abs(ref[i])));
              printf("       full unroll T<=8: %6.2f us (diff %.1e)   graph %6.2f us   uniform-T, no descriptor load: %6.2f us (diff %.1e)\n", t8, e8, t8g, t8u, e8u);
            } else if (T <= 16) {
              const double t16 = run_sell<KL, 16, false>(Sl, sa, K, R, false, st, &g2, n, x0);
              double e16 = 0; for (long long i = 0; i < n; ++i) e16 = std::max(e16, std::abs(g2[i] - ref[i]) / (1e-300 + std::abs(ref[i])));
              const double t16u = run_sell<KL, 16, true>(Sl, sa, K, R, false, st, nullptr, n, x0);
              printf("       full unroll T<=16: %6.2f us (diff %.1e)   uniform-T, no descriptor load: %6.2f us\n", t16, e16, t16u);
            }
            hipFree((void*)sa.col); hipFree((void*)sa.val); hipFree((void*)sa.chunk);
          };
          if (getenv("SELL")) {
          sell(std::integral_constant<int, 1>(), "1 lane/row");
          sell(std::integral_constant<int, 2>(), "2 lanes/row");
          sell(std::integral_constant<int, 4>(), "4 lanes/row");
          sell(std::integral_constant<int, 8>(), "8 lanes/row");
          if (LEN >= 27) sell(std::integral_constant<int, 16>(), "16 lanes/row");
          if (LEN >= 100) sell(std::integral_constant<int, 32>(), "32 lanes/row");
          }
        } else if (S == 1024) {
          report("T256 E4 eager xcd", run<256, 4, 0>(L, a, K, false, 1, st, &got, n, x0), &got);
          {
            int herr = 0;
            report("dep T256E4 2 streams cap960", run_dep<256, 4>(L, a, K, 960, true, st, st2, &got, n, x0, &herr), &got);
            printf("       (spin timeout flag %d)\n", herr);
            report("dep T256E4 2 streams cap480", run_dep<256, 4>(L, a, K, 480, true, st, st2, &got, n, x0, &herr), &got);
            printf("       (spin timeout flag %d)\n", herr);
          }
        } else {
          report("T256 E8 eager xcd", run<256, 8, 0>(L, a, K, false, 1, st, &got, n, x0), &got);
        }
        hipFree((void*)a.wcol); hipFree((void*)a.wval); hipFree((void*)a.slot_row); hipFree((void*)a.wmeta);
      }
      hipFree(dx); hipFree(dd); hipFree(db);
    }
  }
  return 0;
}
