// amghip_kernels.hpp — gfx950 (CDNA4, wave64) device kernels of the AMG solve phase.
//
// Everything here is HBM- or latency-bound sparse/stream work (0.135 flop/B on the fine
// Poisson SpMV), so no MFMA: the design rules are coalesced 16-byte streaming of the CSR
// arrays, LDS-staged products with a sequential per-row sum (which also makes the
// result bit-reproducible and equal to a scalar CPU loop), wavefront __shfl
// reductions for norms/dots.  An XCD-contiguous row-block mapping is available as a
// configuration switch; it measured 5-10 % SLOWER than the default round-robin mapping
// on the 256^3 operator (profiles/r01_spmv_variants.log) and is off in the shipped config.
//
// Arithmetic is kept un-contracted (no FMA fusion across the product and the
// running sum: the product is rounded when it is staged in LDS), so a row's sum
// is the same IEEE sequence the reference's scalar loops execute
// (smoother.jl:81-86, SparseArrays mul!).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The arithmetic type of the whole library: Float64 by default; the Float32 instance of the reference's generic code
// (eltype(A) == Float32, test/runtests.jl:244-259) is the SAME source compiled with -DAMGH_REAL=float into its own
// shared object (libamghip_f32.so, same entry points with float in place of double).
#ifndef AMGH_REAL
#define AMGH_REAL double
#endif

namespace amgh {

using real = AMGH_REAL;

constexpr int kWave = 64;
constexpr int kThreads = 256;        // 4 waves / workgroup
constexpr int kNumXcd = 8;

enum StreamMode : int {
  M_SPMV = 0,    // y = A x
  M_RESID = 1,   // y = b - A x            (multilevel.jl:219-220)
  M_ADD = 2,     // y = y + A x            (multilevel.jl:233-234)
  M_JACOBI = 3,  // damped Jacobi sweep    (smoother.jl:113-141)
  M_GS = 4,      // Gauss-Seidel rows of one dependency level (smoother.jl:61-90)
  M_SOR = 5      // SOR rows of one dependency level           (smoother.jl:193-221)
};

struct StreamArgs {
  const int32_t* rowptr;  // CSR of the (possibly level-permuted) matrix
  const int32_t* col;
  const real* val;
  const real* x;     // gather source (length ncols)
  real* y;           // output
  const real* b;     // RESID / smoothers
  const int32_t* dpos; // position of the diagonal entry of each row, -1 if absent
  const real* diag;  // diagonal value of each row (0 if absent)
  const int32_t* perm; // GS/SOR: original row id of permuted row p (nullptr = identity)
  real omega;
  int32_t row_begin;   // rows [row_begin, row_end) of the matrix are processed
  int32_t row_end;
  // multi-RHS (n x bs blocks, multilevel.jl:28-59): ncolv = bs columns of x / y / b at these element strides;
  // the grid is (row tiles padded to a multiple of 8) x ncolv workgroups, see multi_column_block()
  int64_t ldx, ldy, ldb;
  int32_t ncolv;
  // VALUE-CODED columns (CodedCols): entry k as one word, column | code << 24, the code an index into vtab (the operator's
  // distinct values, at most 256 of them: a constant-coefficient stencil has two, its first Galerkin product eight, their
  // interpolation weights a few dozen) — 4 bytes per entry instead of 12; the products and their order are the same
  const uint32_t* ccol;
  const real* vtab;
  int32_t vtab_n;
};
constexpr int kCodeBits = 24, kCodeMax = 256;
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

// LDS index skew: breaks the power-of-two strides of rows with 8/16/32 entries
// (ds_read_b64 banks = (addr/4) mod 64; one pad slot per 32 keeps a 32-lane
// group conflict-free for row lengths 7, 8, 16, 27).
__device__ __forceinline__ int skew(int k) { return k + (k >> 5); }
template <bool SKEWED>
__device__ __forceinline__ int lidx(int k) { return SKEWED ? skew(k) : k; }
__device__ __forceinline__ real seq_sum_skip(const real* s, int lo, int hi, int dp, real acc);

// Optional XCD-contiguous block mapping (workgroup b is observed to run on XCD b % 8):
// give each XCD a contiguous eighth of the row blocks so that the +-nx rows'
// x entries are re-used out of that XCD's own L2.  Speed only, never correctness
// (and measured slower than round-robin here, see the header comment).
// The grid must be launched with 8*ceil(nb/8) workgroups.
__device__ __forceinline__ int xcd_block(int b, int nb) {
  const int per = (nb + kNumXcd - 1) / kNumXcd;
  return (b % kNumXcd) * per + b / kNumXcd;  // may be >= nb: caller bounds-checks
}
// Workgroup -> (tile, right-hand-side column) for a launch over ncolv columns.  Consecutive workgroup ids go to
// consecutive XCDs, so within a group of 8 * ncolv workgroups XCD q gets the ncolv columns of tile 8 * group + q
// back to back: the matrix tile is read from HBM once and served to the other columns by that XCD's L2.
__device__ __forceinline__ void multi_column_block(int ncolv, int& tile, int& column) {
  const int bid = blockIdx.x;
  if (ncolv <= 1) { tile = bid; column = 0; return; }
  const int per = kNumXcd * ncolv;
  const int grp = bid / per, rem = bid - grp * per;
  tile = grp * kNumXcd + (rem & (kNumXcd - 1));
  column = rem / kNumXcd;
}

typedef real d2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));
typedef int i4_t __attribute__((ext_vector_type(4)));

// acc + the value held by the partner lane of step w of an xor butterfly, partners reached through DPP lane permutes
// (no LDS crossbar, no wait): w = 1, 2 inside a quad, w = 4 as the mirror image inside 8 lanes, w = 8 inside 16 — after the
// steps below w the mirrored partner holds the same partial sum as the xor partner, so every lane ends with the sum
// the __shfl_xor tree forms, added in the same order.  Wider steps use __shfl_xor.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int W>
__device__ __forceinline__ real tree_add(real acc) {
#pragma unroll
  for (int w = 1; w < W; w <<= 1) {
    if (w == 1) acc += dpp_move<0xB1>(acc);          // quad_perm [1, 0, 3, 2]
    else if (w == 2) acc += dpp_move<0x4E>(acc);     // quad_perm [2, 3, 0, 1]
    else if (w == 4) acc += dpp_move<0x141>(acc);    // row_half_mirror
    else if (w == 8) acc += dpp_move<0x140>(acc);    // row_mirror
    else acc += __shfl_xor(acc, w, kWave);
  }
  return acc;
}

// matrix streams are read once: optional non-temporal loads (measured neutral-to-negative, off by default)
template <bool NT, class T>
__device__ __forceinline__ T ld_stream(const T* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// Tunables of the stream kernel (chosen by tools/spmv_bench.hip measurements).
template <int THREADS_, int ROWS_, int LDS_NNZ_, int VEC_, bool NT_, bool XCD_>
struct StreamCfg {
  static constexpr int THREADS = THREADS_;  // workgroup size
  static constexpr int ROWS = ROWS_;        // rows per workgroup (may be < THREADS: latency-bound launches)
  static constexpr int RPT = (ROWS_ + THREADS_ - 1) / THREADS_;  // rows per summing thread
  static constexpr int LDS_NNZ = LDS_NNZ_;  // products staged per pass
  static constexpr int VEC = VEC_;          // nonzeros per thread per load (1, 2, 4)
  static constexpr bool NT = NT_;           // non-temporal loads of col/val
  static constexpr bool XCD = XCD_;         // XCD-contiguous block mapping
};

// In-order sum of the staged products s[lo..hi) (indices relative to c0), skipping
// position dp when SKIPD.  The LDS reads of 8 products are issued together; the
// additions stay strictly sequential, so the result is the scalar loop's.
template <bool SKIPD>
__device__ __forceinline__ real seq_sum(const real* s_prod, int lo, int hi, int c0, int dp, real acc) {
  int j = lo;
  for (; j + 8 <= hi; j += 8) {
    real p[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) p[e] = s_prod[skew(j + e - c0)];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (SKIPD) acc = (j + e != dp) ? acc + p[e] : acc;
      else acc += p[e];
    }
  }
  for (; j < hi; ++j) {
    const real p = s_prod[skew(j - c0)];
    if (SKIPD) acc = (j != dp) ? acc + p : acc;
    else acc += p;
  }
  return acc;
}

// "CSR-stream": a workgroup owns ROWS consecutive rows.  Their nonzeros form one
// contiguous range of col/val, loaded with fully coalesced (vector) accesses;
// products val*x[col] are staged in LDS; each thread then sums its own rows'
// segments sequentially in index order.  Row ranges with more products than fit
// in LDS are processed in several passes (any row length is handled).
template <int MODE, class CFG, bool CODED = false>
__global__ __launch_bounds__(CFG::THREADS) void csr_stream_kernel(StreamArgs a) {
  constexpr int T = CFG::THREADS, RPT = CFG::RPT, LDSN = CFG::LDS_NNZ, VEC = CFG::VEC;
  constexpr bool NT = CFG::NT;
  static_assert(!CODED || (VEC == 4 && T >= kCodeMax), "the value-coded variant is the big-operator configuration's");
  __shared__ real s_tab[CODED ? kCodeMax : 1];
  if (CODED) {
    if ((int)threadIdx.x < a.vtab_n) s_tab[threadIdx.x] = a.vtab[threadIdx.x];
    __syncthreads();
  }
  // few rows per workgroup (latency-bound GS launches): plain LDS indices + the pipelined in-order
  // sum; many rows per workgroup (SpMV): skewed indices keep the row-strided reads conflict-free
  constexpr bool SK = CFG::ROWS > 64;
  __shared__ real s_prod[LDSN + (LDSN >> 5) + 2];
  int bid, cv;
  multi_column_block(a.ncolv, bid, cv);
  if (cv > 0) {
    a.x += cv * a.ldx;
    a.y += cv * a.ldy;
    if (a.b) a.b += cv * a.ldb;
  }

  const int nrows = a.row_end - a.row_begin;
  const int nb = (nrows + CFG::ROWS - 1) / CFG::ROWS;
  const int lb = CFG::XCD ? xcd_block(bid, nb) : bid;
  if (lb >= nb) return;
  const int r0 = a.row_begin + lb * CFG::ROWS;
  const int r1 = min(r0 + CFG::ROWS, a.row_end);
  const int tid = threadIdx.x;

  int rs[RPT], re[RPT], dp[RPT], gi[RPT];
  real acc[RPT], gd[RPT], gb[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = r0 + tid + q * T;
    gi[q] = 0; gd[q] = 0.0; gb[q] = 0.0;
    if (tid + q * T < CFG::ROWS && r < r1) {
      rs[q] = a.rowptr[r];
      re[q] = a.rowptr[r + 1];
      dp[q] = (MODE >= M_JACOBI) ? a.dpos[r] : -1;
      if (MODE >= M_GS) {  // issued up front: independent of the products, off the critical path
        gi[q] = a.perm ? a.perm[r] : r;  // x in dependency-level order: row r writes position r
        gd[q] = a.diag[r];
        gb[q] = a.b[r];  // b pre-gathered into dependency-level order
      }
    } else {
      rs[q] = re[q] = 0;
      dp[q] = -1;
    }
    acc[q] = 0.0;
  }
  const int p0 = a.rowptr[r0];
  const int p1 = a.rowptr[r1];

  for (int c0 = p0; c0 < p1; c0 += LDSN) {
    const int c1 = min(c0 + LDSN, p1);
    if (VEC == 1) {
      for (int k = c0 + tid; k < c1; k += T) {
        const real v = ld_stream<NT>(a.val + k);
        const int c = ld_stream<NT>(a.col + k);
        s_prod[lidx<SK>(k - c0)] = v * a.x[c];
      }
    } else {
      const int a0 = c0 & ~(VEC - 1);  // hipMalloc'ed arrays: index multiple of VEC => 16 B aligned
      for (int k = a0 + VEC * tid; k < c1; k += VEC * T) {
        if (k >= c0 && k + VEC <= c1) {
          real v[VEC];
          int c[VEC];
          if (VEC == 2) {
            const d2_t vv = ld_stream<NT>((const d2_t*)(a.val + k));
            const i2_t cc = ld_stream<NT>((const i2_t*)(a.col + k));
            v[0] = vv.x; v[1] = vv.y; c[0] = cc.x; c[1] = cc.y;
          } else if (CODED) {
            const u4_t cc = ld_stream<NT>((const u4_t*)(a.ccol + k));
            c[0] = (int)(cc.x & ((1u << kCodeBits) - 1)); c[1] = (int)(cc.y & ((1u << kCodeBits) - 1));
            c[VEC - 2] = (int)(cc.z & ((1u << kCodeBits) - 1)); c[VEC - 1] = (int)(cc.w & ((1u << kCodeBits) - 1));
            v[0] = s_tab[cc.x >> kCodeBits]; v[1] = s_tab[cc.y >> kCodeBits];
            v[VEC - 2] = s_tab[cc.z >> kCodeBits]; v[VEC - 1] = s_tab[cc.w >> kCodeBits];
          } else {
            const d2_t v0 = ld_stream<NT>((const d2_t*)(a.val + k));
            const d2_t v1 = ld_stream<NT>((const d2_t*)(a.val + k + 2));
            const i4_t cc = ld_stream<NT>((const i4_t*)(a.col + k));
            v[0] = v0.x; v[1] = v0.y; v[VEC - 2] = v1.x; v[VEC - 1] = v1.y;
            c[0] = cc.x; c[1] = cc.y; c[VEC - 2] = cc.z; c[VEC - 1] = cc.w;
          }
          real xv[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) xv[e] = a.x[c[e]];
#pragma unroll
          for (int e = 0; e < VEC; ++e) s_prod[lidx<SK>(k + e - c0)] = v[e] * xv[e];
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const int kk = k + e;
            if (CODED) {
              if (kk >= c0 && kk < c1) { const unsigned w = a.ccol[kk]; s_prod[lidx<SK>(kk - c0)] = s_tab[w >> kCodeBits] * a.x[w & ((1u << kCodeBits) - 1)]; }
            } else
            if (kk >= c0 && kk < c1) s_prod[lidx<SK>(kk - c0)] = a.val[kk] * a.x[a.col[kk]];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int lo = max(rs[q], c0), hi = min(re[q], c1);
      if (SK) acc[q] = seq_sum<(MODE >= M_JACOBI)>(s_prod, lo, hi, c0, dp[q], acc[q]);
      else acc[q] = seq_sum_skip(s_prod, lo - c0, hi - c0, (MODE >= M_JACOBI) ? dp[q] - c0 : -1, acc[q]);
    }
    if (c1 < p1) __syncthreads();
  }

#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = r0 + tid + q * T;
    if (tid + q * T >= CFG::ROWS || r >= r1) continue;
    if (MODE == M_SPMV) {
      a.y[r] = acc[q];
    } else if (MODE == M_RESID) {
      a.y[r] = a.b[r] - acc[q];
    } else if (MODE == M_ADD) {
      a.y[r] = a.y[r] + acc[q];
    } else if (MODE == M_JACOBI) {
      const real d = a.diag[r];
      const real t = a.x[r];
      const real cand = (1.0 - a.omega) * t + a.omega * ((a.b[r] - acc[q]) / d);
      a.y[r] = (d == 0.0) ? t : cand;
    } else {
      const int i = gi[q];
      const real d = gd[q];
      if (d != 0.0) {
        if (MODE == M_GS) {
          a.y[i] = (gb[q] - acc[q]) / d;
        } else {
          a.y[i] = (1.0 - a.omega) * a.y[i] + (a.omega / d) * (gb[q] - acc[q]);
        }
      }
    }
  }
}

// default configuration used by the library (tools/spmv_bench.hip, profiles/r01_spmv_variants.log)
using DefaultCfg = StreamCfg<1024, 1024, 8192, 4, false, false>;

// In-order sum of s[lo..hi) (plain, un-skewed LDS indices): batches of 8 LDS reads with immediate
// offsets from one base address, the next batch in flight while the current one is added — the
// additions themselves stay one strictly sequential chain (the reference's order).
__device__ __forceinline__ real seq_sum_range(const real* s, int lo, int hi, real acc) {
  int j = lo;
  if (hi - j >= 8) {
    real p[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) p[e] = s[j + e];
    j += 8;
    // two register sets ping-pong (no copies): while one batch is being added the other is loading
    while (hi - j >= 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = s[j + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += p[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) p[e] = s[j + 8 + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += q[e];
      j += 16;
    }
    if (hi - j >= 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = s[j + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += p[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += q[e];
      j += 8;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += p[e];
    }
  }
  if (j < hi) {  // tail of 1..7: the reads go out together, the adds stay ordered and conditional
    real r[7];
#pragma unroll
    for (int e = 0; e < 7; ++e) r[e] = s[min(j + e, hi - 1)];
#pragma unroll
    for (int e = 0; e < 7; ++e)
      if (j + e < hi) acc += r[e];
  }
  return acc;
}
// the same, leaving out position dp (the diagonal) when it falls inside [lo, hi)
__device__ __forceinline__ real seq_sum_skip(const real* s, int lo, int hi, int dp, real acc) {
  if (dp >= lo && dp < hi) {
    acc = seq_sum_range(s, lo, dp, acc);
    return seq_sum_range(s, dp + 1, hi, acc);
  }
  return seq_sum_range(s, lo, hi, acc);
}

// bp[r] = b[perm[r]]: right-hand side gathered into dependency-level order once per
// smoother application, so that every level kernel reads it coalesced with no
// perm -> b dependent hop.
__global__ void gather_perm_kernel(const real* __restrict__ b, const int32_t* __restrict__ perm,
                                   real* __restrict__ bp, int n, int64_t ld_src, int64_t ld_dst) {
  b += blockIdx.y * ld_src;  // blockIdx.y = right-hand-side column
  bp += blockIdx.y * ld_dst;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) bp[r] = b[perm[r]];
}
// x[perm[r]] = xp[r]: back from dependency-level order to natural order
__global__ void scatter_perm_kernel(const real* __restrict__ xp, const int32_t* __restrict__ perm,
                                    real* __restrict__ x, int n, int64_t ld_src, int64_t ld_dst) {
  xp += blockIdx.y * ld_src;
  x += blockIdx.y * ld_dst;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) x[perm[r]] = xp[r];
}

// Right-hand side of the NEXT merged sweep when the directions alternate: after a forward sweep
// (D + L) x = s_f holds row by row, so the backward pre-pass b - L x is b - s_f + D x — no matrix pass
// (and symmetrically after a backward sweep).  Level order; blockIdx.y = right-hand-side column.
// (SOR with factor w: diag holds D / w and the formula is b - s_f + (2 - w) (D / w) x, scale = 2 - w.)
__global__ void gs_flip_rhs_kernel(const real* __restrict__ bp, const real* __restrict__ diag,
                                   const real* __restrict__ xp, real* __restrict__ sp, int n, int64_t ldb, int64_t ldx,
                                   real scale) {
  bp += blockIdx.y * ldb;
  xp += blockIdx.y * ldx;
  sp += blockIdx.y * ldx;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
    sp[r] = bp[r] - sp[r] + scale * (diag[r] * xp[r]);
}

// dst[column][0..n) = src[column][0..n): strided copy of a block of right-hand-side columns (blockIdx.y = column)
__global__ void copy_cols_kernel(real* __restrict__ dst, const real* __restrict__ src, int n, int64_t ld_dst,
                                 int64_t ld_src) {
  dst += blockIdx.y * ld_dst;
  src += blockIdx.y * ld_src;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) dst[r] = src[r];
}

// the two kernels above for a block of BS right-hand sides whose sweep also keeps [x ; s] interleaved (gs_slot_il_kernel):
// a thread takes all columns of a row — the same expressions — and writes s to the column-major vector and, as one run of
// BS values, to the interleaved one (sil = position of s[0] there)
template <int BS>
__global__ void gs_flip_rhs_il_kernel(const real* __restrict__ bp, const real* __restrict__ diag, const real* __restrict__ xp,
                                      real* __restrict__ sp, int n, int64_t ldb, int64_t ldx, real scale, real* __restrict__ sil) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    const real d = diag[r];
#pragma unroll
    for (int q = 0; q < BS; ++q) {
      const real v = bp[r + q * ldb] - sp[r + q * ldx] + scale * (d * xp[r + q * ldx]);
      sp[r + q * ldx] = v;
      sil[(int64_t)r * BS + q] = v;
    }
  }
}
template <int BS>
__global__ void copy_cols_il_kernel(real* __restrict__ dst, const real* __restrict__ src, int n, int64_t ld_dst, int64_t ld_src,
                                    real* __restrict__ sil) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
#pragma unroll
    for (int q = 0; q < BS; ++q) {
      const real v = src[r + q * ld_src];
      dst[r + q * ld_dst] = v;
      sil[(int64_t)r * BS + q] = v;
    }
  }
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for outstanding
// global loads (s_waitcnt vmcnt(0)), so prefetches issued before it stay in flight across it.  Only valid where
// the data exchanged between the threads goes through LDS.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Single-workgroup chain over consecutive NARROW dependency levels of a
// Gauss-Seidel/SOR sweep: one thread per row, a workgroup barrier between
// dependency levels (visibility of x inside one CU needs only the barrier).
// Used for whole small hierarchy levels and for the thin head/tail of a large
// level's wavefront, where a kernel boundary per dependency level (~1.5 us)
// would dominate.
struct ChainArgs {
  const int32_t* col;      // level-permuted matrix
  const real* val;
  real* x;
  const real* bp;        // b in dependency-level order (gather_perm_kernel)
  const real* diag;      // per permuted row
  const i4_t* rowmeta;     // per permuted row: {row start, row end, diagonal position, original row id}
  const i4_t* desc;        // per dependency level: {first row, last row + 1, first nnz, last nnz + 1}
  real omega;
  int32_t lvl_begin;       // dependency levels [lvl_begin, lvl_end) in sweep order
  int32_t lvl_end;
  int32_t step;            // +1 forward, -1 backward (then lvl_begin > lvl_end)
  unsigned long long* tim; // diagnostics (amgh_debug_chain_timing): per-phase shader-cycle sums, or nullptr
  int64_t ldx, ldb;        // multi-RHS: workgroup blockIdx.x runs the chain for column blockIdx.x of x / bp
};

constexpr int kChainThreads = 1024;  // = max rows of a chained dependency level: one row per thread
constexpr int kChainLds = 8192;      // products staged per pass (64 KiB + skew)
constexpr int kChainLdsX = 8192;     // operators with at most this many columns keep x itself in LDS

// The chain runs as ONE workgroup of T threads.  Every wave of the workgroup executes the whole
// per-level loop body, so T is the smallest of 64 / 256 / 1024 that gives one thread per row of the
// segment's widest level (a 1024-thread workgroup on a 6-row level spends its time issuing
// instructions for 15 idle waves).  PF = nonzeros per thread prefetched for the NEXT level.
template <int PF>
struct ChainRow {
  i4_t m;       // rowmeta
  real d, b;  // diagonal, right-hand side
  real pv[PF];
  int pc[PF];
};

// Everything a dependency level needs that does NOT depend on x: issued one level
// ahead so that only the x gather, the LDS pass and the x store are on the
// critical path between two workgroup barriers.
template <int T, int PF>
__device__ __forceinline__ void chain_prefetch(const ChainArgs& a, const i4_t ds, int tid, ChainRow<PF>& o) {
  const int r = ds.x + tid;
  if (r < ds.y) {
    o.m = a.rowmeta[r];
    o.d = a.diag[r];
    o.b = a.bp[r];
  } else {
    o.m = i4_t{0, 0, -1, -1};
    o.d = 0.0;
    o.b = 0.0;
  }
#pragma unroll
  for (int e = 0; e < PF; ++e) {
    const int k = ds.z + tid + e * T;
    if (k < ds.w) {
      o.pv[e] = a.val[k];
      o.pc[e] = a.col[k];
    } else {
      o.pv[e] = 0.0;
      o.pc[e] = 0;
    }
  }
}

// LDSX: the whole x vector of the operator lives in LDS for the duration of the launch (small
// hierarchy levels): no global memory access is left on the per-level critical path — gathers
// and updates are LDS traffic, the prefetches of the next level's matrix rows are the only
// vector-memory operations in flight and are consumed one level later.
template <bool SOR, bool LDSX, int T, int PF>
__global__ __launch_bounds__(T) void gs_chain_kernel(ChainArgs a, int n) {
  static_assert(PF * T <= kChainLds, "prefetched nonzeros must fit the first LDS pass");
  __shared__ real s_prod[kChainLds];  // un-skewed: few rows per level, bank conflicts are not the issue here
  __shared__ real s_x[LDSX ? kChainLdsX : 1];
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {  // independent right-hand-side columns, one workgroup each
    a.x += blockIdx.x * a.ldx;
    a.bp += blockIdx.x * a.ldb;
    a.tim = nullptr;
  }
  int lv = a.lvl_begin;
  if (lv == a.lvl_end) return;
  i4_t ds = a.desc[lv];
  i4_t ds_next = (lv + a.step != a.lvl_end) ? a.desc[lv + a.step] : ds;
  ChainRow<PF> cur;
  chain_prefetch<T, PF>(a, ds, tid, cur);
  if (LDSX) {
    for (int i = tid; i < n; i += T) s_x[i] = a.x[i];
    __syncthreads();
  }
  const real* xs = LDSX ? (const real*)s_x : (const real*)a.x;
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, nlv = 0;
  for (;;) {
    const unsigned long long t0 = a.tim ? clock64() : 0;
    const bool has_next = (lv + a.step != a.lvl_end);
    // descriptor two levels ahead (scalar load), rows + leading nonzeros one level ahead
    const bool has_next2 = has_next && (lv + 2 * a.step != a.lvl_end);
    const i4_t ds_next2 = has_next2 ? a.desc[lv + 2 * a.step] : ds_next;
    const int p0 = ds.z, p1 = ds.w;
    // critical path first: the x gathers of this level's (already prefetched) leading
    // nonzeros are issued BEFORE the next level's prefetch loads — vector-memory results
    // return in order, so anything issued ahead of the gathers would delay them.
    real xv[PF];
#pragma unroll
    for (int e = 0; e < PF; ++e) {
      const int k = p0 + tid + e * T;
      xv[e] = (k < p1) ? xs[cur.pc[e]] : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0a = a.tim ? clock64() : 0;
    ChainRow<PF> nxt;
    if (has_next) chain_prefetch<T, PF>(a, ds_next, tid, nxt);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = a.tim ? clock64() : 0;
    unsigned long long t2 = 0, t3 = 0;

    real acc = 0.0;
    for (int c0 = p0; c0 < p1; c0 += kChainLds) {
      const int c1 = min(c0 + kChainLds, p1);
      if (c0 == p0) {
#pragma unroll
        for (int e = 0; e < PF; ++e) {
          const int k = p0 + tid + e * T;
          if (k < c1) s_prod[k - c0] = cur.pv[e] * xv[e];
        }
        for (int k = p0 + tid + PF * T; k < c1; k += T) s_prod[k - c0] = a.val[k] * xs[a.col[k]];
      } else {
        for (int k = c0 + tid; k < c1; k += T) s_prod[k - c0] = a.val[k] * xs[a.col[k]];
      }
      if (a.tim && c0 == p0) t2 = clock64();
      if (LDSX) lds_barrier();  // x and the products live in LDS: the next level's prefetch stays in flight
      else __syncthreads();
      if (a.tim && c0 == p0) t3 = clock64();
      const int lo = max(cur.m.x, c0), hi = min(cur.m.y, c1);
      acc = seq_sum_skip(s_prod, lo - c0, hi - c0, cur.m.z - c0, acc);
      if (c1 < p1) { if (LDSX) lds_barrier(); else __syncthreads(); }
    }
    if (cur.m.w >= 0 && cur.d != 0.0) {
      const int i = ds.x + tid;  // x is in dependency-level order: this row's own position
      const real xn = SOR ? (1.0 - a.omega) * xs[i] + (a.omega / cur.d) * (cur.b - acc) : (cur.b - acc) / cur.d;
      if (LDSX) s_x[i] = xn;
      a.x[i] = xn;  // LDSX: fire-and-forget copy to HBM, nobody in this launch reads it back
    }
    const unsigned long long t4 = a.tim ? clock64() : 0;
    if (!has_next) break;
    if (LDSX) {
      lds_barrier();  // x of this dependency level (in LDS) visible to the whole workgroup
    } else {
      __threadfence_block();
      __syncthreads();  // x of this dependency level (in HBM) visible to the whole workgroup (one CU)
    }
    if (a.tim) {
      const unsigned long long t5 = clock64();
      tacc[0] += t1 - t0; tacc[5] += t0a - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2; tacc[3] += t4 - t3; tacc[4] += t5 - t4; ++nlv;
    }
    lv += a.step;
    ds = ds_next;
    ds_next = ds_next2;
    cur = nxt;
  }
  if (a.tim && tid == 0) {
    for (int q = 0; q < 5; ++q) atomicAdd(a.tim + q, tacc[q]);
    atomicAdd(a.tim + 5, nlv);
    atomicAdd(a.tim + 6, 1ull);
    atomicAdd(a.tim + 7, tacc[5]);
  }
}

// ---- the chain with the WHOLE operator in LDS --------------------------------------------------------------------------
// gs_chain_kernel keeps x in LDS and prefetches the next level's rows one level ahead, but every level still waits for
// that prefetch: one global-memory round trip (~2 us at the clocks of an otherwise idle chip) per dependency level,
// against ~0.2 us of LDS work.  A tiny operator (lin_elastic_2d: 208 rows, 2 632 nonzeros, 46 levels; the reference's
// nns_test.jl:213-226 configuration) fits LDS entirely — matrix, row data, level descriptors, x, b: it is loaded once,
// and the per-level loop touches no global memory at all.  Same products, same in-order row sums as gs_chain_kernel
// (bitwise the same result; the skipped entry is the diagonal, whose product is never added); x goes back to HBM when
// the sweep is over.
constexpr int kTinyRows = 1024;     // rows (= columns, = levels at most)
constexpr int kTinyNnz = 6144;
constexpr int kTinyLvlNnz = 2048;   // nonzeros of one dependency level
template <bool SOR, int T>
__global__ __launch_bounds__(T) void gs_chain_tiny_kernel(ChainArgs a, int n, int nnz, int nlev) {
  __shared__ real s_val[kTinyNnz];
  __shared__ int32_t s_col[kTinyNnz];
  __shared__ i4_t s_meta[kTinyRows];
  __shared__ i4_t s_desc[kTinyRows];
  __shared__ real s_d[kTinyRows], s_b[kTinyRows], s_x[kTinyRows];
  __shared__ real s_prod[kTinyLvlNnz];
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {  // independent right-hand-side columns, one workgroup each
    a.x += blockIdx.x * a.ldx;
    a.bp += blockIdx.x * a.ldb;
  }
  if (a.lvl_begin == a.lvl_end) return;
  for (int k = tid; k < nnz; k += T) { s_val[k] = a.val[k]; s_col[k] = a.col[k]; }
  for (int r = tid; r < n; r += T) { s_meta[r] = a.rowmeta[r]; s_d[r] = a.diag[r]; s_b[r] = a.bp[r]; s_x[r] = a.x[r]; }
  for (int l = tid; l < nlev; l += T) s_desc[l] = a.desc[l];
  __syncthreads();
  // Per level: products of the level's entries by all threads (col / val of the level were fetched into registers one
  // level ahead: only the x reads are on the critical path), one barrier, in-order row sums by the row's own thread (its
  // row data fetched one level ahead too), x update, one barrier.
  constexpr int PF = 2;                      // entries per thread held in registers for the next level
  i4_t ds = s_desc[a.lvl_begin];
  i4_t m = i4_t{0, 0, -1, -1};
  real d = 0.0, bv = 0.0, pv[PF];
  int pc[PF];
  auto fetch = [&](const i4_t& dd, i4_t& mm, real& dg, real& bb, real* v, int* c) {
    const int r = dd.x + tid;
    mm = i4_t{0, 0, -1, -1}; dg = 0.0; bb = 0.0;
    if (r < dd.y) { mm = s_meta[r]; dg = s_d[r]; bb = s_b[r]; }
#pragma unroll
    for (int e = 0; e < PF; ++e) {
      const int k = dd.z + tid + e * T;
      v[e] = 0.0; c[e] = 0;
      if (k < dd.w) { v[e] = s_val[k]; c[e] = s_col[k]; }
    }
  };
  fetch(ds, m, d, bv, pv, pc);
  for (int lv = a.lvl_begin; lv != a.lvl_end; lv += a.step) {
    const int p0 = ds.z, p1 = ds.w;
#pragma unroll
    for (int e = 0; e < PF; ++e) {
      const int k = p0 + tid + e * T;
      if (k < p1) s_prod[k - p0] = pv[e] * s_x[pc[e]];
    }
    for (int k = p0 + tid + PF * T; k < p1; k += T) s_prod[k - p0] = s_val[k] * s_x[s_col[k]];
    // the next level's registers: independent of x, in flight across the barrier
    const int lvn = lv + a.step;
    i4_t dsn = ds, mn;
    real dn, bn, pvn[PF];
    int pcn[PF];
    if (lvn != a.lvl_end) dsn = s_desc[lvn];
    fetch(dsn, mn, dn, bn, pvn, pcn);
    lds_barrier();
    const int r = ds.x + tid;
    if (r < ds.y) {
      const real acc = seq_sum_skip(s_prod, m.x - p0, m.y - p0, m.z - p0, 0.0);
      if (d != 0.0) s_x[r] = SOR ? (1.0 - a.omega) * s_x[r] + (a.omega / d) * (bv - acc) : (bv - acc) / d;
    }
    lds_barrier();   // this level's x is visible, s_prod may be overwritten
    ds = dsn; m = mn; d = dn; bv = bn;
#pragma unroll
    for (int e = 0; e < PF; ++e) { pv[e] = pvn[e]; pc[e] = pcn[e]; }
  }
  for (int r = tid; r < n; r += T) a.x[r] = s_x[r];
}

// ---- a whole small operator walked by ONE wave --------------------------------------------------------------------
// gs_chain_tiny_kernel pays two workgroup barriers and ~300 instructions per dependency level (staged products, row
// sums that skip the diagonal, descriptors).  A single wave needs no barrier at all — its LDS operations execute in
// program order — and what bounds it is its own instruction count, ~3 ns per instruction whatever it is
// (tools/wave_chain_probe, profiles/r03_block_wave.log).  So the level loop is cut to the bone.  The operator is laid out
// once, at schedule build time, as a RECORD of PACKED ROWS in level order: a row is one run of 16-byte chunks
//     [ v0 v1 | v2 v3 | ... | (.. dg rc) | c0 .. c7 | c8 .. ]
// off-diagonal values in their stored order, the diagonal, its reciprocal, then the columns as uint16 BYTE offsets into
// the LDS copy of x (short rows padded with 0 * x[zero slot]); the chunk count is odd, so 64 lanes read rows 16 k bytes
// apart without bank conflicts.  Behind the rows: step_ptr (uint16), a step being a piece of at most 64 rows of one level.
// All waves copy the record, b and x into LDS; wave 0 walks the steps, lane = row: a handful of ds_read_b128 off ONE
// address fetch a row's operands, those of the NEXT step are requested before the current step's x values are gathered
// (two register sets, loop unrolled by two), the step pointers sit in two registers (v_readlane).  Per row: gather, MAXK
// separately rounded multiply-adds in entry order, then the quotient by the diagonal as q0 = RN(n rc), the exact
// remainder n - dg q0 (one fma), RN(q0 + rem rc) — the correctly rounded n / dg (Markstein) as long as nothing leaves the
// normal range, else the division itself: the scalar loop's arithmetic, bit for bit.
constexpr int kWaveThreads = 256;
constexpr int kWaveMaxSteps = 124;
constexpr int kWaveMaxK = 36;
constexpr int kWaveVpc = 16 / (int)sizeof(real);   // values per chunk
inline __host__ __device__ int wave_nvc(int maxk) { return (maxk + 2 + kWaveVpc - 1) / kWaveVpc; }
inline __host__ __device__ int wave_ncc(int maxk) { return (maxk + 7) / 8; }
inline __host__ __device__ int wave_row_bytes(int maxk) { return 16 * ((wave_nvc(maxk) + wave_ncc(maxk)) | 1); }
struct WaveArgs {
  const unsigned char* rec; const real* bp; real* x; int64_t ldb, ldx; real omega; int32_t n, steps;
};
typedef double wave_f64x2 __attribute__((ext_vector_type(2)));
typedef float wave_f32x4 __attribute__((ext_vector_type(4)));
template <typename R> struct WaveVec16;
template <> struct WaveVec16<double> { typedef wave_f64x2 type; };
template <> struct WaveVec16<float> { typedef wave_f32x4 type; };
__device__ __forceinline__ double wave_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float wave_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <int MAXK>
struct WaveRow {
  static constexpr int NVC = (MAXK + 2 + kWaveVpc - 1) / kWaveVpc, NCC = (MAXK + 7) / 8, RS = 16 * ((NVC + NCC) | 1);
  real v[NVC * kWaveVpc];   // values, the diagonal (slot MAXK), its reciprocal (slot MAXK + 1)
  uint32_t c[NCC * 4];      // column byte offsets, two per word
  real bb; int p;
};
template <int MAXK>
__device__ __forceinline__ void wave_load(WaveRow<MAXK>& o, int p, bool act, const unsigned char* rec, const real* bl) {
  typedef WaveRow<MAXK> O;
  typedef typename WaveVec16<real>::type V;
  o.p = act ? p : -1;
  const int q = act ? p : 0;
  const unsigned char* row = rec + __umul24((unsigned)q, (unsigned)O::RS);   // (24-bit multiply: full rate)
#pragma unroll
  for (int k = 0; k < O::NVC; ++k) *(V*)&o.v[k * kWaveVpc] = *(const V*)(row + 16 * k);
#pragma unroll
  for (int k = 0; k < O::NCC; ++k) *(uint4*)&o.c[4 * k] = *(const uint4*)(row + 16 * (O::NVC + k));
  o.bb = bl[q];
}
template <int MAXK, bool SOR>
__device__ __forceinline__ void wave_row(const WaveRow<MAXK>& o, real* xl, real omega) {
  real xv[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const uint32_t w = o.c[k >> 1];
    xv[k] = *(const real*)((const char*)xl + ((k & 1) ? (w >> 16) : (w & 0xffffu)));
  }
  real acc = 0.0;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) acc += o.v[k] * xv[k];
  const real dg = o.v[MAXK], rc = o.v[MAXK + 1];
  if (o.p >= 0 && dg != 0.0) {
    if (SOR) { xl[o.p] = (1.0 - omega) * xl[o.p] + (omega / dg) * (o.bb - acc); return; }
    const real nn = o.bb - acc;
    real q = nn * rc;
    const real rem = wave_fma(-dg, q, nn);
    q = wave_fma(rem, rc, q);
    const real an = nn < (real)0 ? -nn : nn;
    const bool safe = sizeof(real) == 8 ? (an > (real)1e-200 && an < (real)1e200) : (an > (real)1e-25 && an < (real)1e25);
    // (a wave-uniform branch: as a plain select the compiler evaluates the whole division sequence in every step)
    if (__builtin_amdgcn_ballot_w64(!(rc != 0.0 && safe)) != 0) {
      asm volatile("; rows outside the normal range: the division itself" ::: "memory");   // (keeps the branch a branch)
      if (!(rc != 0.0 && safe)) q = nn / dg;
    }
    xl[o.p] = q;
  }
}
// one directional walk over the record's steps (wave 0 only)
template <bool SOR, bool BWD, int MAXK>
__device__ __forceinline__ void wave_walk(const unsigned char* rec, const real* bl, real* xl, int ns, int lp0, int lp1, real omega, int tid) {
#define AMGH_WAVE_SP(i, out)                                                   \
  {                                                                            \
    const int i_ = (i);                                                        \
    const int u0_ = __builtin_amdgcn_readlane(lp0, i_ & 63);                   \
    const int u1_ = __builtin_amdgcn_readlane(lp1, i_ & 63);                   \
    out = i_ < 64 ? u0_ : u1_;                                                 \
  }
  // step k of the sweep = step k (forward) or ns - 1 - k (backward) of the record; past the end: an empty step
#define AMGH_WAVE_RANGE(k, r0, r1)                                             \
  {                                                                            \
    const int k_ = (k);                                                        \
    const int st_ = BWD ? ns - 1 - k_ : k_;                                    \
    const bool in_ = k_ < ns;                                                  \
    int q0_, q1_;                                                              \
    AMGH_WAVE_SP(in_ ? st_ : 0, q0_);                                          \
    AMGH_WAVE_SP(in_ ? st_ + 1 : 0, q1_);                                      \
    r0 = q0_; r1 = in_ ? q1_ : q0_;                                            \
  }
  WaveRow<MAXK> A, B;
  int r0, r1;
  AMGH_WAVE_RANGE(0, r0, r1);
  wave_load<MAXK>(A, r0 + tid, r0 + tid < r1, rec, bl);
  for (int k = 0; k < ns; k += 2) {
    AMGH_WAVE_RANGE(k + 1, r0, r1);
    wave_load<MAXK>(B, r0 + tid, r0 + tid < r1, rec, bl);
    wave_row<MAXK, SOR>(A, xl, omega);
    AMGH_WAVE_RANGE(k + 2, r0, r1);
    wave_load<MAXK>(A, r0 + tid, r0 + tid < r1, rec, bl);
    wave_row<MAXK, SOR>(B, xl, omega);
  }
#undef AMGH_WAVE_RANGE
#undef AMGH_WAVE_SP
}
// DIR: 0 forward, 1 backward, 2 forward then backward (a symmetric sweep in one launch: the record is loaded once)
template <bool SOR, int DIR, int MAXK>
__global__ __launch_bounds__(kWaveThreads) void gs_wave_kernel(WaveArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wave_lds[];
  const int tid = threadIdx.x;
  const real* __restrict__ b = a.bp + (int64_t)blockIdx.x * a.ldb;   // independent right-hand-side columns, one workgroup each
  real* __restrict__ x = a.x + (int64_t)blockIdx.x * a.ldx;
  const int n = a.n, ns = a.steps;
  // LDS: x (n + 1 entries, the last one the zero slot) | b | the record — each padded to 16 bytes
  const int nxb = (int)(((size_t)(n + 1) * sizeof(real) + 15) & ~(size_t)15), nbb = (int)(((size_t)n * sizeof(real) + 15) & ~(size_t)15);
  real* xl = (real*)wave_lds;
  real* bl = (real*)(wave_lds + nxb);
  unsigned char* rec = wave_lds + nxb + nbb;
  const int recb = (int)(((size_t)n * WaveRow<MAXK>::RS + (size_t)(ns + 1) * 2 + 15) & ~(size_t)15);
  {
    const uint4* src = (const uint4*)a.rec;
    uint4* dst = (uint4*)rec;
    for (int e = tid; e < (recb >> 4); e += kWaveThreads) dst[e] = src[e];
  }
  for (int p = tid; p < n; p += kWaveThreads) { bl[p] = b[p]; xl[p] = x[p]; }
  if (tid == 0) xl[n] = 0.0;
  __syncthreads();
  if (tid >= kWave) return;
  const uint16_t* stp = (const uint16_t*)(rec + (size_t)n * WaveRow<MAXK>::RS);
  const int lp0 = tid <= ns ? (int)stp[tid] : n;
  const int lp1 = tid + 64 <= ns ? (int)stp[tid + 64] : n;
  if (DIR != 1) wave_walk<SOR, false, MAXK>(rec, bl, xl, ns, lp0, lp1, a.omega, tid);
  if (DIR != 0) wave_walk<SOR, true, MAXK>(rec, bl, xl, ns, lp0, lp1, a.omega, tid);
  for (int p = tid; p < n; p += kWave) x[p] = xl[p];
}

// ---- the same walk with FOUR lanes per row (round 6) -----------------------------------------------------------------------
// What bounds the single wave is its instruction count, and with lane = row that count is the row length: the 18- and 30-entry
// rows of config C5 (lin_elastic_2d, 208 / 39 rows, ~4 rows per dependency level) cost 0.47 / 0.65 us per step while 60 of the
// 64 lanes idle — the two symmetric sweeps of its levels are 157 of the V-cycle's 185 us.  Here a row is four MINI-ROWS of the
// same packed format (entry k of the row = entry k / 4 of mini-row k % 4; each carries the diagonal and its reciprocal), lane =
// (row, quarter): a quarter of the chunks, gathers and multiply-adds per lane, the four partial sums added by a DPP butterfly
// inside the quad, then the same quotient in all four lanes, lane 0 of the quad stores.  A step is a piece of at most 16 rows of
// one level.  Exact Gauss-Seidel / SOR in the scalar loop's row order; a row's additions are reassociated into four
// interleaved partial sums (deterministic; sweeps <= 1e-14 from the scalar loop, tests/test_gpu_waveq.py); the one-lane walk
// stays behind the tunable gs_wave_quad = 0 for the tests that pin bits.
constexpr int kWaveQ = 4;
template <int E>
__device__ __forceinline__ void waveq_load(WaveRow<E>& o, int p, bool act, int sub, const unsigned char* rec, const real* bl) {
  typedef WaveRow<E> O;
  typedef typename WaveVec16<real>::type V;
  o.p = act ? p : -1;
  const int q = act ? p : 0;
  const unsigned char* row = rec + __umul24((unsigned)(q * kWaveQ + sub), (unsigned)O::RS);
#pragma unroll
  for (int k = 0; k < O::NVC; ++k) *(V*)&o.v[k * kWaveVpc] = *(const V*)(row + 16 * k);
#pragma unroll
  for (int k = 0; k < O::NCC; ++k) *(uint4*)&o.c[4 * k] = *(const uint4*)(row + 16 * (O::NVC + k));
  o.bb = bl[q];
}
template <int E, bool SOR>
__device__ __forceinline__ void waveq_row(const WaveRow<E>& o, real* xl, real omega, int sub) {
  real xv[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const uint32_t w = o.c[k >> 1];
    xv[k] = *(const real*)((const char*)xl + ((k & 1) ? (w >> 16) : (w & 0xffffu)));
  }
  real acc = 0.0;
#pragma unroll
  for (int k = 0; k < E; ++k) acc += o.v[k] * xv[k];
  acc = tree_add<kWaveQ>(acc);
  const real dg = o.v[E], rc = o.v[E + 1];
  if (o.p >= 0 && dg != 0.0) {
    if (SOR) { if (sub == 0) xl[o.p] = (1.0 - omega) * xl[o.p] + (omega / dg) * (o.bb - acc); return; }
    const real nn = o.bb - acc;
    real q = nn * rc;
    const real rem = wave_fma(-dg, q, nn);
    q = wave_fma(rem, rc, q);
    const real an = nn < (real)0 ? -nn : nn;
    const bool safe = sizeof(real) == 8 ? (an > (real)1e-200 && an < (real)1e200) : (an > (real)1e-25 && an < (real)1e25);
    if (__builtin_amdgcn_ballot_w64(!(rc != 0.0 && safe)) != 0) {
      asm volatile("; rows outside the normal range: the division itself" ::: "memory");
      if (!(rc != 0.0 && safe)) q = nn / dg;
    }
    if (sub == 0) xl[o.p] = q;
  }
}
template <bool SOR, bool BWD, int E>
__device__ __forceinline__ void waveq_walk(const unsigned char* rec, const real* bl, real* xl, int ns, int lp0, int lp1, real omega, int tid) {
#define AMGH_WAVE_SP(i, out)                                                   \
  {                                                                            \
    const int i_ = (i);                                                        \
    const int u0_ = __builtin_amdgcn_readlane(lp0, i_ & 63);                   \
    const int u1_ = __builtin_amdgcn_readlane(lp1, i_ & 63);                   \
    out = i_ < 64 ? u0_ : u1_;                                                 \
  }
#define AMGH_WAVE_RANGE(k, r0, r1)                                             \
  {                                                                            \
    const int k_ = (k);                                                        \
    const int st_ = BWD ? ns - 1 - k_ : k_;                                    \
    const bool in_ = k_ < ns;                                                  \
    int q0_, q1_;                                                              \
    AMGH_WAVE_SP(in_ ? st_ : 0, q0_);                                          \
    AMGH_WAVE_SP(in_ ? st_ + 1 : 0, q1_);                                      \
    r0 = q0_; r1 = in_ ? q1_ : q0_;                                            \
  }
  const int rw = tid >> 2, sub = tid & (kWaveQ - 1);
  WaveRow<E> A, B;
  int r0, r1;
  AMGH_WAVE_RANGE(0, r0, r1);
  waveq_load<E>(A, r0 + rw, r0 + rw < r1, sub, rec, bl);
  for (int k = 0; k < ns; k += 2) {
    AMGH_WAVE_RANGE(k + 1, r0, r1);
    waveq_load<E>(B, r0 + rw, r0 + rw < r1, sub, rec, bl);
    waveq_row<E, SOR>(A, xl, omega, sub);
    AMGH_WAVE_RANGE(k + 2, r0, r1);
    waveq_load<E>(A, r0 + rw, r0 + rw < r1, sub, rec, bl);
    waveq_row<E, SOR>(B, xl, omega, sub);
  }
#undef AMGH_WAVE_RANGE
#undef AMGH_WAVE_SP
}
template <bool SOR, int DIR, int E>
__global__ __launch_bounds__(kWaveThreads) void gs_waveq_kernel(WaveArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wave_lds[];
  const int tid = threadIdx.x;
  const real* __restrict__ b = a.bp + (int64_t)blockIdx.x * a.ldb;
  real* __restrict__ x = a.x + (int64_t)blockIdx.x * a.ldx;
  const int n = a.n, ns = a.steps;
  // LDS: x (n + 1 entries, the last one the zero slot) | b | the record (4 n mini-rows, the step pointers) — each padded to 16 bytes
  const int nxb = (int)(((size_t)(n + 1) * sizeof(real) + 15) & ~(size_t)15), nbb = (int)(((size_t)n * sizeof(real) + 15) & ~(size_t)15);
  real* xl = (real*)wave_lds;
  real* bl = (real*)(wave_lds + nxb);
  unsigned char* rec = wave_lds + nxb + nbb;
  const int recb = (int)(((size_t)n * kWaveQ * WaveRow<E>::RS + (size_t)(ns + 1) * 2 + 15) & ~(size_t)15);
  {
    const uint4* src = (const uint4*)a.rec;
    uint4* dst = (uint4*)rec;
    for (int e = tid; e < (recb >> 4); e += kWaveThreads) dst[e] = src[e];
  }
  for (int p = tid; p < n; p += kWaveThreads) { bl[p] = b[p]; xl[p] = x[p]; }
  if (tid == 0) xl[n] = 0.0;
  __syncthreads();
  if (tid >= kWave) return;
  const uint16_t* stp = (const uint16_t*)(rec + (size_t)n * kWaveQ * WaveRow<E>::RS);
  const int lp0 = tid <= ns ? (int)stp[tid] : n;
  const int lp1 = tid + 64 <= ns ? (int)stp[tid + 64] : n;
  if (DIR != 1) waveq_walk<SOR, false, E>(rec, bl, xl, ns, lp0, lp1, a.omega, tid);
  if (DIR != 0) waveq_walk<SOR, true, E>(rec, bl, xl, ns, lp0, lp1, a.omega, tid);
  for (int p = tid; p < n; p += kWave) x[p] = xl[p];
}

// Damped Jacobi on x = 0 (every pre-smoother below the fine level of a cycle, and the fine one of ldiv!): the sweep's
// matrix pass multiplies zeros — x_new = (1 - w) 0 + w ((b - 0) / d) is the SAME expression the stream kernel evaluates
// with a row sum of +0 (bitwise the same result), as a vector kernel.  Rows with a zero diagonal keep their x = 0
// (smoother.jl:132-137).  gridDim.y = right-hand-side columns.
__global__ void jacobi_zero_kernel(const real* __restrict__ b, const real* __restrict__ diag, real* __restrict__ y, int64_t n,
                                   real omega, int64_t ldb, int64_t ldy) {
  b += blockIdx.y * ldb;
  y += blockIdx.y * ldy;
  const real t = 0.0, acc = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    const real d = diag[r];
    const real cand = (1.0 - omega) * t + omega * ((b[r] - acc) / d);
    y[r] = (d == 0.0) ? t : cand;
  }
}

// ---- restriction / prolongation of a BLOCK of right-hand sides -------------------------------------------------------
// In the level-ordered cycle the gathers of R and P are one 64-byte sector per matrix entry (a coarse row's fine
// neighbours sit in different dependency levels, tools/order_probe.py): column by column a block of bs right-hand sides
// pays that bs times (bs = 8: 3.3 + 3.4 ms on the fine level of the 256^3 hierarchy).  With the gathered vector stored
// INTERLEAVED (the bs values of a row side by side: one sector for bs = 8) one gather serves every column: BS lanes per
// row, lane q multiplies column q.  Each lane walks its row's entries in index order with strictly ordered adds — the
// sums of the single-column stream kernel, bit for bit.
template <int BS>
__global__ __launch_bounds__(256) void to_interleaved_kernel(const real* __restrict__ src, int64_t ld, real* __restrict__ dst,
                                                             int64_t n) {
  __shared__ real tile[BS][65];
  const int64_t i0 = (int64_t)blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * BS; e += 256) {
    const int q = e >> 6, r = e & 63;
    if (i0 + r < n) tile[q][r] = src[i0 + r + (int64_t)q * ld];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * BS; e += 256) {
    const int r = e / BS, q = e % BS;
    if (i0 + r < n) dst[(i0 + r) * BS + q] = tile[q][r];
  }
}
// y[:, q] = M xil[:, q] (ADD = false) or y[:, q] += M xil[:, q]: xil interleaved (ncols x BS), y column-major (ldy apart)
// (CODED: value-coded columns, StreamArgs::ccol — col holds the words, val the table of vtab_n distinct values)
template <bool ADD, int BS, bool CODED = false>
__global__ __launch_bounds__(256) void csr_il_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     const real* __restrict__ val, int64_t nrows, const real* __restrict__ xil,
                                                     real* __restrict__ y, int64_t ldy, int vtab_n = 0) {
  __shared__ real s_tab[CODED ? kCodeMax : 1];
  if (CODED) {
    if ((int)threadIdx.x < vtab_n) s_tab[threadIdx.x] = val[threadIdx.x];
    __syncthreads();
  }
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / BS;
  const int q = (int)(t % BS);
  if (row >= nrows) return;
  const int32_t j0 = rowptr[row], j1 = rowptr[row + 1];
  real acc = 0.0;
  int32_t j = j0;
  for (; j + 4 <= j1; j += 4) {   // four entries' loads in flight, adds in index order
    int32_t c[4]; real v[4], xv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (CODED) { const unsigned w = (unsigned)col[j + e]; c[e] = (int32_t)(w & ((1u << kCodeBits) - 1)); v[e] = s_tab[w >> kCodeBits]; }
      else { c[e] = col[j + e]; v[e] = val[j + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) xv[e] = xil[(int64_t)c[e] * BS + q];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += v[e] * xv[e];
  }
  for (; j < j1; ++j) {
    if (CODED) { const unsigned w = (unsigned)col[j]; acc += s_tab[w >> kCodeBits] * xil[(int64_t)(w & ((1u << kCodeBits) - 1)) * BS + q]; }
    else acc += val[j] * xil[(int64_t)col[j] * BS + q];
  }
  real* yp = y + row + (int64_t)q * ldy;
  *yp = ADD ? *yp + acc : acc;
}
// r[:, q] = b[:, q] - A x[:, q] for a block of BS right-hand sides with x COLUMN-MAJOR (ldx apart; the smoother's own
// level-ordered vector): BS lanes per row again, so the row's entries are loaded once for the whole block (the stream
// kernel re-reads the matrix tile per column, from L2 at best); lane q gathers column q — neighbouring rows of a wave
// gather neighbouring x, sector sharing is per column as in the single-column kernel.  In-order sums: bitwise the
// stream kernel's residual.
template <int BS, bool CODED = false>
__global__ __launch_bounds__(256) void csr_resid_cols_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             const real* __restrict__ val, int64_t nrows,
                                                             const real* __restrict__ x, int64_t ldx, const real* __restrict__ b,
                                                             int64_t ldb, real* __restrict__ y, int64_t ldy, int vtab_n = 0) {
  __shared__ real s_tab[CODED ? kCodeMax : 1];
  if (CODED) {
    if ((int)threadIdx.x < vtab_n) s_tab[threadIdx.x] = val[threadIdx.x];
    __syncthreads();
  }
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / BS;
  const int q = (int)(t % BS);
  if (row >= nrows) return;
  const real* xq = x + (int64_t)q * ldx;
  const int32_t j0 = rowptr[row], j1 = rowptr[row + 1];
  real acc = 0.0;
  int32_t j = j0;
  for (; j + 4 <= j1; j += 4) {
    int32_t c[4]; real v[4], xv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (CODED) { const unsigned w = (unsigned)col[j + e]; c[e] = (int32_t)(w & ((1u << kCodeBits) - 1)); v[e] = s_tab[w >> kCodeBits]; }
      else { c[e] = col[j + e]; v[e] = val[j + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) xv[e] = xq[c[e]];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += v[e] * xv[e];
  }
  for (; j < j1; ++j) {
    if (CODED) { const unsigned w = (unsigned)col[j]; acc += s_tab[w >> kCodeBits] * xq[w & ((1u << kCodeBits) - 1)]; }
    else acc += val[j] * xq[col[j]];
  }
  y[row + (int64_t)q * ldy] = b[row + (int64_t)q * ldb] - acc;
}

// ---- one WIDE dependency level from the slot layout -------------------------------------------
// A wide level's launch is latency-bound: kernarg -> row pointers -> col/val -> x gather are four
// dependent round trips.  The slot layout removes one: the level's nonzeros are packed into slots of
// kSlot entries (whole rows only, zero padding), workgroup b owns slot slot0 + b, so the
// addresses of its col/val loads follow from the kernel arguments and blockIdx alone; which rows the
// slot holds (slot_row, rowmeta) is fetched in parallel and only needed for the row sums.
constexpr int kSlot = 512;  // 128: 39.5, 256: 37.5, 512: 36.9, 1024: 37.2 ms per smoother pass over all levels of the 256^3 hierarchy

struct SlotArgs {
  const int32_t* wcol;      // slot arrays: nslots * kSlot entries (padding: col 0, val 0)
  const real* wval;
  const int32_t* slot_row;  // per slot: {first row, end row} in dependency-level order
  const i4_t* wmeta;        // per level-ordered row: {start, end, diagonal position} in the slot arrays
  const real* diag;
  const real* bp;
  real* x;                // x in dependency-level order
  real omega;
  int32_t slot0;            // first slot of this dependency level
  int32_t nslots;           // slots of this dependency level
  int32_t xcd_map;          // 1: XCD-contiguous slot mapping (grid padded to a multiple of 8)
  int64_t ldx, ldb;         // multi-RHS: column strides of x / bp
  int32_t ncolv;            // column GROUPS of this launch (NCV columns each), mapped by multi_column_block()
};

// NCV = right-hand-side columns one workgroup sweeps (multi-RHS blocks): the slot's (col, val) are loaded once and
// the NCV x gathers of every entry are in flight together, so a block of columns costs the level's latency chain
// once AND keeps the workgroup count of a single column (a workgroup per column made wide levels throughput-bound:
// 8 columns = 3 x the time).  Two entries per thread (half the waves) measured 2 % slower than one.
template <bool SOR, int NCV>
__global__ __launch_bounds__(kSlot) void gs_slot_kernel(SlotArgs a) {
  __shared__ real s_prod[NCV * kSlot];
  const int tid = threadIdx.x;
  int lb, cg;  // slot, column group
  multi_column_block(a.ncolv, lb, cg);
  if (cg > 0) {
    a.x += (int64_t)cg * NCV * a.ldx;
    a.bp += (int64_t)cg * NCV * a.ldb;
  }
  if (a.xcd_map) lb = xcd_block(lb, a.nslots);
  if (lb >= a.nslots) return;
  const int s = a.slot0 + lb;
  const int base = s * kSlot;
  const real v = a.wval[base + tid];
  const int c = a.wcol[base + tid];
  const int r0 = a.slot_row[2 * s], r1 = a.slot_row[2 * s + 1];
  real xv[NCV];
#pragma unroll
  for (int q = 0; q < NCV; ++q) xv[q] = a.x[c + q * a.ldx];
  // (row, column) tasks of the slot are spread over the threads: task t = column * nrows + row.  The first
  // TPT tasks of a thread have their row data requested up front (same round trip as the gathers).
  constexpr int TPT = NCV > 1 ? 2 : 1;
  const int nrows = r1 - r0, ntask = nrows * NCV;
  i4_t m[TPT];
  real d[TPT], bb[TPT];
  int tr[TPT], tq[TPT];
#pragma unroll
  for (int u = 0; u < TPT; ++u) {
    const int t = tid + u * kSlot;
    m[u] = i4_t{0, 0, -1, 0}; d[u] = 0.0; bb[u] = 0.0; tr[u] = 0; tq[u] = 0;
    if (t < ntask) {
      if (NCV > 1) { tq[u] = t / nrows; tr[u] = t - tq[u] * nrows; }   // (one column: task = row, no integer division)
      else tr[u] = t;
      m[u] = a.wmeta[r0 + tr[u]];
      d[u] = a.diag[r0 + tr[u]];
      bb[u] = a.bp[r0 + tr[u] + tq[u] * a.ldb];
    }
  }
#pragma unroll
  for (int q = 0; q < NCV; ++q) s_prod[q * kSlot + tid] = v * xv[q];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < TPT; ++u) {
    if (tid + u * kSlot < ntask && d[u] != 0.0) {
      const real acc = seq_sum_skip(s_prod + tq[u] * kSlot, m[u].x - base, m[u].y - base, m[u].z - base, 0.0);
      real* xq = a.x + tq[u] * a.ldx;
      const int i = r0 + tr[u];
      xq[i] = SOR ? (1.0 - a.omega) * xq[i] + (a.omega / d[u]) * (bb[u] - acc) : (bb[u] - acc) / d[u];
    }
  }
  for (int t = tid + TPT * kSlot; t < ntask; t += kSlot) {  // slots of very short rows: more tasks than 2 per thread
    const int q = NCV > 1 ? t / nrows : 0, r = t - q * nrows;
    const i4_t m2 = a.wmeta[r0 + r];
    const real d2 = a.diag[r0 + r];
    if (d2 != 0.0) {
      const real acc = seq_sum_skip(s_prod + q * kSlot, m2.x - base, m2.y - base, m2.z - base, 0.0);
      real* xq = a.x + q * a.ldx;
      const int i = r0 + r;
      const real bq = a.bp[i + q * a.ldb];
      xq[i] = SOR ? (1.0 - a.omega) * xq[i] + (a.omega / d2) * (bq - acc) : (bq - acc) / d2;
    }
  }
}

// The same slot launch with LPR lanes per row in the row sums: a composite row of 30-100 entries summed by ONE thread
// is 30-100 dependent LDS reads + adds while 500 threads of the workgroup idle; LPR lanes take the entries LPR apart
// and a __shfl_xor tree adds the partial sums.  Changes the order of the additions (not the scalar loop's any more):
// used for merged groups only, whose composite rows already differ from the scalar recurrence at the 1e-16 level.
// EPT = 2: 256 threads, two consecutive entries per thread (one 16-byte + one 8-byte load): twice as many
// workgroups are resident per CU, so that a group of more than ~1000 slots still runs in one residency round.
template <bool SOR, int LPR, int EPT>
__global__ __launch_bounds__(kSlot / EPT) void gs_slot_lpr_kernel(SlotArgs a) {
  static_assert(LPR >= 1 && LPR <= 32 && (LPR & (LPR - 1)) == 0, "lanes per row: a power of two inside one wave");
  static_assert(EPT == 1 || EPT == 2, "entries per thread");
  constexpr int T = kSlot / EPT;
  __shared__ real s_prod[kSlot];
  const int tid = threadIdx.x;
  int lb = blockIdx.x;
  if (a.xcd_map) lb = xcd_block(lb, a.nslots);
  if (lb >= a.nslots) return;
  const int s = a.slot0 + lb;
  const int base = s * kSlot;
  real v[EPT], xv[EPT];
  int c[EPT];
  if (EPT == 1) {
    v[0] = a.wval[base + tid]; c[0] = a.wcol[base + tid];
  } else {
    const d2_t vv = *(const d2_t*)(a.wval + base + 2 * tid);
    const i2_t cc = *(const i2_t*)(a.wcol + base + 2 * tid);
    v[0] = vv.x; v[EPT - 1] = vv.y; c[0] = cc.x; c[EPT - 1] = cc.y;
  }
  const int r0 = a.slot_row[2 * s], nrows = a.slot_row[2 * s + 1] - r0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) xv[e] = a.x[c[e]];
  constexpr int RPP = T / LPR;  // rows per pass
  const int sub = tid % LPR;
  int r = tid / LPR;
  i4_t m = i4_t{0, 0, -1, 0};
  real d = 0.0, bb = 0.0;
  // (row data through the scalar cache — the rows of a wave have wave-uniform addresses, and with LPR >= 8 the lanes fetch the
  // same 32 bytes 8 or 16 times over — was measured: 2.19 -> 2.43 ms per symmetric sweep of the 1.4 M-row level of the 256^3
  // hierarchy; 12 scalar loads of distinct lines per wave into a cache that starts every launch cold are slower than 3 vector loads)
  if (r < nrows) { m = a.wmeta[r0 + r]; d = a.diag[r0 + r]; bb = a.bp[r0 + r]; }  // same round trip as the gather
#pragma unroll
  for (int e = 0; e < EPT; ++e) s_prod[EPT * tid + e] = v[e] * xv[e];
  __syncthreads();
  for (int pass = 0; pass * RPP < nrows; ++pass, r += RPP) {
    if (pass > 0) {
      m = i4_t{0, 0, -1, 0}; d = 0.0; bb = 0.0;
      if (r < nrows) { m = a.wmeta[r0 + r]; d = a.diag[r0 + r]; bb = a.bp[r0 + r]; }
    }
    real acc = 0.0;
    const int dz = m.z - base, qe = m.y - base;
    for (int q = m.x - base + sub; q < qe; q += LPR)
      if (q != dz) acc += s_prod[q];
    acc = tree_add<LPR>(acc);
    if (sub == 0 && r < nrows && d != 0.0) {
      const int i = r0 + r;
      a.x[i] = SOR ? (1.0 - a.omega) * a.x[i] + (a.omega / d) * (bb - acc) : (bb - acc) / d;
    }
  }
}

// ---- merged groups from a SELL-like layout ---------------------------------------------------------------------------
// The slot kernels stage every product in LDS and pay two dependent phases behind the matrix stream (gather, then
// LDS + row sums).  Here a WAVE owns a chunk of 64 / K consecutive rows of the group, K lanes per row: entry t * K + sub
// of row r sits at chunk_base + t * 64 + lane (lane = r * K + sub), so the col / val loads of an iteration are one
// coalesced 64-entry line, neighbouring lanes gather neighbouring x (neighbouring rows read neighbouring columns), and
// a lane accumulates its own entries in registers — no LDS, no barrier; the K partial sums of a row meet in a
// __shfl_xor tree.  Padding entries carry col = -1.  (tools/gs_step_bench: -18 % per launch on 27-entry rows, -35 % on
// 60-entry rows against the slot kernel; profiles/r02_gs_step_bench_sell.log.)  K = 1 adds in index order like the
// scalar loop; K > 1 changes the order of the additions (merged groups only, whose rows already differ at 1e-16).
struct SellArgs {
  const int32_t* scol;      // padded entries, chunk after chunk
  const real* sval;
  const i2_t* chunk;        // per chunk {offset in units of 64 entries, iterations}
  const real* diag;
  const real* bp;
  real* x;
  real omega;
  int32_t row0, nrows;      // the group's rows (level order)
  int32_t chunk0, nchunks;
  int32_t xcd_map;
  int64_t ldx, ldb;         // column strides of x and bp (blocks of right-hand sides)
};
// NCV columns of a block of right-hand sides per launch (blockIdx.y picks the column group): the matrix entries are
// read once for all of them; each column's additions run in the order of the single-column launch (bitwise the same).
template <bool SOR, int K, int BATCH, int NCV>
__global__ __launch_bounds__(256) void gs_sell_kernel(SellArgs a) {
  constexpr int C = kWave / K;
  const int nwg = (a.nchunks + 3) >> 2;
  int wb = blockIdx.x;
  if (a.xcd_map) wb = xcd_block(wb, nwg);
  if (wb >= nwg) return;
  const int ch = wb * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: the chunk descriptor — and with K = 64 the row's diagonal and right-hand side — through the scalar cache)
  if (ch >= a.nchunks) return;
  const int lane = threadIdx.x & (kWave - 1);
  const i2_t cd = a.chunk[a.chunk0 + ch];
  const int64_t base = (int64_t)(uint32_t)cd.x * kWave + lane;
  const int r = ch * C + lane / K;
  const bool live = r < a.nrows;
  real* x = a.x + (int64_t)blockIdx.y * NCV * a.ldx;
  const real* bp = a.bp + (int64_t)blockIdx.y * NCV * a.ldb;
  real d = 0.0, bb[NCV], acc[NCV];
#pragma unroll
  for (int k = 0; k < NCV; ++k) { bb[k] = 0.0; acc[k] = 0.0; }
  if (live) {
    d = a.diag[a.row0 + r];
#pragma unroll
    for (int k = 0; k < NCV; ++k) bb[k] = bp[a.row0 + r + k * a.ldb];
  }
  // entries t .. t + BATCH - 1 of every lane per round; the NEXT round's (col, val) are requested before this round's x
  // gathers (round 6: with 8 columns per launch and 2 entries per round a 200-entry row was four dependent round trips —
  // col / val, gather, col / val, gather — where the single column has two).  Per lane the products are added in entry
  // order whatever BATCH is: the sums keep their bits.
  const int nit = cd.y;   // (uniform over the wave: one chunk per wave)
  int c[BATCH];
  real v[BATCH];
#pragma unroll
  for (int e = 0; e < BATCH; ++e) {
    c[e] = -1; v[e] = 0.0;
    if (e < nit) { c[e] = a.scol[base + (int64_t)e * kWave]; v[e] = a.sval[base + (int64_t)e * kWave]; }
  }
  for (int t = 0; t < nit; t += BATCH) {
    int cn[BATCH];
    real vn[BATCH];
#pragma unroll
    for (int e = 0; e < BATCH; ++e) {
      cn[e] = -1; vn[e] = 0.0;
      if (t + BATCH + e < nit) { cn[e] = a.scol[base + (int64_t)(t + BATCH + e) * kWave]; vn[e] = a.sval[base + (int64_t)(t + BATCH + e) * kWave]; }
    }
    real xv[BATCH][NCV];
#pragma unroll
    for (int e = 0; e < BATCH; ++e)
#pragma unroll
      for (int k = 0; k < NCV; ++k) xv[e][k] = c[e] >= 0 ? x[c[e] + k * a.ldx] : 0.0;
#pragma unroll
    for (int e = 0; e < BATCH; ++e)
      if (c[e] >= 0) {
#pragma unroll
        for (int k = 0; k < NCV; ++k) acc[k] += v[e] * xv[e][k];
      }
#pragma unroll
    for (int e = 0; e < BATCH; ++e) { c[e] = cn[e]; v[e] = vn[e]; }
  }
#pragma unroll
  for (int k = 0; k < NCV; ++k) {
    acc[k] = tree_add<K>(acc[k]);
  }
  if (live && (lane % K) == 0 && d != 0.0) {
    const int i = a.row0 + r;
#pragma unroll
    for (int k = 0; k < NCV; ++k) {
      real* xi = x + i + k * a.ldx;
      *xi = SOR ? (1.0 - a.omega) * *xi + (a.omega / d) * (bb[k] - acc[k]) : (bb[k] - acc[k]) / d;
    }
  }
}
// a group's rows from the (composite) CSR into the SELL arrays: one thread per lane slot of a chunk
__global__ void sell_fill_kernel(const int32_t* prow, const int32_t* pcol, const real* pval, const i2_t* chunk,
                                 int chunk0, int nchunks, int row0, int nrows, int K, int32_t* scol, real* sval) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int ch = (int)(gid >> 6), lane = (int)(gid & 63);
  if (ch >= nchunks) return;
  const i2_t cd = chunk[chunk0 + ch];
  const int C = kWave / K;
  const int r = ch * C + lane / K, sub = lane % K;
  const int64_t base = (int64_t)(uint32_t)cd.x * kWave + lane;
  int32_t a0 = 0, len = 0;
  if (r < nrows) { a0 = prow[row0 + r]; len = prow[row0 + r + 1] - a0; }
  for (int t = 0; t < cd.y; ++t) {
    const int e = t * K + sub;
    if (e < len) { scol[base + (int64_t)t * kWave] = pcol[a0 + e]; sval[base + (int64_t)t * kWave] = pval[a0 + e]; }
    else { scol[base + (int64_t)t * kWave] = -1; sval[base + (int64_t)t * kWave] = 0.0; }
  }
}

// ---- merged groups for a block of right-hand sides from an INTERLEAVED vector (round 6) -----------------------------------
// With the columns of a block ldx apart every matrix entry of a merged group costs one gather PER COLUMN, each its own
// cache line: at bs = 8 the 62-entry rows of the 1.4 M-row level of the 256^3 hierarchy ran 13.6 us per launch against 5.4
// for one column, and a what-if with one gather per entry said 4 of the 8 extra microseconds are those gathers (the L1 looks
// up one line per lane and cycle).  Here the sweep's vector [x ; s] is ALSO kept interleaved (xil: the BS values of a
// position side by side — one 64-byte sector at BS = 8); lane = (entry, column): the BS lanes of an entry read one sector,
// a wave's gather touches 64 / BS lines instead of 64.  (col, val) are loaded once per entry and handed to the BS lanes by
// wave shuffles.  Results go to xil AND to the column-major x (the rest of the cycle — residual, restriction, flip — reads
// that one), so segments without an interleaved kernel (chains, long-row slots) run as before and only their rows are
// copied over.  Exact Gauss-Seidel on the merged system; the order of a row's additions is the kernel's own (lanes per
// row = 64 / BS), as with every several-lanes-per-row kernel of the merged groups.
struct SlotIlArgs {
  const int32_t* wcol;      // slot arrays (as SlotArgs)
  const real* wval;
  const int32_t* slot_row;
  const i4_t* wmeta;
  const real* diag;
  real* xil;                // [x ; s] interleaved: position p, column q at p * BS + q
  real* x;                  // the column-major x (columns ldx apart): written too
  real omega;
  int32_t slot0, nslots, xcd_map;
  int32_t soff;             // position of s[0] in xil (= the operator's ncols)
  int64_t ldx;
};
template <bool SOR, int BS>
__global__ __launch_bounds__(kSlot) void gs_slot_il_kernel(SlotIlArgs a) {
  static_assert(BS == 2 || BS == 4 || BS == 8 || BS == 16, "columns per block");
  // Gather phase: a lane takes TWO columns of an entry (one 16-byte load at Float64); what bounds an 8-column launch once the
  // gathers are sectors is the NUMBER of vector-memory instructions a wave issues (replacing every gather address by one cached
  // address changed nothing, 12 more wave-uniform loads of row data per thread cost +33 %): 4 gathers + 12 shuffles per thread
  // instead of 8 + 24, and the row data through the scalar cache (the wave index made uniform).
  constexpr int LPE = BS / 2;         // lanes per entry in the gather phase
  constexpr int EPG = kWave / LPE;    // entries a wave gathers per round
  constexpr int NRG = LPE;            // rounds (64 entries per wave)
  constexpr int EPW = kWave / BS;     // lanes per (row, column) sum
  constexpr int NW = kSlot / kWave;
  typedef real r2_t __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) real s_prod[kSlot * BS];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(tid / kWave);   // (uniform: a wave's row data through the scalar cache — one row per wave here: level 2 of the 256^3 hierarchy at bs = 8 8.69 -> 8.30 ms)
  int lb = blockIdx.x;
  if (a.xcd_map) lb = xcd_block(lb, a.nslots);
  if (lb >= a.nslots) return;
  const int s = a.slot0 + lb;
  const int base = s * kSlot;
  const real v = a.wval[base + tid];
  const int c = a.wcol[base + tid];
  const int r0 = a.slot_row[2 * s], nrows = a.slot_row[2 * s + 1] - r0;
  const int qp = lane % LPE, gsub = lane / LPE;
  r2_t xv[NRG];
  real vv[NRG];
#pragma unroll
  for (int k = 0; k < NRG; ++k) {   // round k: entries w * 64 + k * EPG + (0 .. EPG - 1), LPE lanes each
    const int src = k * EPG + gsub;
    const int ck = __shfl(c, src, kWave);
    vv[k] = __shfl(v, src, kWave);
    xv[k] = *(const r2_t*)(a.xil + (int64_t)ck * BS + 2 * qp);
  }
  const int q = lane % BS, sub = lane / BS;
  // row data of this wave's first row, requested with the gathers (wave-uniform: scalar loads, but for the right-hand side)
  i4_t m = i4_t{0, 0, -1, 0};
  real d = 0.0, bb = 0.0;
  if (w < nrows) { m = a.wmeta[r0 + w]; d = a.diag[r0 + w]; bb = a.xil[(int64_t)(a.soff + r0 + w) * BS + q]; }
#pragma unroll
  for (int k = 0; k < NRG; ++k)     // entry e, column q at e * BS + q: lane's two products side by side
    *(r2_t*)(s_prod + (w * kWave + k * EPG) * BS + 2 * lane) = r2_t{vv[k] * xv[k].x, vv[k] * xv[k].y};
  __syncthreads();
  for (int r = w; r < nrows; r += NW) {   // a wave per row: EPW lanes per column
    if (r != w) { m = a.wmeta[r0 + r]; d = a.diag[r0 + r]; bb = a.xil[(int64_t)(a.soff + r0 + r) * BS + q]; }
    const int dz = m.z - base, qe = m.y - base;
    real acc = 0.0;
    for (int e = m.x - base + sub; e < qe; e += EPW)
      if (e != dz) acc += s_prod[e * BS + q];
#pragma unroll
    for (int o = BS; o < kWave; o <<= 1) acc += __shfl_xor(acc, o, kWave);
    if (sub == 0 && d != 0.0) {
      const int64_t i = r0 + r;
      real* xi = a.xil + i * BS + q;
      const real xn = SOR ? (1.0 - a.omega) * *xi + (a.omega / d) * (bb - acc) : (bb - acc) / d;
      *xi = xn;
      a.x[i + q * a.ldx] = xn;
    }
  }
}

// The SELL-like copy of a merged group the same way: the wave's 64 entry lanes of a storage iteration are taken 2 * 64 / BS at a
// time, BS / 2 lanes each (a lane gathers two columns); BATCH iterations' gathers are in flight together and the next round's
// (col, val) are requested before them.  Needs BS >= 4 and K >= 2 * 64 / BS (a row's K entry lanes are whole rounds); other
// shapes keep gs_sell_kernel and have their rows copied over.
struct SellIlArgs {
  const int32_t* scol;
  const real* sval;
  const i2_t* chunk;
  const real* diag;
  real* xil;
  real* x;
  real omega;
  int32_t row0, nrows, chunk0, nchunks, xcd_map;
  int32_t soff;
  int64_t ldx;
};
template <bool SOR, int K, int BS, int BATCH>
__global__ __launch_bounds__(256) void gs_sell_il_kernel(SellIlArgs a) {
  constexpr int C = kWave / K;        // rows per chunk
  constexpr int LPE = BS / 2;         // lanes per entry: a lane takes two columns (one 16-byte gather at Float64)
  constexpr int EPG = kWave / LPE;    // entry lanes per round
  constexpr int NR = LPE;             // rounds per storage iteration
  constexpr int G = K / EPG;          // rounds per row
  static_assert(BS >= 4 && K >= EPG && K >= BS && K % EPG == 0, "a row's entry lanes are whole rounds");
  typedef real r2_t __attribute__((ext_vector_type(2)));
  const int nwg = (a.nchunks + 3) >> 2;
  int wb = blockIdx.x;
  if (a.xcd_map) wb = xcd_block(wb, nwg);
  if (wb >= nwg) return;
  const int ch = wb * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: the chunk descriptor — and with K = 64 the row's diagonal and right-hand side — through the scalar cache)
  if (ch >= a.nchunks) return;
  const int lane = threadIdx.x & (kWave - 1);
  const i2_t cd = a.chunk[a.chunk0 + ch];
  const int64_t base = (int64_t)(uint32_t)cd.x * kWave + lane;
  const int qp = lane % LPE, sub = lane / LPE;
  const int r = ch * C + sub;   // the row this lane writes (lanes sub < C), columns 2 qp and 2 qp + 1
  const bool live = sub < C && r < a.nrows;
  real d = 0.0;
  r2_t bb = r2_t{0.0, 0.0};
  if (live) { d = a.diag[a.row0 + r]; bb = *(const r2_t*)(a.xil + (int64_t)(a.soff + a.row0 + r) * BS + 2 * qp); }
  r2_t acc[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) acc[k] = r2_t{0.0, 0.0};
  const int nit = cd.y;
  int c[BATCH];
  real v[BATCH];
#pragma unroll
  for (int e = 0; e < BATCH; ++e) {
    c[e] = -1; v[e] = 0.0;
    if (e < nit) { c[e] = a.scol[base + (int64_t)e * kWave]; v[e] = a.sval[base + (int64_t)e * kWave]; }
  }
  for (int t = 0; t < nit; t += BATCH) {
    int cn[BATCH];
    real vn[BATCH];
#pragma unroll
    for (int e = 0; e < BATCH; ++e) {
      cn[e] = -1; vn[e] = 0.0;
      if (t + BATCH + e < nit) { cn[e] = a.scol[base + (int64_t)(t + BATCH + e) * kWave]; vn[e] = a.sval[base + (int64_t)(t + BATCH + e) * kWave]; }
    }
    int cc[BATCH][NR];
    real vv[BATCH][NR];
    r2_t xv[BATCH][NR];
#pragma unroll
    for (int e = 0; e < BATCH; ++e)
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int src = k * EPG + sub;
        cc[e][k] = __shfl(c[e], src, kWave);
        vv[e][k] = __shfl(v[e], src, kWave);
        xv[e][k] = cc[e][k] >= 0 ? *(const r2_t*)(a.xil + (int64_t)cc[e][k] * BS + 2 * qp) : r2_t{0.0, 0.0};
      }
#pragma unroll
    for (int e = 0; e < BATCH; ++e)
#pragma unroll
      for (int k = 0; k < NR; ++k)
        if (cc[e][k] >= 0) { acc[k].x += vv[e][k] * xv[e][k].x; acc[k].y += vv[e][k] * xv[e][k].y; }
#pragma unroll
    for (int e = 0; e < BATCH; ++e) { c[e] = cn[e]; v[e] = vn[e]; }
  }
  // entry lane k * EPG + sub belongs to row (k * EPG + sub) / K = k / G of the chunk: the rounds of a row in this lane,
  // then the EPG lanes of the column pair
  r2_t tot = r2_t{0.0, 0.0};
#pragma unroll
  for (int mrow = 0; mrow < C; ++mrow) {
    r2_t ra = acc[mrow * G];
#pragma unroll
    for (int g2 = 1; g2 < G; ++g2) { ra.x += acc[mrow * G + g2].x; ra.y += acc[mrow * G + g2].y; }
#pragma unroll
    for (int o = LPE; o < kWave; o <<= 1) { ra.x += __shfl_xor(ra.x, o, kWave); ra.y += __shfl_xor(ra.y, o, kWave); }
    if (sub == mrow) tot = ra;
  }
  if (live && d != 0.0) {
    const int64_t i = a.row0 + r;
    r2_t* xi = (r2_t*)(a.xil + i * BS + 2 * qp);
    r2_t xn;
    if (SOR) { const r2_t xo = *xi; xn.x = (1.0 - a.omega) * xo.x + (a.omega / d) * (bb.x - tot.x); xn.y = (1.0 - a.omega) * xo.y + (a.omega / d) * (bb.y - tot.y); }
    else { xn.x = (bb.x - tot.x) / d; xn.y = (bb.y - tot.y) / d; }
    *xi = xn;
    a.x[i + (2 * qp) * a.ldx] = xn.x;
    a.x[i + (2 * qp + 1) * a.ldx] = xn.y;
  }
}

// Long-row slots: composite rows of deeply merged groups have hundreds to ~2000 entries.  Same idea as
// gs_slot_kernel with 2048 entries per slot (4 per thread) and at most 64 rows per slot; a row is summed by a
// whole wave (entries interleaved over the 64 lanes: conflict-free LDS reads, a 1600-entry row is 26 steps +
// a wave reduction instead of 1600 dependent adds), wave w takes rows w, w + 8, ... of the slot.
constexpr int kBigSlot = 2048;
constexpr int kBigRows = 64;

template <bool SOR>
__global__ __launch_bounds__(kSlot) void gs_bigslot_kernel(SlotArgs a) {
  constexpr int EPT = kBigSlot / kSlot, NW = kSlot / kWave;
  __shared__ real s_prod[kBigSlot];
  const int tid = threadIdx.x;
  int lb, cv;
  multi_column_block(a.ncolv, lb, cv);
  if (cv > 0) {
    a.x += (int64_t)cv * a.ldx;
    a.bp += (int64_t)cv * a.ldb;
  }
  if (a.xcd_map) lb = xcd_block(lb, a.nslots);
  if (lb >= a.nslots) return;
  const int s = a.slot0 + lb;
  const int base = s * kBigSlot;
  real v[EPT], xv[EPT];
  int c[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    v[e] = a.wval[base + tid + e * kSlot];
    c[e] = a.wcol[base + tid + e * kSlot];
  }
  const int r0 = a.slot_row[2 * s], nrows = a.slot_row[2 * s + 1] - r0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) xv[e] = a.x[c[e]];
  const int wv = tid / kWave, ln = tid % kWave;
  // row data of this wave's first two rows, requested with the gathers
  i4_t m[2];
  real d[2], bb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = wv + u * NW;
    m[u] = i4_t{0, 0, -1, 0}; d[u] = 0.0; bb[u] = 0.0;
    if (r < nrows) { m[u] = a.wmeta[r0 + r]; d[u] = a.diag[r0 + r]; bb[u] = a.bp[r0 + r]; }
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) s_prod[tid + e * kSlot] = v[e] * xv[e];
  __syncthreads();
  int u = 0;
  for (int r = wv; r < nrows; r += NW, ++u) {
    i4_t mm;
    real dd, bv;
    if (u < 2) { mm = m[u]; dd = d[u]; bv = bb[u]; }
    else { mm = a.wmeta[r0 + r]; dd = a.diag[r0 + r]; bv = a.bp[r0 + r]; }
    const int dz = mm.z - base, qe = mm.y - base;
    real acc = 0.0;
    for (int q = mm.x - base + ln; q < qe; q += kWave)
      if (q != dz) acc += s_prod[q];
    acc = tree_add<kWave>(acc);
    if (ln == 0 && dd != 0.0) {
      const int i = r0 + r;
      a.x[i] = SOR ? (1.0 - a.omega) * a.x[i] + (a.omega / dd) * (bv - acc) : (bv - acc) / dd;
    }
  }
}

// ---- block-inverse Gauss-Seidel for small, densely coupled operators ------------------------
// Coarse AMG levels have few rows but long rows (60-124 nonzeros) and therefore almost as many
// dependency levels as rows: level scheduling degenerates to ~1 row per step.  For those operators
// the sweep is re-blocked by ROW INDEX: with B = kBlk consecutive rows per block,
//   forward:  x_I <- T_II^{-1} ( b_I - sum_{j not in-block-lower} a_ij x_j ),  T_II = (D + L)_II
// visits the blocks in order, so the result is the lexicographic Gauss-Seidel iterate; the
// in-block triangular solve is one dense B x B product with the pre-inverted block (rounding
// differs from the scalar recurrence at the 1e-15 level; the exact-order path stays available,
// amgh_debug_set_tunable("gs_block_inverse", 0)).  n/B sequential steps instead of ~n.
constexpr int kBlk = 128;
constexpr int kBlkThreads = 1024;  // 128 rows x 8 partial sums in the dense phase
constexpr int kBlkLds = 8192;
constexpr int kBlkSingle = 16;     // operators with at most this many blocks: one gs_block_kernel launch per sweep, no split

struct BlockArgs {
  const int32_t* rowptr;  // "outer" matrix of this direction: the operator minus the in-block triangle
  const int32_t* col;     //   (forward: minus in-block entries with col <= row; backward: col >= row)
  const real* val;
  const real* tinv;     // nblk x kBlk x kBlk row-major: inverse of the in-block triangle (+ diagonal)
  const real* diag;     // diagonal of each row, 0 if absent (such rows keep x, smoother.jl:87)
  real* x;
  const real* b;
  int32_t n;
  int32_t nblk;      // blocks swept by this launch: blk0 .. blk0 + nblk - 1 (descending when backward)
  int32_t blk0;
  int32_t backward;
  int64_t ld;  // multi-RHS: workgroup blockIdx.x sweeps column blockIdx.x of x / b
  // "near" entries: outer entries of block k that reference the block swept just before it (gs_block_pipe_kernel)
  const int32_t* near_ptr;  // nblk + 1
  const i2_t* near_pi;      // {position inside the block's outer range, column - first row of the previous block}
  const real* near_val;
  unsigned long long* tim;  // diagnostics (amgh_debug_chain_timing): per-phase shader-cycle sums, or nullptr
};

// this thread's 16 entries of the block inverse (row drow, columns part*16..+15); the half of the
// block that is structurally zero (above the diagonal forward, below it backward) is not read
__device__ __forceinline__ void blk_load_tinv(const BlockArgs& a, int blk, int drow, int part, real (&tv)[16]) {
  const bool nz = a.backward ? (part * 16 + 15 >= drow) : (part * 16 <= drow);
  if (nz) {
    const d2_t* tp = (const d2_t*)(a.tinv + ((size_t)blk * kBlk + drow) * kBlk + part * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const d2_t v = tp[e]; tv[2 * e] = v.x; tv[2 * e + 1] = v.y; }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) tv[e] = 0.0;
  }
}

__global__ __launch_bounds__(kBlkThreads) void gs_block_kernel(BlockArgs a) {
  __shared__ real s_prod[kBlkLds];
  __shared__ real s_vec[kBlk];
  const int tid = threadIdx.x;
  a.x += blockIdx.x * a.ld;
  a.b += blockIdx.x * a.ld;
  const int drow = tid >> 3, part = tid & 7;  // dense phase: 8 lanes share a row, 16 columns each
  real tv[16];
  blk_load_tinv(a, a.blk0 + (a.backward ? a.nblk - 1 : 0), drow, part, tv);
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int step = 0; step < a.nblk; ++step) {
    const unsigned long long t0 = a.tim ? clock64() : 0;
    unsigned long long t1 = 0, t2 = 0, t3 = 0, t1a = 0, t1b = 0;
    const int blk = a.blk0 + (a.backward ? a.nblk - 1 - step : step);
    const int i0 = blk * kBlk;
    const int rows = min(kBlk, a.n - i0);
    // phase 1: s_i = b_i - (outer row i) . x      (all x entries referenced are final or old)
    const int p0 = a.rowptr[i0], p1 = a.rowptr[i0 + rows];
    // row drow is summed by its 8 lanes, entries interleaved (lane p: p, p+8, ...): conflict-free LDS reads.
    // (One thread per row walking its own segment put 64 lanes on ~16 banks: 12 k of the 28 k cycles per step.)
    int rs = 0, re = 0;
    real d = 0.0, bb = 0.0, xo = 0.0, acc = 0.0;
    if (drow < rows) {
      rs = a.rowptr[i0 + drow]; re = a.rowptr[i0 + drow + 1];
      if (part == 0) { d = a.diag[i0 + drow]; bb = a.b[i0 + drow]; xo = a.x[i0 + drow]; }
    }
    for (int c0 = p0; c0 < p1; c0 += kBlkLds) {
      const int c1 = min(c0 + kBlkLds, p1);
      // one CU streams the whole operator: keep 4 independent (val, col) -> x chains in flight per thread
      int k = c0 + tid;
      for (; k + 3 * kBlkThreads < c1; k += 4 * kBlkThreads) {
        real v[4], xv[4];
        int c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a.val[k + e * kBlkThreads]; c[e] = a.col[k + e * kBlkThreads]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = a.x[c[e]];
#pragma unroll
        for (int e = 0; e < 4; ++e) s_prod[k + e * kBlkThreads - c0] = v[e] * xv[e];
      }
      for (; k < c1; k += kBlkThreads) s_prod[k - c0] = a.val[k] * a.x[a.col[k]];
      __syncthreads();
      if (a.tim && c0 == p0) t1 = clock64();
      for (int q = max(rs, c0) - c0 + part, qe = min(re, c1) - c0; q < qe; q += 8) acc += s_prod[q];
      if (c1 < p1) __syncthreads();
    }
    if (a.tim) t1a = clock64();
    acc += __shfl_xor(acc, 1, kWave);
    acc += __shfl_xor(acc, 2, kWave);
    acc += __shfl_xor(acc, 4, kWave);
    if (a.tim) t1b = clock64();
    if (part == 0) s_vec[drow] = (drow < rows) ? ((d == 0.0) ? xo : bb - acc) : 0.0;
    __syncthreads();
    if (a.tim) t2 = clock64();
    // phase 2: x_I = Tinv_II . s
    real sum = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += tv[e] * s_vec[part * 16 + e];
    sum += __shfl_xor(sum, 1, kWave);
    sum += __shfl_xor(sum, 2, kWave);
    sum += __shfl_xor(sum, 4, kWave);
    // the next block's inverse streams in while this step finishes and the next one starts
    if (step + 1 < a.nblk) blk_load_tinv(a, a.backward ? blk - 1 : blk + 1, drow, part, tv);
    if (a.tim) t3 = clock64();
    if (part == 0 && drow < rows) a.x[i0 + drow] = sum;
    __threadfence_block();
    __syncthreads();
    if (a.tim) {
      const unsigned long long t4 = clock64();
      tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2; tacc[3] += t4 - t3; tacc[4] += 1;
      tacc[5] += t1a - t1; tacc[6] += t1b - t1a; tacc[7] += t2 - t1b;
    }
  }
  if (a.tim && tid == 0 && blockIdx.x == 0) {
    // [0] loads -> products in LDS, [1] in-order row sums, [2] dense product, [3] x store + fence + barrier, [4] steps
    for (int q = 0; q < 8; ++q) atomicAdd(a.tim + q, tacc[q]);
  }
}

// Software-pipelined block sweep.  gs_block_kernel spends most of a 128-row block step waiting: row pointers ->
// col/val -> x gather are dependent round trips (and a block has more entries than one round of loads covers),
// then the x store is fenced (measured: 10.3 k + 7.6 k + 4.1 k of 26.6 k cycles per step on the 38 k-row level of
// the 256^3 hierarchy, `tools/block_phase.py`).  Here the products of block k+1 are produced WHILE block k is
// being finished, into a second LDS buffer:
//   (a) issue block k+1's col/val, row metadata, near list and the ranges two blocks ahead
//   (b) patch block k's "near" entries (those that reference block k-1, whose x was not final when the products
//       were formed; a host-built list, a few per cent of the entries) from the LDS copy of x_{k-1}
//   (c) row sums of block k, 8 lanes per row, entries interleaved (conflict-free LDS reads)
//   (d) issue block k+1's x gathers      (e) dense product with the inverted triangle, store x_k
//   (f) block k+1's products -> the other LDS buffer
// Barriers inside the step order LDS only (lds_barrier), so the loads of (a)/(d) stay in flight across them; the
// step ends with a full barrier so that every x store is complete before the next gathers are issued.
constexpr int kPipeThreads = 1024;
constexpr int kPipePF = 8;                          // leading entries per thread: 2 rounds of 4 consecutive entries (16-B loads)
constexpr int kPipeCap = kPipePF * kPipeThreads;    // products per LDS buffer (2 x 64 KiB); longer blocks: extra passes
constexpr int kPipeNPF = 2;                         // near entries per thread held in registers

// The leading entries [p0, lim) of a block, 4 consecutive entries per thread and round, from the 16-B aligned
// position a0 = p0 & ~3 (a single CU is bound by the number of vector-memory instructions it can issue, not by
// bytes: one 16-B load instead of four 4-B loads).  Entries outside [p0, lim) get col = -1.
struct PipeEntries {
  int c[kPipePF];
  real v[kPipePF];
};
__device__ __forceinline__ void pipe_load_entries(const BlockArgs& a, int p0, int lim, int tid, PipeEntries& o) {
  const int a0 = p0 & ~3;
#pragma unroll
  for (int r = 0; r < kPipePF / 4; ++r) {
    const int k = a0 + 4 * (tid + r * kPipeThreads);
    if (k < lim) {  // hipMalloc'ed arrays are padded past nnz? no: guard the tail below
      if (k + 4 <= lim && k >= p0) {
        const i4_t cc = *(const i4_t*)(a.col + k);
        const d2_t v0 = *(const d2_t*)(a.val + k);
        const d2_t v1 = *(const d2_t*)(a.val + k + 2);
        o.c[4 * r] = cc.x; o.c[4 * r + 1] = cc.y; o.c[4 * r + 2] = cc.z; o.c[4 * r + 3] = cc.w;
        o.v[4 * r] = v0.x; o.v[4 * r + 1] = v0.y; o.v[4 * r + 2] = v1.x; o.v[4 * r + 3] = v1.y;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool in = k + e >= p0 && k + e < lim;
          o.c[4 * r + e] = in ? a.col[k + e] : -1;
          o.v[4 * r + e] = in ? a.val[k + e] : 0.0;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { o.c[4 * r + e] = -1; o.v[4 * r + e] = 0.0; }
    }
  }
}
__device__ __forceinline__ void pipe_gather(const BlockArgs& a, const PipeEntries& o, real (&px)[kPipePF]) {
#pragma unroll
  for (int e = 0; e < kPipePF; ++e) px[e] = o.c[e] >= 0 ? a.x[o.c[e]] : 0.0;
}
__device__ __forceinline__ void pipe_store_products(real* s_dst, int p0, int tid, const PipeEntries& o,
                                                    const real (&px)[kPipePF]) {
  const int a0 = p0 & ~3;
#pragma unroll
  for (int r = 0; r < kPipePF / 4; ++r) {
    const int k = a0 + 4 * (tid + r * kPipeThreads) - p0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (o.c[4 * r + e] >= 0) s_dst[k + e] = o.v[4 * r + e] * px[4 * r + e];
  }
}
// entries of [p0, p1) the leading rounds cover (they start at the aligned position below p0)
__device__ __forceinline__ int pipe_lead_end(int p0, int p1) { return min(p1, (p0 & ~3) + kPipeCap); }

__global__ __launch_bounds__(kPipeThreads) void gs_block_pipe_kernel(BlockArgs a) {
  constexpr int T = kPipeThreads, PF = kPipePF, NPF = kPipeNPF;
  __shared__ real s_buf[2][kPipeCap];
  __shared__ real s_vec[kBlk];
  __shared__ real s_xnew[kBlk];
  const int tid = threadIdx.x;
  a.x += blockIdx.x * a.ld;
  a.b += blockIdx.x * a.ld;
  const int drow = tid >> 3, part = tid & 7;  // 8 lanes per row, in the row sums and in the dense product
  const int dir = a.backward ? -1 : 1;
  int blk = a.blk0 + (a.backward ? a.nblk - 1 : 0);
  real tv[16];
  blk_load_tinv(a, blk, drow, part, tv);

  // state of the block about to be processed (filled one step ahead)
  int np0, np1, nrs = 0, nre = 0;
  real nd = 0.0, nbb = 0.0, nxo = 0.0;
  i2_t qpi[NPF];
  real qv[NPF];
  int nq0 = 0, nq1 = 0;
  // ranges of the block after that (fetched two steps ahead)
  int n2p0 = 0, n2p1 = 0, n2q0 = 0, n2q1 = 0;
  {
    const int i0 = blk * kBlk, rows = min(kBlk, a.n - i0);
    np0 = a.rowptr[i0];
    np1 = a.rowptr[i0 + rows];
    if (drow < rows) {
      nrs = a.rowptr[i0 + drow]; nre = a.rowptr[i0 + drow + 1];
      if (part == 0) { nd = a.diag[i0 + drow]; nbb = a.b[i0 + drow]; nxo = a.x[i0 + drow]; }
    }
#pragma unroll
    for (int e = 0; e < NPF; ++e) { qpi[e] = i2_t{-1, 0}; qv[e] = 0.0; }  // the first block has no predecessor
    PipeEntries pe;
    real px[PF];
    pipe_load_entries(a, np0, pipe_lead_end(np0, np1), tid, pe);
    pipe_gather(a, pe, px);
    pipe_store_products(s_buf[0], np0, tid, pe, px);
    if (a.nblk > 1) {
      const int j = blk + dir, j0 = j * kBlk, jr = min(kBlk, a.n - j0);
      n2p0 = a.rowptr[j0]; n2p1 = a.rowptr[j0 + jr];
      n2q0 = a.near_ptr[j]; n2q1 = a.near_ptr[j + 1];
    }
  }
  __syncthreads();

  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int step = 0; step < a.nblk; ++step, blk += dir) {
    const unsigned long long t0 = a.tim ? clock64() : 0;
    unsigned long long t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    real* s_cur = s_buf[step & 1];
    real* s_nxt = s_buf[(step & 1) ^ 1];
    const int i0 = blk * kBlk;
    const int rows = min(kBlk, a.n - i0);
    const int p0 = np0, p1 = np1, rs = nrs, re = nre, q0 = nq0, q1 = nq1;
    const real d = nd, bb = nbb, xo = nxo;
    const bool has_next = step + 1 < a.nblk;
    const int lead = pipe_lead_end(p0, p1) - p0;  // entries [0, lead) of this block are in s_cur
    // (b) near entries of this block, from the registers filled a step ago
#pragma unroll
    for (int e = 0; e < NPF; ++e)
      if (qpi[e].x >= 0 && qpi[e].x < lead) s_cur[qpi[e].x] = qv[e] * s_xnew[qpi[e].y];
    for (int i = q0 + NPF * T + tid; i < q1; i += T) {  // more near entries than registers hold
      const i2_t pi = a.near_pi[i];
      if (pi.x < lead) s_cur[pi.x] = a.near_val[i] * s_xnew[pi.y];
    }
    // (a) everything the next block needs
    PipeEntries pe;
    if (has_next) {
      const int j = blk + dir, j0 = j * kBlk, jr = min(kBlk, a.n - j0);
      np0 = n2p0; np1 = n2p1; nq0 = n2q0; nq1 = n2q1;
      pipe_load_entries(a, np0, pipe_lead_end(np0, np1), tid, pe);
      nrs = nre = 0; nd = nbb = nxo = 0.0;
      if (drow < jr) {
        nrs = a.rowptr[j0 + drow]; nre = a.rowptr[j0 + drow + 1];
        if (part == 0) { nd = a.diag[j0 + drow]; nbb = a.b[j0 + drow]; nxo = a.x[j0 + drow]; }
      }
#pragma unroll
      for (int e = 0; e < NPF; ++e) {
        const int i = nq0 + tid + e * T;
        qpi[e] = i2_t{-1, 0}; qv[e] = 0.0;
        if (i < nq1) { qpi[e] = a.near_pi[i]; qv[e] = a.near_val[i]; }
      }
      if (step + 2 < a.nblk) {
        const int m = blk + 2 * dir, m0 = m * kBlk, mr = min(kBlk, a.n - m0);
        n2p0 = a.rowptr[m0]; n2p1 = a.rowptr[m0 + mr];
        n2q0 = a.near_ptr[m]; n2q1 = a.near_ptr[m + 1];
      }
    } else {
#pragma unroll
      for (int e = 0; e < PF; ++e) { pe.c[e] = -1; pe.v[e] = 0.0; }
    }
    lds_barrier();
    if (a.tim) t1 = clock64();
    // (c) row sums: 8 lanes per row, interleaved
    real acc = 0.0;
    for (int q = rs - p0 + part, qe = min(re - p0, lead); q < qe; q += 8) acc += s_cur[q];
    for (int c0 = p0 + lead; c0 < p1; c0 += kPipeCap) {  // block longer than the buffer: fetched on demand
      const int c1 = min(c0 + kPipeCap, p1);
      lds_barrier();
#pragma unroll
      for (int e = 0; e < PF; ++e) {
        const int k = c0 + tid + e * T;
        if (k < c1) s_cur[k - c0] = a.val[k] * a.x[a.col[k]];  // x in HBM is current: the last step ended with a full barrier
      }
      lds_barrier();
      for (int q = max(rs, c0) - c0 + part, qe = min(re, c1) - c0; q < qe; q += 8) acc += s_cur[q];
    }
    acc += __shfl_xor(acc, 1, kWave);
    acc += __shfl_xor(acc, 2, kWave);
    acc += __shfl_xor(acc, 4, kWave);
    if (part == 0) s_vec[drow] = (drow < rows) ? ((d == 0.0) ? xo : bb - acc) : 0.0;
    lds_barrier();
    if (a.tim) t2 = clock64();
    // (d) x gathers of the next block's leading entries
    real px[PF];
    pipe_gather(a, pe, px);
    // (e) x_I = Tinv_II . s
    real sum = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += tv[e] * s_vec[part * 16 + e];
    sum += __shfl_xor(sum, 1, kWave);
    sum += __shfl_xor(sum, 2, kWave);
    sum += __shfl_xor(sum, 4, kWave);
    if (a.tim) t3 = clock64();
    if (has_next) blk_load_tinv(a, blk + dir, drow, part, tv);
    if (part == 0 && drow < rows) {
      a.x[i0 + drow] = sum;
      s_xnew[drow] = sum;
    }
    // (f) next block's products
    pipe_store_products(s_nxt, np0, tid, pe, px);
    if (a.tim) t4 = clock64();
    __syncthreads();  // full: x stores complete before the next step's gathers
    if (a.tim) {
      t5 = clock64();
      tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2; tacc[3] += t4 - t3; tacc[4] += 1; tacc[5] += t5 - t4;
    }
  }
  if (a.tim && tid == 0 && blockIdx.x == 0) {
    // [0] near patch + issue loads + barrier, [1] row sums, [2] gathers issue + dense, [3] products of next block, [5] full barrier
    for (int q = 0; q < 8; ++q) atomicAdd(a.tim + q, tacc[q]);
  }
}

// ---- vector kernels ------------------------------------------------------
__global__ void fill_kernel(real* x, int64_t n, real v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void copy_kernel(real* dst, const real* src, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// y = y + alpha * x, alpha read from device memory (scaled by sign)
__global__ void axpy_dev_kernel(real* y, const real* x, const real* alpha, real sign, int64_t n) {
  const real al = sign * alpha[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = y[i] + al * x[i];
}
// u = c + beta * u
__global__ void xpby_dev_kernel(real* u, const real* c, const real* beta, int64_t n) {
  const real be = beta[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) u[i] = c[i] + be * u[i];
}

// wave64 + LDS block reduction of a real
__device__ __forceinline__ real block_reduce_sum(real v, real* s_part) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  if (lane == 0) s_part[w] = v;
  __syncthreads();
  real r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + kWave - 1) / kWave;
    for (int i = 0; i < nw; ++i) r += s_part[i];
  }
  return r;  // valid on thread 0
}

constexpr int kRedBlocks = 1024;
// partial[b] = sum_i x[i]*y[i] over this block's grid-stride slice (deterministic)
__global__ __launch_bounds__(kThreads) void dot_partial_kernel(const real* x, const real* y, int64_t n, real* partial) {
  __shared__ real s_part[kThreads / kWave];
  real v = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v += x[i] * y[i];
  const real r = block_reduce_sum(v, s_part);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// out[0] = op(sum partial); op: 0 identity, 1 sqrt
__global__ __launch_bounds__(kThreads) void reduce_final_kernel(const real* partial, int np, real* out, int op) {
  __shared__ real s_part[kThreads / kWave];
  real v = 0.0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) v += partial[i];
  const real r = block_reduce_sum(v, s_part);
  if (threadIdx.x == 0) out[0] = op ? sqrt(r) : r;
}
// tiny scalar programs for the device-resident PCG recurrence
// op 0: out = a / b ; op 1: out = a (copy)
__global__ void scalar_kernel(real* out, const real* a, const real* b, int op) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (op == 0) ? a[0] / b[0] : a[0];
}

// ---- the device-resident PCG recurrence in fewer launches (round 6) ----------------------------------------------------------
// Between two V-cycles an iteration of pcg_dev was 15 launches (memset, copy of rho, two-stage dots, scalar divisions, xpby,
// SpMV, two axpys, two-stage norm).  The same sums, products
// and quotients in the same order (bitwise the unfused recurrence), fused:
//   pcg_scal_kernel      the second stage of a dot product + the scalar step that consumes it
//   pcg_update_kernel    x += alpha u, r -= alpha c, c = 0 (the next cycle's x = 0), first stage of |r|^2
// 8 launches.  C5: 13 iterations 3.16 -> 3.05 ms; 160^3: 7.18 -> 7.12 ms per iteration.  (Everything between two cycles in ONE
// workgroup was built for small operators and measured: C5 3.13 ms, 13 824 rows 0.79 instead of 0.55 ms per iteration — the
// launches of the recurrence queue behind the cycle and cost ~2 us each, a one-workgroup SpMV costs more; removed.)
// scal: [0] |r|, [1] rho, [2] rho_prev, [3] alpha, [4] beta, [5] u.c
// which = 0: rho_prev = rho, rho = sum, beta = rho / rho_prev; which = 1: u.c = sum, alpha = rho / u.c; which = 2: |r| = sqrt(sum)
__global__ __launch_bounds__(kThreads) void pcg_scal_kernel(const real* partial, int np, real* scal, int which) {
  __shared__ real s_part[kThreads / kWave];
  real v = 0.0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) v += partial[i];
  const real r = block_reduce_sum(v, s_part);
  if (threadIdx.x == 0) {
    if (which == 0) { const real rp = scal[1]; scal[2] = rp; scal[1] = r; scal[4] = r / rp; }
    else if (which == 1) { scal[5] = r; scal[3] = scal[1] / r; }
    else scal[0] = sqrt(r);
  }
}
__global__ __launch_bounds__(kThreads) void pcg_update_kernel(real* x, const real* u, real* r, real* c, const real* alpha, int64_t n,
                                                                real* partial) {
  __shared__ real s_part[kThreads / kWave];
  const real ap = 1.0 * alpha[0], am = -1.0 * alpha[0];
  real v = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    x[i] = x[i] + ap * u[i];
    const real rn = r[i] + am * c[i];
    r[i] = rn;
    c[i] = 0.0;
    v += rn * rn;
  }
  const real t = block_reduce_sum(v, s_part);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// ---- dense triangular inverse (small operators) -------------------------------------------------------------------
// X = (D + L)^-1 (upper = 0) or (D + U)^-1 (upper = 1) of a diagonal block of a CSR matrix, dense n x n ROW-major: one thread per column c
// runs the substitution T X[:, c] = e_c down (up) the rows; a column depends on itself only, and the threads of a
// workgroup walk the rows together, so the row's entries are one broadcast load.  Xf and Xb may be the SAME array:
// the lower triangle of a block's square holds (D + L)^-1, the upper (D + U)^-1, and the diagonal they share is
// 1 / d_ii in both (written twice with the same bits); every read stays inside the column's own triangle.
__global__ void tri_inverse_kernel(const int32_t* rowptr, const int32_t* col, const real* val, int B, int ntot,
                                   const int64_t* off, real* Xf, real* Xb) {
  // blockIdx.y = diagonal block k (rows / columns [k B, min(ntot, (k + 1) B))), blockIdx.z = 0: (D + L)^-1, 1: (D + U)^-1 —
  // all blocks and both triangles in ONE launch (each is a chain of n dependent row steps: they only pay side by side).
  // Entries outside the block are not part of its triangle.
  const int row0 = blockIdx.y * B, n = min(B, ntot - row0), upper = blockIdx.z;
  real* X = (upper ? Xb : Xf) + off[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int c_first = blockIdx.x * blockDim.x, c_last = min(n, c_first + (int)blockDim.x) - 1;
  const int s0 = upper ? n - 1 - c_last : c_first;     // rows before the workgroup's first column are zero in every column
  for (int s = s0; s < n; ++s) {
    const int i = upper ? n - 1 - s : s;
    const bool act = upper ? i <= c : i >= c;
    real acc = (i == c) ? 1.0 : 0.0, d = 0.0;
    for (int32_t j = rowptr[row0 + i]; j < rowptr[row0 + i + 1]; ++j) {
      const int32_t cj = col[j] - row0;
      const real v = val[j];
      if (cj == i) d = v;
      else if (act && cj >= 0 && cj < n && (upper ? (cj > i && cj <= c) : (cj < i && cj >= c))) acc -= v * X[(size_t)cj * n + c];
    }
    if (act) X[(size_t)i * n + c] = acc / d;
  }
}
// out[i] = sum_j |X[i, j]| (inf-norm rows of the inverse: the condition estimate of the triangle)
__global__ void dense_abs_rowsum_kernel(const real* Xf, const real* Xb, int B, int ntot, const int64_t* off, real* out_f,
                                        real* out_b) {
  // blockIdx.y = diagonal block, blockIdx.z = triangle; one wavefront per row of the block
  const int row0 = blockIdx.y * B, n = min(B, ntot - row0);
  const real* X = (blockIdx.z ? Xb : Xf) + off[blockIdx.y];
  const int i = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave, ln = threadIdx.x % kWave;
  if (i >= n) return;
  real acc = 0.0;
  const int lo = blockIdx.z ? i : 0, hi = blockIdx.z ? n : i + 1;   // this triangle's part of the (possibly shared) square
  for (int j = lo + ln; j < hi; j += kWave) acc += fabs(X[(size_t)i * n + j]);
  for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
  if (ln == 0) (blockIdx.z ? out_b : out_f)[row0 + i] = acc;
}
// x = X s with X lower (upper = 0: columns 0..i of row i) or upper triangular, row-major: one 256-thread workgroup per
// row, threads across the columns (coalesced; a thread owns every 256th entry, so all its loads are independent and in
// flight together — a wavefront per row walked a 4 096-entry row in 64 dependent round trips), then a fixed
// shuffle tree and the four wave partials added in order (deterministic).  gridDim.y = right-hand-side columns.
// NC = right-hand-side columns per workgroup (blocks of right-hand sides: the row of X is read ONCE for NC columns — a
// workgroup per column re-read the dense inverse NC times; every column's sum is formed exactly as with NC = 1).
// gridDim.y = column groups.
template <int NC>
__global__ __launch_bounds__(kThreads) void tri_gemv_kernel(const real* X, const real* s, real* x, int n, int upper, int64_t lds,
                                                              int64_t ldx) {
  __shared__ real s_part[NC][kThreads / kWave];
  // (the longest rows first: workgroups are dispatched in blockIdx order, and a launch that ends on its 32 KB rows ends later than
  // one that ends on its 8-byte rows — lower triangle: row n - 1 first; upper: row 0 first)
  const int i = upper ? (int)blockIdx.x : n - 1 - (int)blockIdx.x, tid = threadIdx.x;
  const real* sv = s + (int64_t)blockIdx.y * NC * lds;
  const real* row = X + (size_t)i * n;
  // a thread's entries are a function of the COLUMN alone — lower: j = tid, tid + 256, ... up to the diagonal; upper: j = n - 1 -
  // tid, n - 1 - tid - 256, ... down to it — so that tri_gemm_kernel (several rows per workgroup sharing the loads of s) forms
  // every sum from the same partial sums in the same order
  real acc[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) acc[q] = 0.0;
  if (!upper) {
    for (int j = tid; j <= i; j += kThreads) {
      const real r = row[j];
#pragma unroll
      for (int q = 0; q < NC; ++q) acc[q] += r * sv[j + q * lds];
    }
  } else {
    for (int j = n - 1 - tid; j >= i; j -= kThreads) {
      const real r = row[j];
#pragma unroll
      for (int q = 0; q < NC; ++q) acc[q] += r * sv[j + q * lds];
    }
  }
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    for (int o = kWave / 2; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, kWave);
    if (tid % kWave == 0) s_part[q][tid / kWave] = acc[q];
  }
  __syncthreads();
  if (tid < NC) {
    real t = s_part[tid][0];
#pragma unroll
    for (int w = 1; w < kThreads / kWave; ++w) t += s_part[tid][w];
    x[((int64_t)blockIdx.y * NC + tid) * ldx + i] = t;
  }
}

// The same product for a block of NC right-hand sides with RB consecutive rows per workgroup (round 6): with one row per
// workgroup every entry of X costs NC loads of s beside its own — 9 x the row's bytes through L2 at NC = 8, 38.6 us per launch
// of a 4 096-row block where a single column takes 13.8.  RB rows share a thread's NC loads of s (registers): RB + NC loads per
// RB * NC multiply-adds.  The partial sums, the tree and the order of the wave partials are tri_gemv_kernel's: bit for bit its
// result per column.
template <int NC, int RB>
__global__ __launch_bounds__(kThreads) void tri_gemm_kernel(const real* X, const real* s, real* x, int n, int upper, int64_t lds,
                                                              int64_t ldx) {
  __shared__ real s_part[RB * NC][kThreads / kWave];
  const int i0 = (upper ? (int)blockIdx.x : (int)(gridDim.x - 1 - blockIdx.x)) * RB, tid = threadIdx.x;   // (the longest rows first, as tri_gemv_kernel)
  const real* sv = s + (int64_t)blockIdx.y * NC * lds;
  real acc[RB][NC];
#pragma unroll
  for (int rr = 0; rr < RB; ++rr)
#pragma unroll
    for (int q = 0; q < NC; ++q) acc[rr][q] = 0.0;
  const int ilast = min(i0 + RB, n) - 1;   // the last row of this workgroup
  const int step = upper ? -kThreads : kThreads;
  for (int j = upper ? n - 1 - tid : tid; upper ? j >= i0 : j <= ilast; j += step) {
    real sq[NC], r[RB];
    bool ok[RB];
#pragma unroll
    for (int q = 0; q < NC; ++q) sq[q] = sv[j + q * lds];
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) {
      const int i = i0 + rr;
      ok[rr] = i < n && (upper ? j >= i : j <= i);
      r[rr] = ok[rr] ? X[(size_t)i * n + j] : 0.0;
    }
#pragma unroll
    for (int rr = 0; rr < RB; ++rr)
      if (ok[rr]) {
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[rr][q] += r[rr] * sq[q];
      }
  }
#pragma unroll
  for (int rr = 0; rr < RB; ++rr)
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      real v = acc[rr][q];
      for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
      if (tid % kWave == 0) s_part[rr * NC + q][tid / kWave] = v;
    }
  __syncthreads();
  if (tid < RB * NC) {
    const int rr = tid / NC, q = tid % NC, i = i0 + rr;
    if (i < n) {
      real t = s_part[tid][0];
#pragma unroll
      for (int w = 1; w < kThreads / kWave; ++w) t += s_part[tid][w];
      x[((int64_t)blockIdx.y * NC + q) * ldx + i] = t;
    }
  }
}

// ---- the collapsed coarse tail (round 6; amghip.hip: tail_dense_build / tail_apply) ----
// Every step of __solve! (multilevel.jl:214-239) is linear in (x, b): from a level of a few thousand rows down, the
// whole recursion — smoothers, residual, restriction, the levels below, the coarse solve, prolongation — IS one dense
// n x n operator, built once by running the library's own cycle on the columns of the identity.  One launch replaces
// the ~25-60 launches of ~5 us of those levels (the reference walks them for free, multilevel.jl:227-231).
// B[c * ld + j0 + c] = 1 for the nb columns of a batch (B zeroed before)
__global__ void unit_cols_kernel(real* B, int64_t ld, int j0, int nb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nb) B[(int64_t)c * ld + j0 + c] = (real)1;
}
// M[i][j] (row-major) = Mc[pj * n + pi] (columns of the natural-order operator), pi = perm ? perm[i] : i — the operator between
// the level-ordered vectors of a level that takes its right-hand side / leaves its x in its schedule's order
__global__ void tail_transpose_kernel(const real* __restrict__ Mc, const int32_t* __restrict__ perm, real* __restrict__ M, int n) {
  __shared__ real tile[32][33];
  const int bj = blockIdx.x * 32, bi = blockIdx.y * 32;
  // read: rows of Mc (= columns j of the operator), coalesced along i
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int j = bj + r, i = bi + threadIdx.x;
    if (j < n && i < n) tile[r][threadIdx.x] = Mc[(size_t)(perm ? perm[j] : j) * n + (perm ? perm[i] : i)];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int i = bi + r, j = bj + threadIdx.x;
    if (i < n && j < n) M[(size_t)i * n + j] = tile[threadIdx.x][r];
  }
}
// x (+)= M b, M dense n x n ROW-major: one workgroup of T threads per row, threads across the columns (a thread owns every
// T-th entry: all its loads independent and in flight together), a fixed shuffle tree, the wave partials added in order:
// deterministic, and per column the same sum whatever NC is (NC right-hand sides share the row's loads).
template <int NC, int T, bool ACC>
__global__ __launch_bounds__(T) void dense_rm_gemv_kernel(const real* __restrict__ M, const real* __restrict__ b, real* __restrict__ x, int n,
                                                          int64_t ldb, int64_t ldx) {
  __shared__ real s_part[NC][T / kWave];
  const int i = blockIdx.x, tid = threadIdx.x;
  const real* bv = b + (int64_t)blockIdx.y * NC * ldb;
  const real* row = M + (size_t)i * n;
  real acc[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) acc[q] = 0.0;
  for (int j = tid; j < n; j += T) {
    const real r = row[j];
#pragma unroll
    for (int q = 0; q < NC; ++q) acc[q] += r * bv[j + q * ldb];
  }
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    for (int o = kWave / 2; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, kWave);
    if (tid % kWave == 0) s_part[q][tid / kWave] = acc[q];
  }
  __syncthreads();
  if (tid < NC) {
    real t = s_part[tid][0];
#pragma unroll
    for (int w = 1; w < T / kWave; ++w) t += s_part[tid][w];
    real* xo = x + ((int64_t)blockIdx.y * NC + tid) * ldx + i;
    if (ACC) *xo += t; else *xo = t;
  }
}

// x = M * b, M dense n x n column-major (coarse solve, coarse_solver.jl:16).
// One thread per output row, columns ascending.
__global__ void dense_gemv_kernel(const real* M, const real* b, real* x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  real acc = 0.0;
  for (int j = 0; j < n; ++j) acc += M[i + (size_t)j * n] * b[j];
  x[i] = acc;
}

}  // namespace amgh
