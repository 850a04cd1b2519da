// amghip_kernels.hpp — gfx950 (CDNA4, wave64) device kernels of the AMG solve phase.
//
// Everything here is HBM-bound sparse/stream work (0.135 flop/B on the fine
// Poisson SpMV), so no MFMA: the design rules are coalesced streaming of the CSR
// arrays, LDS-staged products with a sequential per-row sum (which also makes the
// result bit-reproducible and equal to a scalar CPU loop), wavefront __shfl
// reductions for norms/dots, XCD-contiguous row-block mapping for L2 locality.
//
// Arithmetic is kept un-contracted (no FMA fusion across the product and the
// running sum: the product is rounded when it is staged in LDS), so a row's sum
// is the same IEEE sequence the reference's scalar loops execute
// (smoother.jl:81-86, SparseArrays mul!).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amgh {

constexpr int kWave = 64;
constexpr int kThreads = 256;        // 4 waves / workgroup
constexpr int kRowsPerThread = 2;    // rows per thread in the stream kernel
constexpr int kRowsPerBlock = kThreads * kRowsPerThread;
constexpr int kLdsNnz = 4096;        // products staged per pass (32 KiB + skew)
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
  const double* val;
  const double* x;     // gather source (length ncols)
  double* y;           // output
  const double* b;     // RESID / smoothers
  const int32_t* dpos; // position of the diagonal entry of each row, -1 if absent
  const double* diag;  // diagonal value of each row (0 if absent)
  const int32_t* perm; // GS/SOR: original row id of permuted row p (nullptr = identity)
  double omega;
  int32_t row_begin;   // rows [row_begin, row_end) of the matrix are processed
  int32_t row_end;
};

// LDS index skew: breaks the power-of-two strides of rows with 8/16/32 entries
// (ds_read_b64 banks = (addr/4) mod 64; one pad slot per 32 keeps a 32-lane
// group conflict-free for row lengths 7, 8, 16, 27).
__device__ __forceinline__ int skew(int k) { return k + (k >> 5); }

// XCD-contiguous block mapping (workgroup b is observed to run on XCD b % 8):
// give each XCD a contiguous eighth of the row blocks so that the +-nx rows'
// x entries are re-used out of that XCD's own L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_block(int b, int nb) {
  const int per = (nb + kNumXcd - 1) / kNumXcd;
  const int lb = (b % kNumXcd) * per + b / kNumXcd;
  return lb;  // may be >= nb for the ragged tail: caller must bounds-check
}

// "CSR-stream": a workgroup owns kRowsPerBlock consecutive rows.  Their nonzeros
// form one contiguous range of col/val, loaded with fully coalesced accesses;
// products val*x[col] are staged in LDS; each thread then sums its own rows'
// segments sequentially in index order.  Row ranges with more products than fit
// in LDS are processed in several passes (any row length is handled).
template <int MODE>
__global__ __launch_bounds__(kThreads) void csr_stream_kernel(StreamArgs a) {
  __shared__ double s_prod[kLdsNnz + (kLdsNnz >> 5) + 2];

  const int nrows = a.row_end - a.row_begin;
  const int nb = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
  const int lb = xcd_block(blockIdx.x, nb);
  if (lb >= nb) return;
  const int r0 = a.row_begin + lb * kRowsPerBlock;
  const int r1 = min(r0 + kRowsPerBlock, a.row_end);
  const int tid = threadIdx.x;

  int rs[kRowsPerThread], re[kRowsPerThread], dp[kRowsPerThread];
  double acc[kRowsPerThread];
#pragma unroll
  for (int q = 0; q < kRowsPerThread; ++q) {
    const int r = r0 + tid + q * kThreads;
    if (r < r1) {
      rs[q] = a.rowptr[r];
      re[q] = a.rowptr[r + 1];
      dp[q] = (MODE >= M_JACOBI) ? a.dpos[r] : -1;
    } else {
      rs[q] = re[q] = 0;
      dp[q] = -1;
    }
    acc[q] = 0.0;
  }
  const int p0 = a.rowptr[r0];
  const int p1 = a.rowptr[r1];

  for (int c0 = p0; c0 < p1; c0 += kLdsNnz) {
    const int c1 = min(c0 + kLdsNnz, p1);
    for (int k = c0 + tid; k < c1; k += kThreads) {
      const double v = a.val[k];
      const int c = a.col[k];
      s_prod[skew(k - c0)] = v * a.x[c];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kRowsPerThread; ++q) {
      const int lo = max(rs[q], c0), hi = min(re[q], c1);
      for (int j = lo; j < hi; ++j) {
        const double p = s_prod[skew(j - c0)];
        if (MODE >= M_JACOBI) {
          if (j != dp[q]) acc[q] += p;
        } else {
          acc[q] += p;
        }
      }
    }
    if (c1 < p1) __syncthreads();
  }

#pragma unroll
  for (int q = 0; q < kRowsPerThread; ++q) {
    const int r = r0 + tid + q * kThreads;
    if (r >= r1) continue;
    if (MODE == M_SPMV) {
      a.y[r] = acc[q];
    } else if (MODE == M_RESID) {
      a.y[r] = a.b[r] - acc[q];
    } else if (MODE == M_ADD) {
      a.y[r] = a.y[r] + acc[q];
    } else if (MODE == M_JACOBI) {
      const double d = a.diag[r];
      const double t = a.x[r];
      const double cand = (1.0 - a.omega) * t + a.omega * ((a.b[r] - acc[q]) / d);
      a.y[r] = (d == 0.0) ? t : cand;
    } else {
      const int i = a.perm ? a.perm[r] : r;
      const double d = a.diag[r];
      if (d != 0.0) {
        if (MODE == M_GS) {
          a.y[i] = (a.b[i] - acc[q]) / d;
        } else {
          a.y[i] = (1.0 - a.omega) * a.y[i] + (a.omega / d) * (a.b[i] - acc[q]);
        }
      }
    }
  }
}

// Single-workgroup chain over consecutive NARROW dependency levels of a
// Gauss-Seidel/SOR sweep: one thread per row, a workgroup barrier between
// dependency levels (visibility of x inside one CU needs only the barrier).
// Used for whole small hierarchy levels and for the thin head/tail of a large
// level's wavefront, where a kernel boundary per dependency level (~1.5 us)
// would dominate.
struct ChainArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const double* val;
  double* x;
  const double* b;
  const int32_t* dpos;
  const double* diag;
  const int32_t* perm;
  const int32_t* lvl_ptr;  // device copy of the dependency-level pointer
  double omega;
  int32_t lvl_begin;       // dependency levels [lvl_begin, lvl_end) in sweep order
  int32_t lvl_end;
  int32_t step;            // +1 forward, -1 backward (then lvl_begin > lvl_end)
};

constexpr int kChainThreads = 1024;

template <bool SOR>
__global__ __launch_bounds__(kChainThreads) void gs_chain_kernel(ChainArgs a) {
  for (int lv = a.lvl_begin; lv != a.lvl_end; lv += a.step) {
    const int s = a.lvl_ptr[lv], e = a.lvl_ptr[lv + 1];
    for (int r = s + (int)threadIdx.x; r < e; r += kChainThreads) {
      const int js = a.rowptr[r], je = a.rowptr[r + 1];
      const int dp = a.dpos[r];
      double acc = 0.0;
      for (int j = js; j < je; ++j) {
        // keep the product un-fused with the running sum
        const double p = __dmul_rn(a.val[j], a.x[a.col[j]]);
        if (j != dp) acc = __dadd_rn(acc, p);
      }
      const double d = a.diag[r];
      const int i = a.perm[r];
      if (d != 0.0) {
        if (SOR)
          a.x[i] = (1.0 - a.omega) * a.x[i] + (a.omega / d) * (a.b[i] - acc);
        else
          a.x[i] = (a.b[i] - acc) / d;
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- vector kernels ------------------------------------------------------
__global__ void fill_kernel(double* x, int64_t n, double v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void copy_kernel(double* dst, const double* src, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// y = y + alpha * x, alpha read from device memory (scaled by sign)
__global__ void axpy_dev_kernel(double* y, const double* x, const double* alpha, double sign, int64_t n) {
  const double al = sign * alpha[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = y[i] + al * x[i];
}
// u = c + beta * u
__global__ void xpby_dev_kernel(double* u, const double* c, const double* beta, int64_t n) {
  const double be = beta[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) u[i] = c[i] + be * u[i];
}

// wave64 + LDS block reduction of a double
__device__ __forceinline__ double block_reduce_sum(double v, double* s_part) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  if (lane == 0) s_part[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + kWave - 1) / kWave;
    for (int i = 0; i < nw; ++i) r += s_part[i];
  }
  return r;  // valid on thread 0
}

constexpr int kRedBlocks = 1024;
// partial[b] = sum_i x[i]*y[i] over this block's grid-stride slice (deterministic)
__global__ __launch_bounds__(kThreads) void dot_partial_kernel(const double* x, const double* y, int64_t n, double* partial) {
  __shared__ double s_part[kThreads / kWave];
  double v = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v += x[i] * y[i];
  const double r = block_reduce_sum(v, s_part);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// out[0] = op(sum partial); op: 0 identity, 1 sqrt
__global__ __launch_bounds__(kThreads) void reduce_final_kernel(const double* partial, int np, double* out, int op) {
  __shared__ double s_part[kThreads / kWave];
  double v = 0.0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) v += partial[i];
  const double r = block_reduce_sum(v, s_part);
  if (threadIdx.x == 0) out[0] = op ? sqrt(r) : r;
}
// tiny scalar programs for the device-resident PCG recurrence
// op 0: out = a / b ; op 1: out = a (copy)
__global__ void scalar_kernel(double* out, const double* a, const double* b, int op) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (op == 0) ? a[0] / b[0] : a[0];
}

// x = M * b, M dense n x n column-major (coarse solve, coarse_solver.jl:16).
// One thread per output row, columns ascending.
__global__ void dense_gemv_kernel(const double* M, const double* b, double* x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int j = 0; j < n; ++j) acc += M[i + (size_t)j * n] * b[j];
  x[i] = acc;
}

}  // namespace amgh
