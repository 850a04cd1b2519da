// spmv_bench.hip — within-process A/B of csr_stream_kernel configurations on the
// headline matrix (3-D Poisson N^3, 7-point, CSR int32/f64), plus a device-copy
// bandwidth reference.  Measurement tool, not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/spmv_bench.hip -o tools/spmv_bench
//   tools/spmv_bench [N=256] [reps=20] [rounds=3]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../algebraicmultigrid.jl_amd/csrc/hip/amghip_kernels.hpp"

using namespace amgh;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void copy16_kernel(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read16_kernel(const double2* __restrict__ a, double* out, size_t n) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}

// thread-per-row reference (no LDS), for comparison only
__global__ void csr_scalar_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                  const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double acc = 0;
  for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) acc += val[j] * x[col[j]];
  y[r] = acc;
}

__device__ int dummy_;
struct Variant { std::string name; void (*launch)(const StreamArgs&, hipStream_t); };

template <class CFG>
void launch_cfg(const StreamArgs& a, hipStream_t st) {
  const int nrows = a.row_end - a.row_begin;
  const int nb = (nrows + CFG::ROWS - 1) / CFG::ROWS;
  const int grid = CFG::XCD ? ((nb + kNumXcd - 1) / kNumXcd) * kNumXcd : nb;
  hipLaunchKernelGGL((csr_stream_kernel<M_SPMV, CFG>), dim3(grid), dim3(CFG::THREADS), 0, st, a);
}

// ---- experiment (round 4): col / val staged by LDS-DMA (global_load_lds_dwordx4: HBM -> LDS without passing registers),
// a lane per row walking its entries out of LDS in stored order (the scalar loop's sum), x gathered row-aligned: for a
// banded operator the 64 lanes of a gather read consecutive x (8 lines per instruction instead of ~35 when lanes hold
// consecutive ENTRIES).  Persistent workgroups, two staging buffers: the next tile's DMA flies during this tile's gathers.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void lds_dma16_nt(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int T, int ROWS, int EMAX, bool NT>
__global__ __launch_bounds__(T) void csr_dma_kernel(StreamArgs a, int ntiles, long nnz) {
  static_assert(ROWS == T, "a lane per row");
  constexpr int NW = T / 64;
  constexpr int NVI = EMAX * 8 / 1024 / NW, NCI = EMAX * 4 / 1024 / NW;   // DMA instructions per wave and tile
  static_assert(NVI * NW * 1024 == EMAX * 8 && NCI * NW * 1024 == EMAX * 4, "EMAX: whole instructions per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  double* val_s = (double*)lds;                          // [2][EMAX]
  int* col_s = (int*)(lds + 2 * EMAX * 8);               // [2][EMAX]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = 0;   // dynamic LDS starts at 0 (no static LDS in this kernel)
  auto issue = [&](int tile, int buf) {
    const int r0 = tile * ROWS;
    const int p0 = __builtin_amdgcn_readfirstlane(a.rowptr[r0]);
    const long a0 = p0 & ~3;
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
      const int inst = wave * NVI + i;
      long e = a0 + ((long)inst * 64 + lane) * 2;          // entry index of this lane's 16 bytes
      if (e > nnz - 2) e = (nnz - 2) & ~1l;                  // (the tail of the arrays: a harmless re-read)
      const unsigned dst = lds_base + (unsigned)buf * (EMAX * 8) + (unsigned)inst * 1024u;
      if (NT) lds_dma16_nt(a.val + e, dst); else lds_dma16(a.val + e, dst);
    }
#pragma unroll
    for (int i = 0; i < NCI; ++i) {
      const int inst = wave * NCI + i;
      long e = a0 + ((long)inst * 64 + lane) * 4;
      if (e > nnz - 4) e = (nnz - 4) & ~3l;
      const unsigned dst = lds_base + 2u * EMAX * 8 + (unsigned)buf * (EMAX * 4) + (unsigned)inst * 1024u;
      if (NT) lds_dma16_nt(a.col + e, dst); else lds_dma16(a.col + e, dst);
    }
  };
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  issue(tile, 0);
  int buf = 0;
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const int r = tile * ROWS + tid;
    const bool act = r < a.row_end;
    const int rs = act ? a.rowptr[r] : 0, re = act ? a.rowptr[r + 1] : 0;
    const int p0 = __builtin_amdgcn_readfirstlane(a.rowptr[tile * ROWS]);
    const int a0 = p0 & ~3;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                          // this tile has landed; everybody is done with the other buffer
    if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x, buf ^ 1);
    const double* vs = val_s + buf * EMAX;
    const int* cs = col_s + buf * EMAX;
    double acc = 0.0;
    for (int j0 = rs; j0 < re; j0 += 8) {
      int c[8]; double v[8], xv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const int j = min(j0 + e, re - 1) - a0; c[e] = cs[j]; v[e] = vs[j]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = a.x[c[e]];
#pragma unroll
      for (int e = 0; e < 8; ++e) if (j0 + e < re) acc += v[e] * xv[e];
    }
    if (act) a.y[r] = acc;
  }
}
template <int T, int EMAX, bool NT, int WGPC>
void launch_dma(const StreamArgs& a, hipStream_t st) {
  const int nrows = a.row_end - a.row_begin;
  const int ntiles = (nrows + T - 1) / T;
  const int grid = std::min(ntiles, 256 * WGPC);
  static long nnz = -1;
  if (nnz < 0) { int v; hipMemcpy(&v, a.rowptr + nrows, 4, hipMemcpyDeviceToHost); nnz = v; }
  const size_t lds = (size_t)2 * EMAX * 12;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)csr_dma_kernel<T, T, EMAX, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL((csr_dma_kernel<T, T, EMAX, NT>), dim3(grid), dim3(T), lds, st, a, ntiles, nnz);
}
#define VD(T, E, NT, W) {"DMA T" #T " EMAX" #E " NT" #NT " WG/CU" #W, launch_dma<T, E, NT, W>}

// variant A: one tile per workgroup (no persistence, one staging buffer): the overlap comes from the other workgroups of the CU
template <int T, int RPT, int EMAX, bool NT>
__global__ __launch_bounds__(T) void csr_dma1_kernel(StreamArgs a, long nnz) {
  constexpr int NW = T / 64, ROWS = T * RPT;
  constexpr int NVI = EMAX * 8 / 1024 / NW, NCI = (EMAX * 4 / 1024 + NW - 1) / NW;
  static_assert(NVI * NW * 1024 == EMAX * 8, "EMAX: whole instructions per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const double* vs = (const double*)lds;
  const int* cs = (const int*)(lds + EMAX * 8);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * ROWS;
  const int p0 = __builtin_amdgcn_readfirstlane(a.rowptr[r0]);
  const long a0 = p0 & ~3;
#pragma unroll
  for (int i = 0; i < NVI; ++i) {
    const int inst = wave * NVI + i;
    long e = a0 + ((long)inst * 64 + lane) * 2;
    if (e > nnz - 2) e = (nnz - 2) & ~1l;
    if (NT) lds_dma16_nt(a.val + e, (unsigned)inst * 1024u); else lds_dma16(a.val + e, (unsigned)inst * 1024u);
  }
#pragma unroll
  for (int i = 0; i < NCI; ++i) {
    const int inst = wave * NCI + i;
    if (inst * 1024 < EMAX * 4) {
      long e = a0 + ((long)inst * 64 + lane) * 4;
      if (e > nnz - 4) e = (nnz - 4) & ~3l;
      if (NT) lds_dma16_nt(a.col + e, (unsigned)(EMAX * 8) + (unsigned)inst * 1024u); else lds_dma16(a.col + e, (unsigned)(EMAX * 8) + (unsigned)inst * 1024u);
    }
  }
  int rs[RPT], re[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) { const int r = r0 + tid + q * T; const bool act = r < a.row_end; rs[q] = act ? a.rowptr[r] : 0; re[q] = act ? a.rowptr[r + 1] : 0; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  double acc[RPT];
  int c[RPT][8]; double v[RPT][8], xv[RPT][8];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    acc[q] = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int j = max(min(rs[q] + e, re[q] - 1), (int)a0) - (int)a0; c[q][e] = cs[j]; v[q][e] = vs[j]; }
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[q][e] = a.x[c[q][e]];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (rs[q] + e < re[q]) acc[q] += v[q][e] * xv[q][e];
    for (int j0 = rs[q] + 8; j0 < re[q]; j0 += 8)       // (rows longer than 8: the rest, batch by batch)
      for (int e = 0; e < 8 && j0 + e < re[q]; ++e) acc[q] += vs[j0 + e - a0] * a.x[cs[j0 + e - a0]];
    const int r = r0 + tid + q * T;
    if (r < a.row_end) a.y[r] = acc[q];
  }
}
template <int T, int RPT, int EMAX, bool NT>
void launch_dma1(const StreamArgs& a, hipStream_t st) {
  const int nrows = a.row_end - a.row_begin;
  const int ntiles = (nrows + T * RPT - 1) / (T * RPT);
  static long nnz = -1;
  if (nnz < 0) { int v; hipMemcpy(&v, a.rowptr + nrows, 4, hipMemcpyDeviceToHost); nnz = v; }
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)csr_dma1_kernel<T, RPT, EMAX, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL((csr_dma1_kernel<T, RPT, EMAX, NT>), dim3(ntiles), dim3(T), (size_t)EMAX * 12, st, a, nnz);
}
#define VA(T, R, E, NT) {"DMA1 T" #T " RPT" #R " EMAX" #E " NT" #NT, launch_dma1<T, R, E, NT>}

#define V(T, R, L, VEC, NT, XCD) {"T" #T " RPT" #R " LDS" #L " VEC" #VEC " NT" #NT " XCD" #XCD, launch_cfg<StreamCfg<T, (T) * (R), L, VEC, NT, XCD>>}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 256;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int rounds = argc > 3 ? atoi(argv[3]) : 3;
  const bool ship_only = argc > 4 && std::string(argv[4]) == "ship";  // PMC runs: shipped config only
  const long n = (long)N * N * N;
  std::vector<int> rowptr(n + 1), col;
  std::vector<double> val;
  col.reserve(7 * n); val.reserve(7 * n);
  long k = 0;
  for (int z = 0; z < N; ++z) for (int y = 0; y < N; ++y) for (int x = 0; x < N; ++x) {
    long r = x + (long)N * (y + (long)N * z);
    rowptr[r] = (int)k;
    if (z > 0) { col.push_back((int)(r - (long)N * N)); val.push_back(-1); ++k; }
    if (y > 0) { col.push_back((int)(r - N)); val.push_back(-1); ++k; }
    if (x > 0) { col.push_back((int)(r - 1)); val.push_back(-1); ++k; }
    col.push_back((int)r); val.push_back(6); ++k;
    if (x < N - 1) { col.push_back((int)(r + 1)); val.push_back(-1); ++k; }
    if (y < N - 1) { col.push_back((int)(r + N)); val.push_back(-1); ++k; }
    if (z < N - 1) { col.push_back((int)(r + (long)N * N)); val.push_back(-1); ++k; }
  }
  rowptr[n] = (int)k;
  const long nnz = k;
  std::vector<double> hx(n);
  unsigned long long s = 1;
  for (long i = 0; i < n; ++i) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; hx[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
  // perturb values so the matrix is not sign/zero-trivial
  for (long i = 0; i < nnz; ++i) val[i] *= 1.0 + 1e-3 * ((i * 2654435761u) % 97) / 97.0;

  int *d_rowptr, *d_col; double *d_val, *d_x, *d_y, *d_big;
  CK(hipMalloc(&d_rowptr, (n + 1) * 4)); CK(hipMalloc(&d_col, nnz * 4)); CK(hipMalloc(&d_val, nnz * 8));
  CK(hipMalloc(&d_x, n * 8)); CK(hipMalloc(&d_y, n * 8));
  CK(hipMemcpy(d_rowptr, rowptr.data(), (n + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val.data(), nnz * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_x, hx.data(), n * 8, hipMemcpyHostToDevice));
  const double alg_bytes = nnz * 12.0 + (n + 1) * 4.0 + 16.0 * n;
  printf("N=%d n=%ld nnz=%ld algorithmic bytes=%.0f\n", N, n, nnz, alg_bytes);

  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  // reference: host result on a sample of rows
  std::vector<double> ref(n);
  for (long r = 0; r < n; ++r) { double a = 0; for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) a += val[j] * hx[col[j]]; ref[r] = a; }

  // bandwidth references
  {
    size_t nb = (size_t)1 << 30;  // 1 GiB src + 1 GiB dst
    CK(hipMalloc(&d_big, 2 * nb));
    CK(hipMemset(d_big, 1, 2 * nb));
    for (int it = 0; it < 2; ++it) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy16_kernel, dim3(256 * 16), dim3(256), 0, st, (const double2*)d_big, (double2*)((char*)d_big + nb), nb / 16);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("copy16  1GiB->1GiB: %.3f ms  %.0f GB/s (read+write)\n", ms / 10, 2.0 * nb / (ms / 10 * 1e-3) / 1e9);
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(read16_kernel, dim3(256 * 16), dim3(256), 0, st, (const double2*)d_big, d_y, 2 * nb / 16);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("read16  2GiB:       %.3f ms  %.0f GB/s (read only)\n", ms / 10, 2.0 * nb / (ms / 10 * 1e-3) / 1e9);
    }
    CK(hipFree(d_big));
  }

  std::vector<Variant> vs = {
      V(256, 2, 4096, 1, false, true),   // round-1 first version
      V(256, 1, 2048, 2, true, true),
      V(256, 1, 2048, 2, true, false),
      V(256, 1, 2048, 2, false, false),
      V(256, 1, 2048, 4, true, false),
      V(256, 1, 2048, 4, false, false),
      V(256, 1, 2048, 1, false, false),
      V(256, 1, 1792, 2, true, false),
      V(256, 1, 1792, 4, false, false),
      V(256, 2, 4096, 4, true, false),
      V(256, 2, 4096, 4, false, false),
      V(256, 2, 3584, 4, false, false),
      V(512, 1, 4096, 4, false, false),
      V(512, 1, 4096, 2, true, false),
      V(512, 1, 3584, 4, false, false),
      V(1024, 1, 8192, 4, false, false),
      V(1024, 1, 7168, 2, true, false),
      V(128, 1, 1024, 2, true, false),
      V(128, 1, 1024, 4, false, false),
      V(128, 2, 2048, 4, false, false),
      VA(256, 1, 2048, false), VA(256, 1, 2048, true), VA(128, 1, 1024, false), VA(256, 2, 4096, false), VA(512, 1, 4096, false), VA(512, 1, 4096, true), VA(1024, 1, 8192, false), VA(128, 2, 2048, false),
      VD(256, 2048, false, 2), VD(256, 2048, false, 3), VD(256, 2048, true, 3), VD(256, 2048, false, 4), VD(256, 2048, true, 4),
      VD(512, 4096, false, 1), VD(512, 4096, false, 2), VD(512, 4096, true, 2),
  };
  if (ship_only) vs = {V(1024, 1, 8192, 4, false, false)};
  StreamArgs a{};
  a.rowptr = d_rowptr; a.col = d_col; a.val = d_val; a.x = d_x; a.y = d_y; a.row_begin = 0; a.row_end = (int)n;
  std::vector<std::vector<float>> times(vs.size() + 1);
  std::vector<double> hy(n);
  for (int rd = 0; rd < rounds; ++rd) {
    for (size_t v = 0; v <= vs.size(); ++v) {
      auto run = [&]() {
        if (v == vs.size()) hipLaunchKernelGGL(csr_scalar_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (int)n, d_rowptr, d_col, d_val, d_x, d_y);
        else vs[v].launch(a, st);
      };
      if (rd == 0) {
        CK(hipMemsetAsync(d_y, 0xff, n * 8, st));
        run();
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        CK(hipMemcpy(hy.data(), d_y, n * 8, hipMemcpyDeviceToHost));
        long bad = 0;
        for (long r = 0; r < n; ++r) if (hy[r] != ref[r]) ++bad;
        if (bad) printf("  !! variant %zu: %ld rows differ from the host result\n", v, bad);
      }
      for (int i = 0; i < 3; ++i) run();
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) run();
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      times[v].push_back(ms / reps);
    }
  }
  printf("%-44s %9s %9s %8s %7s\n", "variant", "min ms", "med ms", "GB/s", "%8TB/s");
  for (size_t v = 0; v <= vs.size(); ++v) {
    auto t = times[v];
    std::sort(t.begin(), t.end());
    const double mn = t[0], md = t[t.size() / 2];
    printf("%-44s %9.4f %9.4f %8.0f %7.1f\n", v == vs.size() ? "thread-per-row (no LDS)" : vs[v].name.c_str(), mn, md,
           alg_bytes / (md * 1e-3) / 1e9, 100.0 * alg_bytes / (md * 1e-3) / 8e12);
  }
  return 0;
}
