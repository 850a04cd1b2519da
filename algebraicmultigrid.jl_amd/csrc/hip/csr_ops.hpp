// csr_ops.hpp — launches on one CSR operator: SpMV / residual / add / Jacobi (csr_stream_kernel) and the
// Gauss-Seidel / SOR sweeps (slot, chain, block-inverse and stream kernels).  Included by amghip.hip.
#pragma once
#include <atomic>

namespace {


template <int MODE, class CFG = DefaultCfg, bool CODED = false>
int launch_stream(const StreamArgs& a0, hipStream_t st, int ncolv = 1) {
  const int nrows = a0.row_end - a0.row_begin;
  if (nrows <= 0) return AMGH_OK;
  StreamArgs a = a0;
  a.ncolv = ncolv;
  const int nb = (nrows + CFG::ROWS - 1) / CFG::ROWS;
  // multi-column launches: tiles padded to a multiple of 8, times ncolv (multi_column_block)
  const int64_t grid = (CFG::XCD || ncolv > 1) ? (int64_t)((nb + kNumXcd - 1) / kNumXcd) * kNumXcd * ncolv : nb;
  hipLaunchKernelGGL((csr_stream_kernel<MODE, CFG, CODED>), dim3((unsigned)grid), dim3(CFG::THREADS), 0, st, a);
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

// SpMV-type launches pick their shape by size: the default configuration gives a workgroup 1 024 rows — right for the
// big operators (it streams), but a coarse level of a few thousand LONG rows then runs on a handful of CUs
// (5 195 rows of 124 entries: 6 workgroups, 0.18 ms for 8 MB).  Below 2^18 rows: 64 rows per 256-thread workgroup.
// The per-row sums are the same in-order sums either way.
using SmallCfg = StreamCfg<256, 64, 4096, 2, false, false>;
// (the value-coded variant: a third of the bytes per row — measured shapes in profiles/r05_coded_cfg.log)
#ifndef AMGH_CODED_CFG
#define AMGH_CODED_CFG 1024, 1024, 8192, 4
#endif
using CodedCfg = StreamCfg<AMGH_CODED_CFG, false, false>;
// xcd: the operator asked for the XCD-contiguous mapping of its workgroups (amgh_csr::xcd_map, decided by a timing at amgh_finalize:
// the restriction of the second level of the 256^3 hierarchy — long rows gathering from a vector eight times its own — 0.141 -> 0.108 ms;
// every other big operator is flat or slower: profiles/r06_stream_xcd.log).  Same rows, same in-order sums.
template <int MODE>
int launch_stream_sized(const StreamArgs& a, hipStream_t st, int ncolv, bool xcd = false) {
  if (a.row_end - a.row_begin < (1 << 18)) return launch_stream<MODE, SmallCfg>(a, st, ncolv);
  if constexpr (MODE == M_SPMV) {
    if (xcd && g_stream_xcd && ncolv == 1) {
      if (a.ccol && (g_stream_code || !a.col)) return launch_stream<MODE, StreamCfg<1024, 1024, 8192, 4, false, true>, true>(a, st, ncolv);
      return launch_stream<MODE, StreamCfg<1024, 1024, 8192, 4, false, true>>(a, st, ncolv);
    }
  }
  // value-coded columns: 4 bytes per entry (the trimmed footprint keeps ONLY them where an operator has them: a.col == nullptr)
  if (a.ccol && (g_stream_code || !a.col)) return launch_stream<MODE, CodedCfg, true>(a, st, ncolv);
  return launch_stream<MODE>(a, st, ncolv);
}

// ---- value-coded columns (CodedCols) ----------------------------------------------------------------------------------
typedef std::conditional<sizeof(real) == 8, unsigned long long, unsigned int>::type rbits_t;
__device__ __forceinline__ rbits_t real_bits(real v) { rbits_t b; __builtin_memcpy(&b, &v, sizeof(real)); return b; }
constexpr int kCodeMiss = 4096;
// position of v in the sorted table (by bit pattern), or -1
__device__ __forceinline__ int code_find(const rbits_t* tab, int ntab, rbits_t v) {
  int lo = 0, hi = ntab;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (tab[mid] < v) lo = mid + 1; else hi = mid; }
  return (lo < ntab && tab[lo] == v) ? lo : -1;
}
// values of val[0, nnz) that are not in the table: counted, the first kCodeMiss of them recorded
__global__ void code_probe_kernel(const real* val, int64_t nnz, const real* tab, int ntab, unsigned* miss_cnt, real* miss) {
  __shared__ rbits_t s_tab[kCodeMax];
  if ((int)threadIdx.x < ntab) s_tab[threadIdx.x] = real_bits(tab[threadIdx.x]);
  __syncthreads();
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
    const real v = val[k];
    if (code_find(s_tab, ntab, real_bits(v)) >= 0) continue;
    // (a table that is far from complete: the list fills within the first few thousand entries and the rest only look)
    if (__hip_atomic_load(miss_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)kCodeMiss) continue;
    const unsigned i = atomicAdd(miss_cnt, 1u);
    if (i < (unsigned)kCodeMiss) miss[i] = v;
  }
}
__global__ void code_encode_kernel(const int32_t* col, const real* val, int64_t nnz, const real* tab, int ntab, uint32_t* ccol) {
  __shared__ rbits_t s_tab[kCodeMax];
  if ((int)threadIdx.x < ntab) s_tab[threadIdx.x] = real_bits(tab[threadIdx.x]);
  __syncthreads();
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x)
    ccol[k] = (uint32_t)col[k] | ((uint32_t)code_find(s_tab, ntab, real_bits(val[k])) << kCodeBits);
}
// out: the coded columns of (col, val), or left empty where the operator does not qualify (more than 256 distinct values —
// by bit pattern —, 2^24 columns or more, nothing to gain below 2^18 rows).  A few passes over val on the device.
int code_values(const int32_t* col, const real* val, int64_t nrows, int64_t ncols, int64_t nnz, CodedCols* out, hipStream_t st) {
  if (!g_stream_code || out->ccol || nrows < (1 << 18) || ncols > ((int64_t)1 << kCodeBits) || nnz <= 0 || !col || !val) return AMGH_OK;
  unsigned* d_cnt = nullptr; real *d_miss = nullptr, *d_tab = nullptr;
  RC_TRY(dev_alloc(&d_cnt, 1));
  int rc = dev_alloc(&d_miss, kCodeMiss);
  if (rc == AMGH_OK) rc = dev_alloc(&d_tab, kCodeMax);
  std::vector<rbits_t> tab;
  std::vector<real> miss((size_t)kCodeMiss);
  bool ok = rc == AMGH_OK;
  // (the first pass looks at a sample: with an empty table every entry is a miss)
  for (int pass = 0; ok && pass < 8; ++pass) {
    const int64_t span = pass == 0 ? std::min<int64_t>(nnz, 1 << 20) : nnz;
    const int ntab = (int)tab.size();
    if (hipMemsetAsync(d_cnt, 0, 4, st) != hipSuccess) { rc = -1001; break; }
    if (ntab > 0) {
      std::vector<real> tv((size_t)ntab);
      for (int i = 0; i < ntab; ++i) std::memcpy(&tv[(size_t)i], &tab[(size_t)i], sizeof(real));
      if (hipMemcpyAsync(d_tab, tv.data(), sizeof(real) * (size_t)ntab, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -1001; break; }
      if (hipStreamSynchronize(st) != hipSuccess) { rc = -1001; break; }   // (tv goes out of scope)
    }
    hipLaunchKernelGGL(code_probe_kernel, dim3((unsigned)std::min<int64_t>((span + 255) / 256, 8192)), dim3(256), 0, st, val, span, (const real*)d_tab, ntab, d_cnt, d_miss);
    unsigned cnt = 0;
    if (hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = -1001; break; }
    if (cnt == 0) { if (pass > 0) break; else continue; }
    const unsigned got = std::min<unsigned>(cnt, (unsigned)kCodeMiss);
    if (hipMemcpy(miss.data(), d_miss, sizeof(real) * got, hipMemcpyDeviceToHost) != hipSuccess) { rc = -1001; break; }
    for (unsigned i = 0; i < got; ++i) { rbits_t b; std::memcpy(&b, &miss[i], sizeof(real)); tab.push_back(b); }
    std::sort(tab.begin(), tab.end());
    tab.erase(std::unique(tab.begin(), tab.end()), tab.end());
    if ((int)tab.size() > kCodeMax) ok = false;
    if (pass == 7) ok = false;   // (still missing after eight passes: not an operator of few values)
  }
  if (rc == AMGH_OK && ok && !tab.empty()) {
    std::vector<real> tv(tab.size());
    for (size_t i = 0; i < tab.size(); ++i) std::memcpy(&tv[i], &tab[i], sizeof(real));
    rc = dev_alloc(&out->ccol, nnz + 4);   // (the kernel's 16-byte loads may start up to 3 entries before / end 3 behind a range)
    if (rc == AMGH_OK) rc = dev_alloc(&out->vtab, kCodeMax);
    if (rc == AMGH_OK && hipMemcpyAsync(d_tab, tv.data(), sizeof(real) * tv.size(), hipMemcpyHostToDevice, st) != hipSuccess) rc = -1001;
    if (rc == AMGH_OK && hipMemcpyAsync(out->vtab, tv.data(), sizeof(real) * tv.size(), hipMemcpyHostToDevice, st) != hipSuccess) rc = -1001;
    if (rc == AMGH_OK) {
      hipLaunchKernelGGL(code_encode_kernel, dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 8192)), dim3(256), 0, st, col, val, nnz, (const real*)d_tab, (int)tab.size(), out->ccol);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = -1001;
    }
    if (rc == AMGH_OK) { out->n = (int)tab.size(); out->bytes = nnz * 4 + kCodeMax * kRealB; }
    else out->free_dev();
  }
  hipFree(d_cnt); hipFree(d_miss); hipFree(d_tab);
  if (rc == AMGH_ENOMEM) { (void)hipGetLastError(); out->free_dev(); return AMGH_OK; }   // (out of memory: the plain columns stay)
  return rc;
}

// ncolv right-hand-side columns (x: ncols apart, y and b: nrows apart) in one launch
int csr_apply(const amgh_csr* op, int mode, const real* x, const real* b, real* y, hipStream_t st,
              int ncolv = 1) {
  StreamArgs a{};
  a.rowptr = op->rowptr; a.col = op->col; a.val = op->val;
  a.ccol = op->cc.ccol; a.vtab = op->cc.vtab; a.vtab_n = op->cc.n;
  a.x = x; a.y = y; a.b = b;
  a.row_begin = 0; a.row_end = (int32_t)op->nrows;
  a.ldx = op->ncols; a.ldy = op->nrows; a.ldb = op->nrows;
  switch (mode) {
    case M_SPMV: return launch_stream_sized<M_SPMV>(a, st, ncolv, op->xcd_map);
    case M_RESID: return launch_stream_sized<M_RESID>(a, st, ncolv);
    case M_ADD: return launch_stream_sized<M_ADD>(a, st, ncolv);
  }
  return AMGH_EINVAL;
}

int csr_jacobi(amgh_csr* op, real omega, const real* xin, const real* b, real* xout, hipStream_t st,
               int ncolv = 1) {
  RC_TRY(csr_ensure_diag(op, st));
  StreamArgs a{};
  a.rowptr = op->rowptr; a.col = op->col; a.val = op->val;
  a.x = xin; a.y = xout; a.b = b; a.dpos = op->dpos; a.diag = op->diag; a.omega = omega;
  a.row_begin = 0; a.row_end = (int32_t)op->nrows;
  a.ldx = op->ncols; a.ldy = op->nrows; a.ldb = op->nrows;
  return launch_stream_sized<M_JACOBI>(a, st, ncolv);
}

// the same sweep from x = 0: no matrix pass (jacobi_zero_kernel)
int csr_jacobi_zero(amgh_csr* op, real omega, const real* b, real* xout, hipStream_t st, int ncolv = 1) {
  RC_TRY(csr_ensure_diag(op, st));
  if (op->nrows > 0)
    hipLaunchKernelGGL(jacobi_zero_kernel, dim3((unsigned)std::min<int64_t>((op->nrows + 255) / 256, 1 << 20), ncolv), dim3(256), 0, st,
                       b, (const real*)op->diag, xout, (int64_t)op->nrows, omega, (int64_t)op->nrows, (int64_t)op->nrows);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
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
// the whole operator as one record, walked by a single wave (gs_wave_kernel); dir: 0 forward, 1 backward, 2 both
template <int MAXK, int DIR>
int launch_wave_kd(const WaveArgs& a, bool sor, size_t lds, hipStream_t st, int ncolv) {
  // more than the default 64 KB of dynamic LDS needs the attribute: once per kernel AND device (ranks of one process
  // sweep on different devices)
  static std::atomic<uint64_t> attr_set{0};
  int dev = 0;
  if (lds > 64 * 1024 && hipGetDevice(&dev) == hipSuccess && !((attr_set.load() >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute((const void*)gs_wave_kernel<false, DIR, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gs_wave_kernel<true, DIR, MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.fetch_or((uint64_t)1 << (dev & 63));
  }
  if (sor) hipLaunchKernelGGL((gs_wave_kernel<true, DIR, MAXK>), dim3(ncolv), dim3(kWaveThreads), lds, st, a);
  else hipLaunchKernelGGL((gs_wave_kernel<false, DIR, MAXK>), dim3(ncolv), dim3(kWaveThreads), lds, st, a);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
template <int MAXK>
int launch_wave_k(const WaveArgs& a, bool sor, int dir, size_t lds, hipStream_t st, int ncolv) {
  switch (dir) {
    case 0: return launch_wave_kd<MAXK, 0>(a, sor, lds, st, ncolv);
    case 1: return launch_wave_kd<MAXK, 1>(a, sor, lds, st, ncolv);
    default: return launch_wave_kd<MAXK, 2>(a, sor, lds, st, ncolv);
  }
}
int launch_wave(const WaveArgs& a, int maxk, bool sor, int dir, size_t lds, hipStream_t st, int ncolv) {
  switch (maxk) {
    case 6: return launch_wave_k<6>(a, sor, dir, lds, st, ncolv);
    case 12: return launch_wave_k<12>(a, sor, dir, lds, st, ncolv);
    case 18: return launch_wave_k<18>(a, sor, dir, lds, st, ncolv);
    case 24: return launch_wave_k<24>(a, sor, dir, lds, st, ncolv);
    case 30: return launch_wave_k<30>(a, sor, dir, lds, st, ncolv);
    case 36: return launch_wave_k<36>(a, sor, dir, lds, st, ncolv);
  }
  return AMGH_EINVAL;
}
template <int E, int DIR>
int launch_waveq_ed(const WaveArgs& a, bool sor, size_t lds, hipStream_t st, int ncolv) {
  static std::atomic<uint64_t> attr_set{0};
  int dev = 0;
  if (lds > 64 * 1024 && hipGetDevice(&dev) == hipSuccess && !((attr_set.load() >> (dev & 63)) & 1)) {
    (void)hipFuncSetAttribute((const void*)gs_waveq_kernel<false, DIR, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gs_waveq_kernel<true, DIR, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.fetch_or((uint64_t)1 << (dev & 63));
  }
  if (sor) hipLaunchKernelGGL((gs_waveq_kernel<true, DIR, E>), dim3(ncolv), dim3(kWaveThreads), lds, st, a);
  else hipLaunchKernelGGL((gs_waveq_kernel<false, DIR, E>), dim3(ncolv), dim3(kWaveThreads), lds, st, a);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
template <int E>
int launch_waveq_e(const WaveArgs& a, bool sor, int dir, size_t lds, hipStream_t st, int ncolv) {
  switch (dir) {
    case 0: return launch_waveq_ed<E, 0>(a, sor, lds, st, ncolv);
    case 1: return launch_waveq_ed<E, 1>(a, sor, lds, st, ncolv);
    default: return launch_waveq_ed<E, 2>(a, sor, lds, st, ncolv);
  }
}
// the same walk with four lanes per row (gs_waveq_kernel); e = entries per mini-row
int launch_waveq(const WaveArgs& a, int e, bool sor, int dir, size_t lds, hipStream_t st, int ncolv) {
  switch (e) {
    case 3: return launch_waveq_e<3>(a, sor, dir, lds, st, ncolv);
    case 5: return launch_waveq_e<5>(a, sor, dir, lds, st, ncolv);
    case 7: return launch_waveq_e<7>(a, sor, dir, lds, st, ncolv);
    case 9: return launch_waveq_e<9>(a, sor, dir, lds, st, ncolv);
  }
  return AMGH_EINVAL;
}
// does a sweep over this operator run as ONE gs_wave_kernel launch?  (then a symmetric sweep may be asked for in one
// call: csr_gs_sweep(..., sym_pair = true))
bool gs_wave_path(const GsSchedule* g, bool sor) {
  if (!g || !g->ww_rec || g->ncols != g->n || g_gs_tiny != 1 || g_chain_tim) return false;
  if (!sor && g_gs_block_inverse && ((g->dti_f && g_gs_dense_tri) || g->nblk > 0)) return false;   // those paths come first
  return g->segs.size() == 1 && g->segs[0].chain && g->segs[0].l0 == 0 && g->segs[0].l1 == g->nlev;
}
template <int T>
int launch_chain_tiny_t(const ChainArgs& c, bool sor, int n, int nnz, int nlev, hipStream_t st, int ncolv) {
  if (sor) hipLaunchKernelGGL((gs_chain_tiny_kernel<true, T>), dim3(ncolv), dim3(T), 0, st, c, n, nnz, nlev);
  else hipLaunchKernelGGL((gs_chain_tiny_kernel<false, T>), dim3(ncolv), dim3(T), 0, st, c, n, nnz, nlev);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
// the whole operator in LDS (gs_chain_tiny_kernel): workgroup size by the segment's widest level, as below
int launch_chain_tiny(const ChainArgs& c, bool sor, int threads, int n, int nnz, int nlev, hipStream_t st, int ncolv) {
  switch (threads) {
    case 64: return launch_chain_tiny_t<64>(c, sor, n, nnz, nlev, st, ncolv);
    case 256: return launch_chain_tiny_t<256>(c, sor, n, nnz, nlev, st, ncolv);
    default: return launch_chain_tiny_t<1024>(c, sor, n, nnz, nlev, st, ncolv);
  }
}
// threads = workgroup size class of the segment (64 / 256 / 1024), see gs_build
int launch_chain(const ChainArgs& c, bool sor, bool ldsx, int threads, int nx, hipStream_t st, int ncolv) {
  switch (threads) {
    case 64: return launch_chain_t<64, 4>(c, sor, ldsx, nx, st, ncolv);
    case 256: return launch_chain_t<256, 4>(c, sor, ldsx, nx, st, ncolv);
    default: return launch_chain_t<1024, 4>(c, sor, ldsx, nx, st, ncolv);  // PF = 8 spills at 1024 threads (128 VGPRs)
  }
}

// SpMV-type launch on raw CSR arrays with explicit column strides (operators and vectors in level order)
int raw_apply(int mode, const int32_t* rowptr, const int32_t* col, const real* val, int64_t nrows, const real* x,
              int64_t ldx, const real* b, int64_t ldb, real* y, int64_t ldy, hipStream_t st, int ncolv, const CodedCols* cc = nullptr) {
  StreamArgs a{};
  a.rowptr = rowptr; a.col = col; a.val = val;
  if (cc) { a.ccol = cc->ccol; a.vtab = cc->vtab; a.vtab_n = cc->n; }
  a.x = x; a.y = y; a.b = b;
  a.row_begin = 0; a.row_end = (int32_t)nrows;
  a.ldx = ldx; a.ldy = ldy; a.ldb = ldb;
  switch (mode) {
    case M_SPMV: return launch_stream_sized<M_SPMV>(a, st, ncolv);
    case M_RESID: return launch_stream_sized<M_RESID>(a, st, ncolv);
    case M_ADD: return launch_stream_sized<M_ADD>(a, st, ncolv);
  }
  return AMGH_EINVAL;
}

// y = M x / y += M x for a block of bs in {2, 4, 8, 16} right-hand sides through the interleaved copy of x (csr_il_kernel):
// x: nx rows, ldx apart, column-major; il: scratch of nx * bs reals.  Returns AMGH_EUNSUPPORTED for other block sizes.
template <int BS>
int il_apply_t(bool add, const int32_t* rowptr, const int32_t* col, const real* val, int64_t nrows, const real* x, int64_t nx,
               int64_t ldx, real* il, real* y, int64_t ldy, hipStream_t st, const CodedCols* cc) {
  if (nx > 0)
    hipLaunchKernelGGL((to_interleaved_kernel<BS>), dim3((unsigned)((nx + 63) / 64)), dim3(256), 0, st, x, ldx, il, nx);
  const int64_t threads = nrows * BS;
  if (threads > 0) {
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (cc && cc->ccol && (g_stream_code || !col)) {   // value-coded columns: the words in place of the columns, the table in place of the values
      if (add) hipLaunchKernelGGL((csr_il_kernel<true, BS, true>), dim3(grid), dim3(256), 0, st, rowptr, (const int32_t*)cc->ccol, (const real*)cc->vtab, nrows, (const real*)il, y, ldy, cc->n);
      else hipLaunchKernelGGL((csr_il_kernel<false, BS, true>), dim3(grid), dim3(256), 0, st, rowptr, (const int32_t*)cc->ccol, (const real*)cc->vtab, nrows, (const real*)il, y, ldy, cc->n);
    } else
    if (add) hipLaunchKernelGGL((csr_il_kernel<true, BS>), dim3(grid), dim3(256), 0, st, rowptr, col, val, nrows, (const real*)il, y, ldy);
    else hipLaunchKernelGGL((csr_il_kernel<false, BS>), dim3(grid), dim3(256), 0, st, rowptr, col, val, nrows, (const real*)il, y, ldy);
  }
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
int il_apply(int bs, bool add, const int32_t* rowptr, const int32_t* col, const real* val, int64_t nrows, const real* x,
             int64_t nx, int64_t ldx, real* il, real* y, int64_t ldy, hipStream_t st, const CodedCols* cc = nullptr) {
  switch (bs) {
    case 2: return il_apply_t<2>(add, rowptr, col, val, nrows, x, nx, ldx, il, y, ldy, st, cc);
    case 4: return il_apply_t<4>(add, rowptr, col, val, nrows, x, nx, ldx, il, y, ldy, st, cc);
    case 8: return il_apply_t<8>(add, rowptr, col, val, nrows, x, nx, ldx, il, y, ldy, st, cc);
    case 16: return il_apply_t<16>(add, rowptr, col, val, nrows, x, nx, ldx, il, y, ldy, st, cc);
  }
  return AMGH_EUNSUPPORTED;
}
inline bool il_block(int bs) { return g_rhs_il && (bs == 2 || bs == 4 || bs == 8 || bs == 16); }
// r = b - A x for a block of bs in {2, 4, 8, 16} right-hand sides, the matrix read once (csr_resid_cols_kernel)
template <int BS>
int resid_cols_t(const int32_t* rowptr, const int32_t* col, const real* val, int64_t nrows, const real* x, int64_t ldx,
                 const real* b, int64_t ldb, real* y, int64_t ldy, hipStream_t st, const CodedCols* cc) {
  const int64_t threads = nrows * BS;
  if (threads > 0 && cc && cc->ccol && (g_stream_code || !col))
    hipLaunchKernelGGL((csr_resid_cols_kernel<BS, true>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, rowptr, (const int32_t*)cc->ccol,
                       (const real*)cc->vtab, nrows, x, ldx, b, ldb, y, ldy, cc->n);
  else
  if (threads > 0)
    hipLaunchKernelGGL((csr_resid_cols_kernel<BS>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, rowptr, col, val, nrows,
                       x, ldx, b, ldb, y, ldy);
  HIP_TRY(hipGetLastError());
  return AMGH_OK;
}
int resid_cols(int bs, const int32_t* rowptr, const int32_t* col, const real* val, int64_t nrows, const real* x, int64_t ldx,
               const real* b, int64_t ldb, real* y, int64_t ldy, hipStream_t st, const CodedCols* cc = nullptr) {
  switch (bs) {
    case 2: return resid_cols_t<2>(rowptr, col, val, nrows, x, ldx, b, ldb, y, ldy, st, cc);
    case 4: return resid_cols_t<4>(rowptr, col, val, nrows, x, ldx, b, ldb, y, ldy, st, cc);
    case 8: return resid_cols_t<8>(rowptr, col, val, nrows, x, ldx, b, ldb, y, ldy, st, cc);
    case 16: return resid_cols_t<16>(rowptr, col, val, nrows, x, ldx, b, ldb, y, ldy, st, cc);
  }
  return AMGH_EUNSUPPORTED;
}

template <int NCV>
int launch_slot_t(const SlotArgs& sa, bool sor, int grid, hipStream_t st) {
  if (sor) hipLaunchKernelGGL((gs_slot_kernel<true, NCV>), dim3(grid), dim3(kSlot), 0, st, sa);
  else hipLaunchKernelGGL((gs_slot_kernel<false, NCV>), dim3(grid), dim3(kSlot), 0, st, sa);
  return AMGH_OK;
}
template <int LPR, int EPT>
int launch_slot_lpr_t(const SlotArgs& sa, bool sor, int grid, hipStream_t st) {
  if (sor) hipLaunchKernelGGL((gs_slot_lpr_kernel<true, LPR, EPT>), dim3(grid), dim3(kSlot / EPT), 0, st, sa);
  else hipLaunchKernelGGL((gs_slot_lpr_kernel<false, LPR, EPT>), dim3(grid), dim3(kSlot / EPT), 0, st, sa);
  return AMGH_OK;
}
template <int EPT>
int launch_slot_lpr_e(const SlotArgs& sa, bool sor, int lpr, int grid, hipStream_t st) {
  switch (lpr) {
    case 1: return launch_slot_lpr_t<1, EPT>(sa, sor, grid, st);
    case 2: return launch_slot_lpr_t<2, EPT>(sa, sor, grid, st);
    case 4: return launch_slot_lpr_t<4, EPT>(sa, sor, grid, st);
    case 8: return launch_slot_lpr_t<8, EPT>(sa, sor, grid, st);
    default: return launch_slot_lpr_t<16, EPT>(sa, sor, grid, st);
  }
}
int launch_slot_lpr(const SlotArgs& sa, bool sor, int lpr, int ept, int grid, hipStream_t st) {
  return ept == 2 ? launch_slot_lpr_e<2>(sa, sor, lpr, grid, st) : launch_slot_lpr_e<1>(sa, sor, lpr, grid, st);
}
template <int K, int NCV>
int launch_sell_kc(const SellArgs& a, bool sor, dim3 grid, hipStream_t st) {
  // (wide blocks: 2 entries in flight per lane instead of 4 — 4 x 8 gathered values would not fit the registers)
  constexpr int BATCH = NCV >= 8 ? 2 : 4;
  if (sor) hipLaunchKernelGGL((gs_sell_kernel<true, K, BATCH, NCV>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gs_sell_kernel<false, K, BATCH, NCV>), grid, dim3(256), 0, st, a);
  return AMGH_OK;
}
template <int K>
int launch_sell_k(const SellArgs& a, bool sor, int grid, int ncolv, hipStream_t st) {
  // columns per launch: the largest of 8 / 4 / 2 / 1 that divides the block size
  if (ncolv % 8 == 0) return launch_sell_kc<K, 8>(a, sor, dim3(grid, ncolv / 8), st);
  if (ncolv % 4 == 0) return launch_sell_kc<K, 4>(a, sor, dim3(grid, ncolv / 4), st);
  if (ncolv % 2 == 0) return launch_sell_kc<K, 2>(a, sor, dim3(grid, ncolv / 2), st);
  return launch_sell_kc<K, 1>(a, sor, dim3(grid, ncolv), st);
}
int launch_sell(const SellArgs& a, bool sor, int k, int grid, int ncolv, hipStream_t st) {
  switch (k) {
    case 8: return launch_sell_k<8>(a, sor, grid, ncolv, st);
    case 16: return launch_sell_k<16>(a, sor, grid, ncolv, st);
    case 32: return launch_sell_k<32>(a, sor, grid, ncolv, st);
    default: return launch_sell_k<64>(a, sor, grid, ncolv, st);
  }
}
// the interleaved kernels of a block of bs = 2 / 4 / 8 right-hand sides (gs_slot_il_kernel, gs_sell_il_kernel)
int launch_slot_il(const SlotIlArgs& a, bool sor, int bs, int grid, hipStream_t st) {
#define AMGH_SLOT_IL(BS_) \
  do { if (sor) hipLaunchKernelGGL((gs_slot_il_kernel<true, BS_>), dim3(grid), dim3(kSlot), 0, st, a); \
       else hipLaunchKernelGGL((gs_slot_il_kernel<false, BS_>), dim3(grid), dim3(kSlot), 0, st, a); } while (0)
  switch (bs) {
    case 8: AMGH_SLOT_IL(8); break;
    case 4: AMGH_SLOT_IL(4); break;
    case 2: AMGH_SLOT_IL(2); break;
    default: return AMGH_EUNSUPPORTED;
  }
#undef AMGH_SLOT_IL
  return AMGH_OK;
}
inline bool sell_il_shape(int k, int bs) { return (bs == 8 || bs == 4) && k >= 2 * kWave / bs && k >= bs; }   // (a row's K entry lanes are whole gather rounds of 2 * 64 / bs entries)
template <int K, int BS>
int launch_sell_il_kb(const SellIlArgs& a, bool sor, int grid, hipStream_t st) {
  if constexpr (BS >= 4 && K >= 2 * kWave / BS && K >= BS) {
    constexpr int BATCH = BS >= 8 ? 2 : 4;   // (storage iterations in flight per lane; at BS = 8: 1: 7.57, 2: 6.13, 4: 6.16 ms on the 228 538-row level of the 256^3 hierarchy)
    if (sor) hipLaunchKernelGGL((gs_sell_il_kernel<true, K, BS, BATCH>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gs_sell_il_kernel<false, K, BS, BATCH>), dim3(grid), dim3(256), 0, st, a);
    return AMGH_OK;
  } else {
    return AMGH_EUNSUPPORTED;
  }
}
template <int K>
int launch_sell_il_k(const SellIlArgs& a, bool sor, int bs, int grid, hipStream_t st) {
  switch (bs) {
    case 8: return launch_sell_il_kb<K, 8>(a, sor, grid, st);
    case 4: return launch_sell_il_kb<K, 4>(a, sor, grid, st);
    case 2: return launch_sell_il_kb<K, 2>(a, sor, grid, st);
  }
  return AMGH_EUNSUPPORTED;
}
int launch_sell_il(const SellIlArgs& a, bool sor, int k, int bs, int grid, hipStream_t st) {
  switch (k) {
    case 8: return launch_sell_il_k<8>(a, sor, bs, grid, st);
    case 16: return launch_sell_il_k<16>(a, sor, bs, grid, st);
    case 32: return launch_sell_il_k<32>(a, sor, bs, grid, st);
    default: return launch_sell_il_k<64>(a, sor, bs, grid, st);
  }
}
// positions [p0, p0 + np) of the column-major xp (columns ld apart) into the interleaved copy
int il_copy_rows(int bs, const real* xp, int64_t ld, real* xil, int64_t p0, int64_t np, hipStream_t st) {
  if (np <= 0) return AMGH_OK;
  const unsigned grid = (unsigned)((np + 63) / 64);
  switch (bs) {
    case 8: hipLaunchKernelGGL((to_interleaved_kernel<8>), dim3(grid), dim3(256), 0, st, xp + p0, ld, xil + p0 * 8, np); break;
    case 4: hipLaunchKernelGGL((to_interleaved_kernel<4>), dim3(grid), dim3(256), 0, st, xp + p0, ld, xil + p0 * 4, np); break;
    case 2: hipLaunchKernelGGL((to_interleaved_kernel<2>), dim3(grid), dim3(256), 0, st, xp + p0, ld, xil + p0 * 2, np); break;
    default: return AMGH_EUNSUPPORTED;
  }
  HIP_TRY(hipGetLastError());
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
// xzero: the caller guarantees x == 0 on entry (every coarse-level pre-smoother of a cycle, and the fine one of
// ldiv!): the gather of x becomes a memset and the first merged pre-pass b - T x is just b.
// reuse_b: b has not changed since the previous smooth! call on this operator gathered it (post-smoother after the
// pre-smoother of the same cycle and level): its level-ordered copy is still in place.
// x_resident: the level-ordered x of this operator (g->xp) is already current: skip the gather of x (the cycle
// kept x there between the pre- and the post-smoother); no_scatter: leave the result in g->xp only.
// the schedule's per-column scratch (level-ordered b, x [; s], block-path s) for blocks of ncolv right-hand sides
int gs_ensure_cols(amgh_csr* op, int ncolv, hipStream_t st) {
  GsSchedule* g = op->gs;
  if (!g || ncolv <= g->cols_alloc) return AMGH_OK;
  HIP_TRY(hipStreamSynchronize(st));
  ++g_sched_epoch;
  hipFree(g->bp); hipFree(g->xp); g->bp = g->xp = nullptr;
  RC_TRY(dev_alloc(&g->bp, g->n * ncolv));
  RC_TRY(dev_alloc(&g->xp, g->xstride * ncolv));
  int64_t grown = 8 * (g->n + g->xstride) * (ncolv - g->cols_alloc);
  if (g->blk_s) {
    hipFree(g->blk_s); g->blk_s = nullptr;
    RC_TRY(dev_alloc(&g->blk_s, g->n * ncolv));
    grown += 8 * g->n * (ncolv - g->cols_alloc);
  }
  if (g->bw.flow.on && ncolv > g->bw.flow.mcols) {   // every column has its own mailboxes (epoch 0 is never a sweep's)
    // (the new buffer is allocated and zeroed BEFORE the old one goes: a failure here leaves the old mailboxes and their
    // column count in place, so that sweeps of fewer columns keep working)
    const size_t mb = (size_t)g->bw.flow.mail_stride * (size_t)ncolv;
    void* fresh = nullptr;
    if (hipMalloc(&fresh, mb) != hipSuccess) { (void)hipGetLastError(); return AMGH_ENOMEM; }
    if (hipMemset(fresh, 0, mb) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(fresh); return -1001; }
    (void)hipFree(g->bw.flow.mbox);
    g->bw.flow.mbox = fresh;
    const int64_t more = g->bw.flow.mail_stride * (int64_t)(ncolv - g->bw.flow.mcols);
    g->bw.flow.bytes += more; g->bw.rec_bytes += more; g->slot_bytes += more;
    grown += more;
    g->bw.flow.mcols = ncolv;
  }
  g->bytes += grown;
  op->bytes += grown;
  g->cols_alloc = ncolv;
  g->bp_cols = 0;
  return AMGH_OK;
}

int csr_gs_sweep(amgh_csr* op, bool backward, bool sor, real omega, real* x, const real* b, hipStream_t st,
                 bool first = true, bool last = true, int ncolv = 1, bool xzero = false, bool reuse_b = false,
                 bool x_resident = false, bool no_scatter = false, bool sym_pair = false) {
  // sym_pair: this call is the forward half of a symmetric sweep and the caller leaves the backward half to it — only
  // where gs_wave_path() says the sweep is one single-wave launch
  RC_TRY(csr_ensure_gs(op));
  GsSchedule* g = op->gs;
  if (g->n <= 0) return AMGH_OK;
  if (sym_pair && (backward || !gs_wave_path(g, sor))) return AMGH_EINVAL;
  RC_TRY(gs_ensure_cols(op, ncolv, st));
  if (g->dti_f && g_gs_dense_tri && g_gs_block_inverse && !sor) {
    // small operator with the triangles of its (large) diagonal blocks inverted densely: block after block,
    // s = b - (everything outside the block's triangle) x on the block's rows, then x_blk = T_blk^-1 s
    g->bp_cols = 0;  // (no level-ordered copy of b is made on this path)
    const GsSchedule::Tri& t = backward ? g->dtri_b : g->dtri_f;
    const int B = g->dti_B, nb = (int)g->dti_off.size() - 1;
    for (int q = 0; q < nb; ++q) {
      const int k = backward ? nb - 1 - q : q;
      const int r0 = k * B, rb = (int)std::min<int64_t>(B, g->n - r0);
      StreamArgs ra{};
      ra.rowptr = t.rowptr; ra.col = t.col; ra.val = t.val;
      ra.x = x; ra.b = b; ra.y = g->blk_s;
      ra.row_begin = r0; ra.row_end = r0 + rb;
      ra.ldx = g->n; ra.ldy = g->n; ra.ldb = g->n;
      // (a block's rows are few and long: few per workgroup, or the pre-pass of a 4 096-row block runs on 64 CUs)
      // (8 rows per workgroup since round 6: 512 workgroups for a 4 096-row block — 14.135 -> 14.10 ms per 256^3 V-cycle against 16 rows,
      // bitwise; 32 rows 14.25, 128 threads with 8 / 4 rows 14.15 / 14.11: profiles/r06_dense_tri_rows.log; tunable gs_dti_pre = 16: the old shape)
      if (nb > 1 && g_gs_dti_pre != 16) RC_TRY((launch_stream<M_RESID, StreamCfg<256, 8, 4096, 2, false, false>>(ra, st, ncolv)));
      else if (nb > 1) RC_TRY((launch_stream<M_RESID, StreamCfg<256, 16, 4096, 2, false, false>>(ra, st, ncolv)));
      else RC_TRY((launch_stream<M_RESID, StreamCfg<256, 64, 4096, 2, false, false>>(ra, st, ncolv)));
      const real* Xk = (const real*)((backward ? g->dti_b : g->dti_f) + g->dti_off[k]);
      // columns per workgroup: the largest of 8 / 4 / 2 / 1 that divides the block of right-hand sides
      // (blocks of right-hand sides: 4 rows per workgroup share the loads of s — tri_gemm_kernel, bitwise tri_gemv_kernel; tunable gs_tri_rb)
      constexpr int RB = 4;
      const unsigned rgrid = (unsigned)((rb + RB - 1) / RB);
      if (ncolv % 8 == 0 && g_gs_tri_rb) hipLaunchKernelGGL((tri_gemm_kernel<8, RB>), dim3(rgrid, ncolv / 8), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
      else if (ncolv % 4 == 0 && g_gs_tri_rb) hipLaunchKernelGGL((tri_gemm_kernel<4, RB>), dim3(rgrid, ncolv / 4), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
      else if (ncolv % 2 == 0 && g_gs_tri_rb) hipLaunchKernelGGL((tri_gemm_kernel<2, RB>), dim3(rgrid, ncolv / 2), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
      else if (ncolv % 8 == 0) hipLaunchKernelGGL(tri_gemv_kernel<8>, dim3((unsigned)rb, ncolv / 8), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
      else if (ncolv % 4 == 0) hipLaunchKernelGGL(tri_gemv_kernel<4>, dim3((unsigned)rb, ncolv / 4), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
      else if (ncolv % 2 == 0) hipLaunchKernelGGL(tri_gemv_kernel<2>, dim3((unsigned)rb, ncolv / 2), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
      // (single columns: several rows per workgroup share the loads of s too — tunable gs_tri_rb1 = 2 / 4 / 8 rows, bitwise the one-row kernel)
      else if (g_gs_tri_rb1 == 2) hipLaunchKernelGGL((tri_gemm_kernel<1, 2>), dim3((unsigned)((rb + 1) / 2), ncolv), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
      else if (g_gs_tri_rb1 == 4) hipLaunchKernelGGL((tri_gemm_kernel<1, 4>), dim3((unsigned)((rb + 3) / 4), ncolv), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
      else if (g_gs_tri_rb1 == 8) hipLaunchKernelGGL((tri_gemm_kernel<1, 8>), dim3((unsigned)((rb + 7) / 8), ncolv), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
      else hipLaunchKernelGGL(tri_gemv_kernel<1>, dim3((unsigned)rb, ncolv), dim3(kThreads), 0, st, Xk, (const real*)(g->blk_s + r0), x + r0, rb, backward ? 1 : 0, (int64_t)g->n, (int64_t)g->n);
    }
    HIP_TRY(hipGetLastError());
    return AMGH_OK;
  }
  if (g->nblk > 0 && g_gs_block_inverse && !sor) {
    // small densely coupled operator: n/128 sequential block steps in natural row order
    g->bp_cols = 0;  // (no level-ordered copy of b is made on this path)
    BlockArgs ba{};
    const GsSchedule::Outer& o = backward ? g->blk_b : g->blk_f;
    ba.rowptr = o.rowptr; ba.col = o.col; ba.val = o.val; ba.tinv = o.tinv; ba.diag = g->blk_diag;
    const bool single = g->nblk <= kBlkSingle;  // few blocks: everything in one launch (see blockgs_build_dir)
    if (!single) {  // s = b - O_next x: every entry read here keeps its old value during this sweep
      StreamArgs ra{};
      ra.rowptr = o.nx_rowptr; ra.col = o.nx_col; ra.val = o.nx_val;
      ra.x = x; ra.b = b; ra.y = g->blk_s;
      ra.row_begin = 0; ra.row_end = (int32_t)g->n;
      ra.ldx = g->n; ra.ldy = g->n; ra.ldb = g->n;
      // few, long rows: 64 rows per workgroup so that the launch still covers the chip
      RC_TRY((launch_stream<M_RESID, StreamCfg<256, 64, 4096, 2, false, false>>(ra, st, ncolv)));
    }
    ba.x = x; ba.b = single ? b : g->blk_s; ba.n = (int32_t)g->n; ba.backward = backward ? 1 : 0;
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
      if (g_gs_block_pipe && !single) hipLaunchKernelGGL(gs_block_pipe_kernel, dim3(ncolv), dim3(kPipeThreads), 0, st, ba);
      else hipLaunchKernelGGL(gs_block_kernel, dim3(ncolv), dim3(kBlkThreads), 0, st, ba);
    }
    HIP_TRY(hipGetLastError());
    return AMGH_OK;
  }
  // which triangular system runs: the level-by-level one, merged Gauss-Seidel groups, or merged SOR groups (built on
  // the first sweep with this relaxation factor); merged SOR runs the GS kernels on the scaled system
  GsSchedule* lay = g;
  GsSchedule::Tri* tri = nullptr;
  real s_key = 0.0, flip_scale = 1.0;
  bool both_dirs = false;
  if (g_gs_merge > 1 && !sor && (backward ? g->mb : g->mf)) {
    lay = backward ? g->mb : g->mf;
    tri = backward ? &g->tri_b : &g->tri_f;
    s_key = 1.0;
    both_dirs = g->mf && g->mb;
  } else if (g_gs_merge > 1 && sor) {
    GsSchedule::SorSet* ss = sor_children(g, omega);
    // A merged system needs x and s in ONE vector (xstride = ncols + n).  The re-layout frees g->xp, so it may only
    // happen while no x lives there: at the first sweep of a smooth! call whose x is still in the caller's vector
    // (amgh_push_level already did it for the hierarchy's own smoothers).  Otherwise this CALL keeps the unmerged
    // layout (lay = g) — same iterate, just slower — and the next smooth! call re-lays out.
    if (ss && (ss->f || ss->b) && g->xstride == g->ncols && first && !x_resident) {
      HIP_TRY(hipStreamSynchronize(st));
      RC_TRY(gs_grow_xp_for_merged(g, &op->bytes));
    }
    if (ss && (backward ? ss->b : ss->f) && g->xstride > g->ncols) {
      lay = backward ? ss->b : ss->f;
      tri = backward ? &ss->tb : &ss->tf;
      s_key = omega;
      flip_scale = 2.0 - omega;
      both_dirs = ss->f && ss->b;
      sor = false;  // the scaled triangular system is swept like Gauss-Seidel
    }
  }
  const int64_t xs = g->xstride;  // doubles per column of xp
  // blocks of 2 / 4 / 8 right-hand sides on merged groups: the groups gather from an interleaved copy of [x ; s] (one
  // sector per entry for all columns instead of one line per column) and write both; xil's x part is current only from one
  // sweep of a smooth! call to the next (prolongation, fills and gathers write xp alone)
  const bool il = g_gs_il && g_gs_lpr != 1 && lay != g && !g->bw.on && (ncolv == 8 || ncolv == 4 || ncolv == 2);
  if (first || !il) g->xil_cols = 0;
  bool il_s_ok = false;   // the s part of xil was written by the kernel that produced s
  if (il && g->xil_cap < xs * ncolv) {   // (the first cycle of every kind runs eagerly: never during a capture)
    HIP_TRY(hipStreamSynchronize(st));
    ++g_sched_epoch;
    hipFree(g->xil); g->xil = nullptr; g->xil_cap = 0;
    RC_TRY(dev_alloc(&g->xil, xs * ncolv));
    g->xil_cap = xs * ncolv;
    g->bytes += (int64_t)sizeof(real) * xs * ncolv; op->bytes += (int64_t)sizeof(real) * xs * ncolv;
    g->xil_cols = 0;
  }
  if (first) {
    if (!(reuse_b && g->bp_cols == ncolv))
      hipLaunchKernelGGL(gather_perm_kernel, dim3(grid_for(g->n), ncolv), dim3(256), 0, st, b, g->perm, g->bp, (int)g->n,
                         (int64_t)g->n, (int64_t)g->n);
    g->bp_cols = ncolv;
    if (x_resident) {
      // nothing: g->xp holds x
    } else if (xzero) {
      // (a dataflow sweep of a square operator from x = 0 needs no x in memory at all: every block starts from zeros in LDS
      // and writes all its rows — see flow_xzero below)
      const bool flow_zero = g_gs_flow_xzero && g->bw.on && g->bw.flow.on && (g_gs_bw_flow || !g->bw.rec) && g->ncols == g->n && lay == g;
      if (!flow_zero) {
        const int64_t cnt = xs * ncolv;  // (a fill kernel: hipMemsetAsync of a small buffer costs far more than a launch)
        hipLaunchKernelGGL(fill_kernel, dim3(grid_for(cnt)), dim3(256), 0, st, g->xp, cnt, 0.0);
        if (il) {   // x = 0 in the interleaved copy too (no transposition pass)
          const int64_t ci = g->ncols * ncolv;
          hipLaunchKernelGGL(fill_kernel, dim3(grid_for(ci)), dim3(256), 0, st, g->xil, ci, 0.0);
          g->xil_cols = ncolv;
        }
      }
    } else {
      hipLaunchKernelGGL(gather_perm_kernel, dim3(grid_for(g->ncols), ncolv), dim3(256), 0, st, (const real*)x, g->permx,
                         g->xp, (int)g->ncols, (int64_t)g->ncols, xs);
    }
    HIP_TRY(hipGetLastError());
  }
  real* xp = g->xp;
  // merged-level sweep: the child schedule of this direction runs on ext = [x ; s], s = b - T x
  const real* rhs = g->bp;
  int64_t ldb = g->n;
  if (first) g->s_dir = -1;
  if (lay != g) {
    const bool flip = g->ncols == g->n && g->diag_nonzero && both_dirs && g_gs_flip &&
                      g->s_dir == (backward ? 0 : 1) && g->s_key == s_key;
    if (flip) {
      // the previous sweep of this smooth! call ran the other way on the same xp: s follows without a matrix pass
      if (il) {   // (the same expression, all columns of a row by one thread: s goes to xp and, as whole sectors, to xil)
        real* sil = g->xil + g->ncols * ncolv;
#define AMGH_FLIP_IL(BS_) hipLaunchKernelGGL((gs_flip_rhs_il_kernel<BS_>), dim3(grid_for(g->n)), dim3(256), 0, st, (const real*)g->bp, \
                         (const real*)lay->diag, (const real*)xp, xp + g->ncols, (int)g->n, (int64_t)g->n, xs, flip_scale, sil)
        if (ncolv == 8) AMGH_FLIP_IL(8); else if (ncolv == 4) AMGH_FLIP_IL(4); else AMGH_FLIP_IL(2);
#undef AMGH_FLIP_IL
        il_s_ok = true;
      } else
      hipLaunchKernelGGL(gs_flip_rhs_kernel, dim3(grid_for(g->n), ncolv), dim3(256), 0, st, (const real*)g->bp,
                         (const real*)lay->diag, (const real*)xp, xp + g->ncols, (int)g->n, (int64_t)g->n, xs,
                         flip_scale);
      HIP_TRY(hipGetLastError());
    } else if (first && xzero) {  // s = b - T * 0
      if (il) {
        real* sil = g->xil + g->ncols * ncolv;
#define AMGH_COPY_IL(BS_) hipLaunchKernelGGL((copy_cols_il_kernel<BS_>), dim3(grid_for(g->n)), dim3(256), 0, st, xp + g->ncols, \
                         (const real*)g->bp, (int)g->n, xs, (int64_t)g->n, sil)
        if (ncolv == 8) AMGH_COPY_IL(8); else if (ncolv == 4) AMGH_COPY_IL(4); else AMGH_COPY_IL(2);
#undef AMGH_COPY_IL
        il_s_ok = true;
      } else
      hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(g->n), ncolv), dim3(256), 0, st, xp + g->ncols,
                         (const real*)g->bp, (int)g->n, xs, (int64_t)g->n);
      HIP_TRY(hipGetLastError());
    } else {
      if (!tri->rowptr && g->n > 0 && tri == (backward ? &g->tri_b : &g->tri_f)) {
        // trimmed footprint: the backward pre-pass triangle is only needed by sweeps that START backward (alternating
        // sweeps derive s from the forward one) — built when a sweep first asks for it (never during graph capture:
        // the first cycle of every kind runs eagerly)
        HIP_TRY(hipStreamSynchronize(st));
        const int64_t before = g->bytes;
        RC_TRY(tri_build_dev(tri, g, backward, &g->bytes, 0.0, backward ? &g->tri_nnz_b : &g->tri_nnz));
        op->bytes += g->bytes - before;
      }
      StreamArgs ra{};
      ra.rowptr = tri->rowptr; ra.col = tri->col; ra.val = tri->val;
      ra.x = xp; ra.b = g->bp; ra.y = xp + g->ncols;
      ra.row_begin = 0; ra.row_end = (int32_t)g->n;
      ra.ldx = xs; ra.ldy = xs; ra.ldb = g->n;
      RC_TRY(launch_stream<M_RESID>(ra, st, ncolv));
    }
    g->s_dir = backward ? 1 : 0;
    g->s_key = s_key;
    rhs = xp + g->ncols;
    ldb = xs;
  } else {
    g->s_dir = -1;
  }
  // a sweep marked as one of a row-sharded level pipelined across the ranks (amghip_dist.hpp) exists only as the relayed dataflow
  // kernel on the extended lists: on any other path (a tunable switched the dataflow off, no block layout) the halo would stay
  // frozen — a hybrid sweep executed silently under the exact mode.  Refuse.
  if (op->pipe_epoch && !(g->bw.on && g->bw.flow.on && (g_gs_bw_flow || !g->bw.rec))) return AMGH_ESTATE;
  if (g->bw.on) {   // wavefront of blocks
    hipError_t e = hipSuccess;
    const bool flow = g->bw.flow.on && (g_gs_bw_flow || !g->bw.rec);
    if (flow) {   // as a dataflow: rows published as they are computed, blocks start on finished faces; a block of right-hand
                  // sides in launches of up to 8 columns (a walker / fetcher pair of waves per column in one workgroup)
      bw::FlowArgs<real> fa{g->bw.blocks, g->bw.flow.fd, g->bw.flow.srec, g->bw.flow.aux, g->bw.ext_col, g->bw.flow.fl_mb, g->bw.flow.fl_slot,
                            g->bw.flow.mbox, rhs, xp, omega, g->bw.head + 1, g->bw.nblocks, (int32_t)g->bw.flow.nmail,
                            g->bw.err, nullptr, (unsigned)g_gs_bw_spin, (int32_t)g_gs_bw_skip_pub};
      fa.ldb = ldb; fa.ldx = xs; fa.mail_stride = g->bw.flow.mail_stride;
      fa.xzero = (g_gs_flow_xzero && first && xzero && !x_resident && g->ncols == g->n) ? 1 : 0;   // the sweep that starts a smooth! call on x = 0
      if (ncolv > g->bw.flow.mcols) return AMGH_ESTATE;   // (gs_ensure_cols sized the mailboxes)
      // (column records + the blocks' dictionaries of value rows; the only layout a trimmed schedule holds where it has one)
      const bool dict = g->bw.flow.dict_on && (g_gs_bw_dict || !g->bw.flow.srec);
      // a single column: the block's walk relayed between walker waves (gs_relay.hpp) — the same layout, the same bits
      if (op->pipe_epoch) {   // one sweep of a row-sharded level pipelined across the ranks: the relayed kernel on the extended lists
        if (ncolv != 1 || !g->bw.flow.xon) return AMGH_ESTATE;
        fa.aux = g->bw.flow.xaux; fa.fl_mb = g->bw.flow.xfl_mb; fa.fl_slot = g->bw.flow.xfl_slot; fa.xlist = g->bw.flow.xlist;
        fa.epoch = op->pipe_epoch; fa.rmbox = op->pipe_rmbox; fa.grid = op->pipe_grid;
        if (dict) { fa.crec = g->bw.flow.crec; fa.dict = g->bw.flow.dict; fa.dict_ent = g->bw.flow.dict_ent; }
        e = bw::sweep_relay<real>(fa, g->bw.maxk, dict ? g->bw.flow.dict_lds : g->bw.flow.lds_max, sor, backward, st, BW_RELAY_W);
      } else if (ncolv == 1 && g_gs_bw_relay > 0) {
        // (the persistent grid pays on the plain 19-point records only: 0.98 -> 0.87 ms there, 0.776 -> 0.787 on the dictionary layout)
        fa.grid = g->bw.maxk > 6 ? (dict ? 0 : g_gs_bw_grid_long) : g_gs_bw_grid;
        fa.late = (g->bw.flow.late_ok && !g_gs_bw_inorder) ? 1 : 0;   // the dependency-aware row sum (gs_relay.hpp, LATE)
        if (dict) { fa.crec = g->bw.flow.crec; fa.dict = g->bw.flow.dict; fa.dict_ent = g->bw.flow.dict_ent; }
        e = bw::sweep_relay<real>(fa, g->bw.maxk, dict ? g->bw.flow.dict_lds : g->bw.flow.lds_max, sor, backward, st, BW_RELAY_W);
      } else {
        if (dict) {
          fa.crec = g->bw.flow.crec; fa.dict = g->bw.flow.dict; fa.dict_ent = g->bw.flow.dict_ent;
          fa.dict_lds = (int32_t)(g->bw.flow.dict_lds - g->bw.flow.lds_max);
        }
        e = bw::sweep_flow<real>(fa, g->bw.maxk, g->bw.flow.lds_max, sor, backward, st, ncolv, g_gs_bw_nc >= 0 ? g_gs_bw_nc : (fa.crec ? 3 : 2));
      }
    } else {
      bw::Args<real> ba{g->bw.blocks, g->bw.rec, g->bw.ext_col, rhs, xp, ldb, xs, omega, 0, nullptr};
      if (g_gs_bw_chain && ncolv == 1 && g->bw.flags) {   // one launch, blocks chained by flags
        bw::ChainArgs<real> ca{ba, backward ? g->bw.sdep_ptr : g->bw.dep_ptr, backward ? g->bw.sdep : g->bw.dep, g->bw.flags, g->bw.head,
                               g->bw.nblocks, g->bw.err};
        e = bw::sweep_chain<real>(ca, g->bw.maxk, g->bw.lds_max, sor, backward, st);
      } else {    // one launch per depth of the quotient DAG, one wave walking each block
        e = bw::sweep<real>(ba, g->bw.maxk, g->bw.launch_ptr, g->bw.lds_max, sor, backward, ncolv, st);
      }
    }
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  if (il) {   // what of xil the producers above did not write
    if (g->xil_cols != ncolv) RC_TRY(il_copy_rows(ncolv, xp, xs, g->xil, 0, il_s_ok ? g->ncols : xs, st));   // x (and s)
    else if (!il_s_ok) RC_TRY(il_copy_rows(ncolv, xp, xs, g->xil, g->ncols, g->n, st));                      // s alone
    g->xil_cols = ncolv;
  }
  const int ns = (int)lay->segs.size();
  // (g_gs_dup_launch: measurement hook — every group launched 1 + that many times: a group's launch is idempotent, the
  // repeats find its arrays in cache: what a prefetch of the next group's arrays could win at most)
  for (int kk = 0; kk < ns * (1 + g_gs_dup_launch); ++kk) {
    const int k = kk / (1 + g_gs_dup_launch);
    const GsSchedule::Seg& s = lay->segs[backward ? ns - 1 - k : k];
    if (s.chain) {
      ChainArgs c{};
      c.col = lay->col; c.val = lay->val; c.x = xp; c.bp = rhs; c.diag = lay->diag;
      c.rowmeta = lay->rowmeta; c.desc = lay->desc; c.omega = omega; c.tim = g_chain_tim;
      if (!backward) { c.lvl_begin = s.l0; c.lvl_end = s.l1; c.step = 1; }
      else { c.lvl_begin = s.l1 - 1; c.lvl_end = s.l0 - 1; c.step = -1; }
      c.ldx = xs; c.ldb = ldb;
      const int64_t nx = lay == g ? g->ncols : xs;  // entries of x (and s) a chained row may read
      const bool ldsx = nx <= kChainLdsX;            // they fit LDS
      if (lay == g && g->ww_rec && g->ncols == g->n && g_gs_tiny == 1 && !g_chain_tim && s.l0 == 0 && s.l1 == g->nlev) {
        WaveArgs wa{};   // one record, one wave, no barrier between the levels
        wa.rec = g->ww_rec; wa.bp = rhs; wa.x = xp; wa.ldb = ldb; wa.ldx = xs; wa.omega = omega;
        wa.n = (int32_t)g->n; wa.steps = g->ww_steps;
        if (g->wq_rec && g_gs_wave_quad) {   // four lanes per row
          wa.rec = g->wq_rec; wa.steps = g->wq_steps;
          RC_TRY(launch_waveq(wa, g->wq_E, sor, sym_pair ? 2 : backward ? 1 : 0, g->wq_lds, st, ncolv));
          continue;
        }
        RC_TRY(launch_wave(wa, g->ww_maxk, sor, sym_pair ? 2 : backward ? 1 : 0, g->ww_lds, st, ncolv));
        continue;
      }
      if (lay == g && g->tiny_ok && g->ncols == g->n && g_gs_tiny && !g_chain_tim) {   // the whole operator fits LDS: no global access per level
        RC_TRY(launch_chain_tiny(c, sor, s.rows, (int)g->n, (int)g->nnz, g->nlev, st, ncolv));
        continue;
      }
      RC_TRY(launch_chain(c, sor, ldsx, s.rows, (int)nx, st, ncolv));
      if (il) RC_TRY(il_copy_rows(ncolv, xp, xs, g->xil, lay->lvl_ptr[s.l0], lay->lvl_ptr[s.l1] - lay->lvl_ptr[s.l0], st));
    } else if (s.sell_k > 0 && lay->scol && g_gs_sell) {
      SellArgs la{};
      la.scol = lay->scol; la.sval = lay->sval; la.chunk = lay->schunk; la.diag = lay->diag; la.bp = rhs; la.x = xp;
      la.omega = omega; la.row0 = lay->lvl_ptr[s.l0]; la.nrows = lay->lvl_ptr[s.l0 + 1] - lay->lvl_ptr[s.l0];
      la.chunk0 = s.sell_chunk0; la.nchunks = s.sell_nchunks;
      la.ldx = xs; la.ldb = ldb;
      const int nwg = (s.sell_nchunks + 3) / 4;
      // XCD-contiguous workgroup mapping only where a launch has several workgroups per CU: below that there is no re-use
      // to win and the mapping unbalances the XCDs (the 228 538-row level of the 256^3 hierarchy: 1.40 -> 1.29 ms per pass)
      la.xcd_map = (g_gs_xcd_map && nwg >= 768) ? 1 : 0;
      const int grid = la.xcd_map ? ((nwg + kNumXcd - 1) / kNumXcd) * kNumXcd : nwg;
      if (il && sell_il_shape(s.sell_k, ncolv)) {
        SellIlArgs ia{};
        ia.scol = la.scol; ia.sval = la.sval; ia.chunk = la.chunk; ia.diag = la.diag; ia.xil = g->xil; ia.x = xp; ia.omega = omega;
        ia.row0 = la.row0; ia.nrows = la.nrows; ia.chunk0 = la.chunk0; ia.nchunks = la.nchunks; ia.xcd_map = la.xcd_map;
        ia.soff = (int32_t)g->ncols; ia.ldx = xs;
        RC_TRY(launch_sell_il(ia, sor, s.sell_k, ncolv, grid, st));
        HIP_TRY(hipGetLastError());
        continue;
      }
      RC_TRY(launch_sell(la, sor, s.sell_k, grid, ncolv, st));
      HIP_TRY(hipGetLastError());
      if (il) RC_TRY(il_copy_rows(ncolv, xp, xs, g->xil, la.row0, la.nrows, st));
    } else if (s.nslots > 0 && (g_gs_slots || lay->compacted)) {
      SlotArgs sa{};
      sa.wcol = lay->wcol; sa.wval = lay->wval; sa.slot_row = lay->slot_row; sa.wmeta = lay->wmeta;
      sa.diag = lay->diag; sa.bp = rhs; sa.x = xp; sa.omega = omega; sa.slot0 = s.slot0;
      // (XCD-contiguous mapping only for launches of several workgroups per CU, as for the SELL launches above)
      const bool xmap = g_gs_xcd_map && (s.nslots >= 768 || ncolv > 1);
      sa.nslots = s.nslots; sa.xcd_map = xmap ? 1 : 0;
      sa.ldx = xs; sa.ldb = ldb;
      if (lay->slot_entries == kBigSlot) {  // long composite rows: 2048-entry slots, 8 lanes per row
        sa.ncolv = ncolv;
        const int grid = ((xmap || ncolv > 1) ? ((s.nslots + kNumXcd - 1) / kNumXcd) * kNumXcd : s.nslots) * ncolv;
        if (sor) hipLaunchKernelGGL(gs_bigslot_kernel<true>, dim3(grid), dim3(kSlot), 0, st, sa);
        else hipLaunchKernelGGL(gs_bigslot_kernel<false>, dim3(grid), dim3(kSlot), 0, st, sa);
        HIP_TRY(hipGetLastError());
        if (il) RC_TRY(il_copy_rows(ncolv, xp, xs, g->xil, lay->lvl_ptr[s.l0], lay->lvl_ptr[s.l1] - lay->lvl_ptr[s.l0], st));
        continue;
      }
      if (il) {
        SlotIlArgs ia{};
        ia.wcol = sa.wcol; ia.wval = sa.wval; ia.slot_row = sa.slot_row; ia.wmeta = sa.wmeta; ia.diag = sa.diag;
        ia.xil = g->xil; ia.x = xp; ia.omega = omega; ia.slot0 = s.slot0; ia.nslots = s.nslots;
        ia.xcd_map = (g_gs_xcd_map && s.nslots >= 768) ? 1 : 0;
        ia.soff = (int32_t)g->ncols; ia.ldx = xs;
        const int grid = ia.xcd_map ? ((s.nslots + kNumXcd - 1) / kNumXcd) * kNumXcd : s.nslots;
        RC_TRY(launch_slot_il(ia, sor, ncolv, grid, st));
        HIP_TRY(hipGetLastError());
        continue;
      }
      // columns per workgroup: the largest of 8 / 4 / 2 / 1 that divides the block size
      const int ncv = (ncolv % 8 == 0) ? 8 : (ncolv % 4 == 0) ? 4 : (ncolv % 2 == 0) ? 2 : 1;
      sa.ncolv = ncolv / ncv;
      const int grid = ((xmap || sa.ncolv > 1) ? ((s.nslots + kNumXcd - 1) / kNumXcd) * kNumXcd : s.nslots) * sa.ncolv;
      // merged groups with long composite rows: several lanes per row in the row sums (order of additions changes,
      // so never on the unmerged schedule, which reproduces the scalar loop bit for bit)
      int lpr = 1;
      if (lay != g && ncolv == 1 && g_gs_lpr != 1) {
        const int avg = s.rows / 16;  // mean row length of the group (schedule build time)
        // measured on the 256^3 hierarchy (profiles/r02_gs_lpr_ept.log): 16 lanes pay from ~100 entries per row
        // (-15 %), 8 lanes from ~50 (-3 %); 2 and 4 lanes on rows of 10-30 entries are slower than one thread
        // (round 3, reduction trees through DPP: 16 lanes already pay from ~56 entries — the 62-entry rows of the 1.4 M-row level 2.44 -> 2.22 ms per pass)
        lpr = g_gs_lpr > 1 ? g_gs_lpr : (avg >= 56 ? 16 : avg >= 48 ? 8 : 1);
      }
      // more slots than 512-thread workgroups fit on the chip at once (4 per CU): 256-thread workgroups, 2 entries each
      int ept = 1;
      // (short rows only: -5 % on the fine level; with rows of ~27 entries the one-thread sums of the 256-thread
      // variant cost more than the second residency round)
      if (lay != g && ncolv == 1) ept = g_gs_ept > 0 ? g_gs_ept : ((s.nslots > 1024 && lpr == 1 && s.rows / 16 < 16) ? 2 : 1);
      if (lpr > 1 || ept > 1) { RC_TRY(launch_slot_lpr(sa, sor, lpr, ept, grid, st)); HIP_TRY(hipGetLastError()); continue; }
      RC_TRY(launch_slot(sa, sor, ncv, grid, st));
      HIP_TRY(hipGetLastError());
    } else {
      StreamArgs a{};
      a.rowptr = lay->rowptr; a.col = lay->col; a.val = lay->val;
      a.x = xp; a.y = xp; a.b = rhs; a.dpos = lay->dpos; a.diag = lay->diag; a.perm = nullptr; a.omega = omega;
      a.row_begin = lay->lvl_ptr[s.l0]; a.row_end = lay->lvl_ptr[s.l0 + 1];
      a.ldx = xs; a.ldy = xs; a.ldb = ldb;
      RC_TRY(sor ? launch_gs_level<M_SOR>(a, s.rows, st, ncolv) : launch_gs_level<M_GS>(a, s.rows, st, ncolv));
      if (il) RC_TRY(il_copy_rows(ncolv, xp, xs, g->xil, a.row_begin, a.row_end - a.row_begin, st));
    }
  }
  if (last && !no_scatter) {
    hipLaunchKernelGGL(scatter_perm_kernel, dim3(grid_for(g->n), ncolv), dim3(256), 0, st, (const real*)xp, g->perm, x,
                       (int)g->n, xs, (int64_t)g->n);
    HIP_TRY(hipGetLastError());
  }
  return AMGH_OK;
}

}  // namespace
