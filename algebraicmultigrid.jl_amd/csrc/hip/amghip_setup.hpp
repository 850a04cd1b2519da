// amghip_setup.hpp — the data-parallel half of the Ruge-Stuben SETUP phase on the GPU:
// classical strength of connection (strength.jl:7-37), direct interpolation (classical.jl:57-189), sparse transposes
// and the Galerkin product R*A*P (classical.jl:44, SparseArrays spmatmul).  The C/F splitting (splitting.jl:25-159) is
// an inherently sequential bucket sweep and stays on the host (libamgsetup), as the survey prescribes: the GPU hands
// it the strength pattern without its diagonal and that pattern's transpose.
//
// Matrices are Julia's CSC (colptr / rowval / nzval, 0-based int32 / f64) on HBM: `amgh_dmat`.  Every kernel walks a
// column in stored order and adds in the order the reference's loops do, so the results are BITWISE those of the host
// library (tests/test_gpu_setup.py compares structure and values exactly):
//   * one thread per column for the O(nnz) passes (strength, interpolation): no reductions across lanes at all;
//   * SpGEMM: a group of 16 lanes owns an output column and an LDS hash table; the products of ONE entry Y[k,j] are
//     spread over the lanes (rows of X[:,k] are distinct, so no two lanes touch the same accumulator), the entries of
//     Y[:,j] are taken one after the other — per accumulator the additions happen in the reference's order;
//   * transposes scatter with atomics and then sort every output column by row index (a column's keys are unique, so
//     the result does not depend on the order the atomics happened in).
#pragma once

struct amgh_dmat {
  int device = 0;
  int64_t m = 0, n = 0, nnz = 0;  // m rows, n columns
  int32_t* ptr = nullptr;         // n + 1
  int32_t* idx = nullptr;         // row indices, ascending inside a column
  double* val = nullptr;
};

namespace {

void dmat_free(amgh_dmat* M) {
  if (!M) return;
  hipFree(M->ptr); hipFree(M->idx); hipFree(M->val);
  delete M;
}

int dmat_alloc(amgh_dmat** out, int device, int64_t m, int64_t n, int64_t nnz, bool with_val = true) {
  amgh_dmat* M = new amgh_dmat;
  M->device = device; M->m = m; M->n = n; M->nnz = nnz;
  int rc = dev_alloc(&M->ptr, n + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&M->idx, nnz);
  if (rc == AMGH_OK && with_val) rc = dev_alloc(&M->val, nnz);
  if (rc != AMGH_OK) { dmat_free(M); return rc; }
  *out = M;
  return AMGH_OK;
}

// ---- strength.jl:7-37 : T = Classical(theta)(At) ------------------------------------------------------------------------
// value an entry of column i takes before dropzeros!: off-diagonal |v| if |v| >= theta * max offdiag |.|, else 0
__device__ __forceinline__ double strength_value(int32_t row, int i, double v, double thr) {
  if (row == i) return v;
  const double a = fabs(v);
  return (a >= thr) ? a : 0.0;
}
__global__ void strength_kernel(const int32_t* ap, const int32_t* ai, const double* av, int64_t n, double theta,
                                const int32_t* tp, int32_t* ti, double* tv, int32_t* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t a0 = ap[i], a1 = ap[i + 1];
  double mx = 0.0;  // find_max_off_diag (strength.jl:39-48)
  for (int32_t j = a0; j < a1; ++j)
    if (ai[j] != i) mx = fmax(mx, fabs(av[j]));
  const double thr = theta * mx;
  if (!tp) {  // count pass
    int32_t c = 0;
    for (int32_t j = a0; j < a1; ++j) c += (strength_value(ai[j], i, av[j], thr) != 0.0);
    cnt[i] = c;
    return;
  }
  // fill pass: kept entries, then scale_cols_by_largest_entry! (strength.jl:61-70: max over ALL stored values from 0)
  int32_t o = tp[i];
  double big = 0.0;
  for (int32_t j = a0; j < a1; ++j) {
    const double nv = strength_value(ai[j], i, av[j], thr);
    if (nv != 0.0) { ti[o] = ai[j]; tv[o] = nv; big = fmax(big, nv); ++o; }
  }
  for (int32_t q = tp[i]; q < o; ++q) tv[q] = tv[q] / big;
}

// pattern of a matrix without its diagonal and without explicit zeros (remove_diag, splitting.jl:8-18)
__global__ void nodiag_kernel(const int32_t* sp, const int32_t* si, const double* sv, int64_t n, const int32_t* op,
                              int32_t* oi, int32_t* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t c = 0, o = op ? op[i] : 0;
  for (int32_t j = sp[i]; j < sp[i + 1]; ++j) {
    if (si[j] != i && sv[j] != 0.0) {
      if (op) oi[o++] = si[j];
      ++c;
    }
  }
  if (!op) cnt[i] = c;
}

// ---- transpose ------------------------------------------------------------------------------------------------------------
__global__ void tr_count_kernel(const int32_t* idx, int64_t nnz, int32_t* cnt) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&cnt[idx[k]], 1);
}
__global__ void tr_fill_kernel(const int32_t* ap, const int32_t* ai, const double* av, int64_t n, int32_t* next,
                               int32_t* ti, double* tv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  for (int32_t k = ap[j]; k < ap[j + 1]; ++k) {
    const int32_t p = atomicAdd(&next[ai[k]], 1);
    ti[p] = j;
    if (tv) tv[p] = av[k];
  }
}
// every column sorted by row index (keys are unique inside a column): insertion sort by one thread, columns are short
__global__ void sort_columns_kernel(const int32_t* ptr, int32_t* idx, double* val, int64_t n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int32_t a = ptr[j], b = ptr[j + 1];
  for (int32_t p = a + 1; p < b; ++p) {
    const int32_t key = idx[p];
    const double v = val ? val[p] : 0.0;
    int32_t q = p - 1;
    while (q >= a && idx[q] > key) {
      idx[q + 1] = idx[q];
      if (val) val[q + 1] = val[q];
      --q;
    }
    idx[q + 1] = key;
    if (val) val[q + 1] = v;
  }
}

int dmat_transpose(const amgh_dmat* A, amgh_dmat** out, bool with_val, hipStream_t st) {
  amgh_dmat* T = nullptr;
  RC_TRY(dmat_alloc(&T, A->device, A->n, A->m, A->nnz, with_val));
  int32_t* cnt = nullptr;
  int rc = dev_alloc(&cnt, A->m + 1);
  if (rc == AMGH_OK && hipMemsetAsync(cnt, 0, sizeof(int32_t) * (A->m + 1), st) != hipSuccess) rc = -1001;
  if (rc == AMGH_OK && A->nnz > 0)
    hipLaunchKernelGGL(tr_count_kernel, dim3(grid_for(A->nnz)), dim3(256), 0, st, (const int32_t*)A->idx, A->nnz, cnt);
  int64_t total = 0;
  if (rc == AMGH_OK) rc = dev_exclusive_scan(cnt, T->ptr, A->m, &total, st);
  if (rc == AMGH_OK && total != A->nnz) rc = AMGH_ESTATE;
  if (rc == AMGH_OK && hipMemcpyAsync(cnt, T->ptr, sizeof(int32_t) * A->m, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = -1001;
  if (rc == AMGH_OK && A->n > 0) {
    hipLaunchKernelGGL(tr_fill_kernel, dim3((unsigned)((A->n + 255) / 256)), dim3(256), 0, st, (const int32_t*)A->ptr,
                       (const int32_t*)A->idx, (const double*)A->val, A->n, cnt, T->idx, with_val ? T->val : nullptr);
    if (A->m > 0)
      hipLaunchKernelGGL(sort_columns_kernel, dim3((unsigned)((A->m + 255) / 256)), dim3(256), 0, st,
                         (const int32_t*)T->ptr, T->idx, with_val ? T->val : nullptr, A->m);
    if (hipGetLastError() != hipSuccess) rc = -1001;
  }
  hipFree(cnt);
  if (rc != AMGH_OK) { dmat_free(T); return rc; }
  *out = T;
  return AMGH_OK;
}

// ---- classical.jl:57-189 : direct interpolation ----------------------------------------------------------------------------
// Column i of the masked T (classical.jl:58-60: At's values on T's pattern, exact zeros not stored) is walked together
// with column i of At; both are sorted by row.
struct InterpArgs {
  const int32_t *ap, *ai; const double* av;     // At
  const int32_t *tp, *ti;                        // pattern of T (a subset of At's)
  const int32_t* split;                          // 1 = C node
  const int32_t* map;                            // exclusive prefix sum of split: coarse index of a C node
  const int32_t* bp;                             // column pointers of R (fill pass), nullptr in the count pass
  int32_t* bj; double* bx; int32_t* cnt;
  int64_t n;
};
__global__ void interp_kernel(InterpArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const bool fill = a.bp != nullptr;
  if (a.split[i] == 1) {  // C node: injection
    if (fill) { a.bj[a.bp[i]] = a.map[i]; a.bx[a.bp[i]] = 1.0; }
    else a.cnt[i] = 1;
    return;
  }
  const int32_t a0 = a.ap[i], a1 = a.ap[i + 1], t0 = a.tp[i], t1 = a.tp[i + 1];
  // strong connections to C nodes, in T's order: count, and the two sums of classical.jl:104-116
  double sum_strong_pos = 0.0, sum_strong_neg = 0.0;
  int32_t c = 0;
  {
    int32_t q = a0;
    for (int32_t t = t0; t < t1; ++t) {
      const int32_t row = a.ti[t];
      while (q < a1 && a.ai[q] < row) ++q;
      if (q < a1 && a.ai[q] == row) {
        const double sval = a.av[q] * 1.0;
        if (sval != 0.0 && a.split[row] == 1) {
          ++c;
          if (sval < 0) sum_strong_neg += sval; else sum_strong_pos += sval;
        }
      }
    }
  }
  if (!fill) { a.cnt[i] = c; return; }
  double sum_all_pos = 0.0, sum_all_neg = 0.0, diag = 0.0;
  for (int32_t j = a0; j < a1; ++j) {
    const double aval = a.av[j];
    if (a.ai[j] == i) diag += aval;
    else if (aval < 0) sum_all_neg += aval;
    else sum_all_pos += aval;
  }
  double alpha, beta;
  if (sum_strong_pos == 0) { beta = 0.0; if (diag >= 0) diag += sum_all_pos; }
  else beta = sum_all_pos / sum_strong_pos;
  if (sum_strong_neg == 0) { alpha = 0.0; if (diag < 0) diag += sum_all_neg; }
  else alpha = sum_all_neg / sum_strong_neg;
  double neg_coeff = 0.0, pos_coeff = 0.0;
  if (!(fabs(diag) <= 2.220446049250313e-16)) { neg_coeff = alpha / diag; pos_coeff = beta / diag; }
  int32_t nz = a.bp[i];
  int32_t q = a0;
  for (int32_t t = t0; t < t1; ++t) {
    const int32_t row = a.ti[t];
    while (q < a1 && a.ai[q] < row) ++q;
    if (q < a1 && a.ai[q] == row) {
      const double sval = a.av[q] * 1.0;
      if (sval != 0.0 && a.split[row] == 1) {
        a.bj[nz] = a.map[row];
        a.bx[nz] = (sval < 0) ? fabs(neg_coeff * sval) : fabs(pos_coeff * sval);
        ++nz;
      }
    }
  }
}

// ---- X * Y (SparseArrays spmatmul semantics) ---------------------------------------------------------------------------------
// A group of kSpgLanes lanes owns output column j and an LDS open-addressing table.  For p over Y[:,j] IN ORDER the
// lanes take the entries q of X[:,k] (k = row of entry p): distinct rows, so no two lanes add to the same accumulator
// inside one p step, and a group-wide barrier separates the steps: per output row the additions happen in the
// reference's order (p ascending, then q ascending — one q per p and row).  Rows come out unsorted and are sorted by
// sort_columns_kernel afterwards.
constexpr int kSpgLanes = 16;
constexpr int kSpgCap = 512;                       // table slots per group (columns of the product up to ~440 entries)
constexpr int kSpgThreads = 64;                    // one wave = 4 groups: 4 * 512 * (12 + 12) B = 48 KiB of LDS
constexpr int kSpgGroups = kSpgThreads / kSpgLanes;

struct SpgemmArgs {
  const int32_t *xp, *xi; const double* xv;
  const int32_t *yp, *yi; const double* yv;
  const int32_t* cp;   // output column pointers (fill pass) or nullptr (count pass)
  int32_t* ci; double* cv; int32_t* cnt;
  int64_t ncols;
  int32_t* overflow;
};
__global__ __launch_bounds__(kSpgThreads) void spgemm_kernel(SpgemmArgs a) {
  __shared__ int32_t s_key[kSpgGroups][kSpgCap];
  __shared__ double s_val[kSpgGroups][kSpgCap];
  __shared__ int32_t s_dkey[kSpgGroups][kSpgCap];   // the column's rows, dense (for the rank sort)
  __shared__ double s_dval[kSpgGroups][kSpgCap];
  __shared__ int32_t s_cnt[kSpgGroups], s_pos[kSpgGroups];
  const int grp = threadIdx.x / kSpgLanes, ln = threadIdx.x % kSpgLanes;
  const int64_t j = (int64_t)blockIdx.x * kSpgGroups + grp;
  const bool live = j < a.ncols;
  int32_t* key = s_key[grp];
  double* val = s_val[grp];
  for (int t = ln; t < kSpgCap; t += kSpgLanes) key[t] = -1;
  if (ln == 0) { s_cnt[grp] = 0; s_pos[grp] = 0; }
  __syncthreads();
  const bool fill = a.cp != nullptr;
  const int32_t y0 = live ? a.yp[j] : 0, y1 = live ? a.yp[j + 1] : 0;
  // the four groups of the wave step together (the barrier between two entries of Y[:,j] is the workgroup's)
  int32_t maxlen = y1 - y0;
#pragma unroll
  for (int w = kSpgLanes; w < kWave; w <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, w, kWave));
  for (int32_t step = 0; step < maxlen; ++step) {
    const int32_t p = y0 + step;
    if (p < y1) {
      const int32_t k = a.yi[p];
      const double ykj = fill ? a.yv[p] : 0.0;
      for (int32_t q = a.xp[k] + ln; q < a.xp[k + 1]; q += kSpgLanes) {
        const int32_t i = a.xi[q];
        uint32_t hsh = ((uint32_t)i * 2654435761u) & (kSpgCap - 1);
        int probes = 0;
        for (;;) {
          const int32_t old = atomicCAS(&key[hsh], -1, i);
          if (old == -1) {  // new row of this column
            atomicAdd(&s_cnt[grp], 1);
            if (fill) val[hsh] = a.xv[q] * ykj;
            break;
          }
          if (old == i) {
            if (fill) val[hsh] += a.xv[q] * ykj;
            break;
          }
          hsh = (hsh + 1) & (kSpgCap - 1);
          if (++probes >= kSpgCap) { *a.overflow = 1; break; }
        }
      }
    }
    __syncthreads();
  }
  const int32_t c = s_cnt[grp];
  if (live && c > kSpgCap - kSpgCap / 8) *a.overflow = 1;  // table too full to trust the probe bound
  if (!fill) { if (live && ln == 0) a.cnt[j] = c; return; }
  // rows out in ascending order (SparseArrays keeps columns sorted): table -> dense list -> rank of every row
  int32_t* dkey = s_dkey[grp];
  double* dval = s_dval[grp];
  for (int t = ln; t < kSpgCap; t += kSpgLanes)
    if (key[t] != -1) {
      const int32_t o = atomicAdd(&s_pos[grp], 1);
      dkey[o] = key[t];
      dval[o] = val[t];
    }
  __syncthreads();
  if (!live) return;
  const int32_t base = a.cp[j];
  for (int32_t e = ln; e < c; e += kSpgLanes) {
    const int32_t ke = dkey[e];
    int32_t rank = 0;
    for (int32_t f = 0; f < c; ++f) rank += (dkey[f] < ke);
    a.ci[base + rank] = ke;
    a.cv[base + rank] = dval[e];
  }
}

int dmat_spgemm(const amgh_dmat* X, const amgh_dmat* Y, amgh_dmat** out, hipStream_t st) {
  if (X->n != Y->m) return AMGH_EINVAL;
  const int64_t n = Y->n;
  int32_t *cnt = nullptr, *ovf = nullptr;
  RC_TRY(dev_alloc(&cnt, n + 1));
  int rc = dev_alloc(&ovf, 1);
  amgh_dmat* Cm = nullptr;
  if (rc == AMGH_OK && hipMemsetAsync(ovf, 0, sizeof(int32_t), st) != hipSuccess) rc = -1001;
  SpgemmArgs a{};
  a.xp = X->ptr; a.xi = X->idx; a.xv = X->val; a.yp = Y->ptr; a.yi = Y->idx; a.yv = Y->val;
  a.ncols = n; a.cnt = cnt; a.overflow = ovf;
  const unsigned grid = (unsigned)std::max<int64_t>(1, (n + kSpgGroups - 1) / kSpgGroups);
  if (rc == AMGH_OK && n > 0) hipLaunchKernelGGL(spgemm_kernel, dim3(grid), dim3(kSpgThreads), 0, st, a);
  amgh_dmat* C0 = nullptr;
  if (rc == AMGH_OK) {
    // column pointers first (the count pass), then the matrix
    int32_t* cp = nullptr;
    rc = dev_alloc(&cp, n + 1);
    int64_t total = 0;
    if (rc == AMGH_OK) rc = dev_exclusive_scan(cnt, cp, n, &total, st);
    int32_t hov = 0;
    if (rc == AMGH_OK && hipMemcpy(&hov, ovf, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = -1001;
    if (rc == AMGH_OK && hov) rc = AMGH_EUNSUPPORTED;  // a column of the product outgrew the LDS table
    if (rc == AMGH_OK) rc = dmat_alloc(&C0, X->device, X->m, n, total);
    if (rc == AMGH_OK) {
      hipFree(C0->ptr);
      C0->ptr = cp;
      cp = nullptr;
      a.cp = C0->ptr; a.ci = C0->idx; a.cv = C0->val;
      if (n > 0) hipLaunchKernelGGL(spgemm_kernel, dim3(grid), dim3(kSpgThreads), 0, st, a);
      if (hipStreamSynchronize(st) != hipSuccess) rc = -1001;
      Cm = C0;
    }
    hipFree(cp);
  }
  hipFree(cnt); hipFree(ovf);
  if (rc != AMGH_OK) { dmat_free(C0); return rc; }
  *out = Cm;
  return AMGH_OK;
}

int dmat_check(const amgh_dmat* M) { return M ? AMGH_OK : AMGH_EINVAL; }

}  // namespace

extern "C" {

int amgh_dmat_upload(amgh_dmat_t** out, int device, int64_t m, int64_t n, const int32_t* colptr, const int32_t* rowval,
                     const double* nzval) {
  if (!out || m < 0 || n < 0 || !colptr) return AMGH_EINVAL;
  *out = nullptr;
  if (m >= INT32_MAX || n >= INT32_MAX) return AMGH_EUNSUPPORTED;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(device));
  const int64_t nnz = colptr[n];
  if (colptr[0] != 0 || nnz < 0 || (nnz > 0 && (!rowval || !nzval))) return AMGH_EINVAL;
  amgh_dmat* M = nullptr;
  RC_TRY(dmat_alloc(&M, device, m, n, nnz));
  hipError_t e = staged_copy(M->ptr, colptr, sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice);
  if (e == hipSuccess && nnz) e = staged_copy(M->idx, rowval, sizeof(int32_t) * nnz, hipMemcpyHostToDevice);
  if (e == hipSuccess && nnz) e = staged_copy(M->val, nzval, sizeof(double) * nnz, hipMemcpyHostToDevice);
  if (e != hipSuccess) { dmat_free(M); return -(1000 + (int)e); }
  *out = M;
  return AMGH_OK;
}
int amgh_dmat_download(const amgh_dmat_t* M, int32_t* colptr, int32_t* rowval, double* nzval) {
  RC_TRY(dmat_check(M));
  HIP_TRY(hipSetDevice(M->device));
  HIP_TRY(hipStreamSynchronize(nullptr));   // the staged copies run on their own streams: whatever produced M (null stream) is done first
  if (colptr) HIP_TRY(staged_copy(colptr, M->ptr, sizeof(int32_t) * (M->n + 1), hipMemcpyDeviceToHost));
  if (rowval && M->nnz) HIP_TRY(staged_copy(rowval, M->idx, sizeof(int32_t) * M->nnz, hipMemcpyDeviceToHost));
  if (nzval && M->nnz && M->val) HIP_TRY(staged_copy(nzval, M->val, sizeof(double) * M->nnz, hipMemcpyDeviceToHost));
  return AMGH_OK;
}
void amgh_dmat_free(amgh_dmat_t* M) {
  if (!M) return;
  hipSetDevice(M->device);
  dmat_free(M);
}
int64_t amgh_dmat_rows(const amgh_dmat_t* M) { return M ? M->m : -1; }
int64_t amgh_dmat_cols(const amgh_dmat_t* M) { return M ? M->n : -1; }
int64_t amgh_dmat_nnz(const amgh_dmat_t* M) { return M ? M->nnz : -1; }

// *same = 1 when the two matrices have identical arrays (A == copy(A') is how the host mirror detects symmetry)
__global__ void dmat_equal_kernel(const int32_t* p1, const int32_t* i1, const double* v1, const int32_t* p2,
                                  const int32_t* i2, const double* v2, int64_t n, int64_t nnz, int32_t* differ) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz + n + 1; k += (int64_t)gridDim.x * blockDim.x) {
    bool d;
    if (k <= n) d = p1[k] != p2[k];
    else { const int64_t q = k - n - 1; d = i1[q] != i2[q] || !(v1[q] == v2[q]); }
    if (d) *differ = 1;
  }
}
int amgh_dmat_equal(const amgh_dmat_t* A, const amgh_dmat_t* B, int* same) {
  RC_TRY(dmat_check(A));
  RC_TRY(dmat_check(B));
  if (!same) return AMGH_EINVAL;
  *same = 0;
  if (A->m != B->m || A->n != B->n || A->nnz != B->nnz) return AMGH_OK;
  HIP_TRY(hipSetDevice(A->device));
  int32_t* differ = nullptr;
  RC_TRY(dev_alloc(&differ, 1));
  HIP_TRY(hipMemset(differ, 0, sizeof(int32_t)));
  hipLaunchKernelGGL(dmat_equal_kernel, dim3(grid_for(A->nnz + A->n + 1)), dim3(256), 0, nullptr, (const int32_t*)A->ptr,
                     (const int32_t*)A->idx, (const double*)A->val, (const int32_t*)B->ptr, (const int32_t*)B->idx,
                     (const double*)B->val, A->n, A->nnz, differ);
  int32_t h = 1;
  hipError_t e = hipMemcpy(&h, differ, sizeof(int32_t), hipMemcpyDeviceToHost);
  hipFree(differ);
  if (e != hipSuccess) return -(1000 + (int)e);
  *same = h == 0;
  return AMGH_OK;
}

int amgh_setup_transpose(const amgh_dmat_t* A, amgh_dmat_t** At) {
  RC_TRY(dmat_check(A));
  if (!At) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(A->device));
  RC_TRY(dmat_transpose(A, At, true, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  return AMGH_OK;
}

// S, T = Classical(theta)(At)  (strength.jl:7-37): T on At's columns, S = T'.  Sn / Tn (optional): the patterns the
// C/F splitting works on — S without its diagonal (remove_diag, splitting.jl:8-18) and the transpose of that.
int amgh_setup_classical_strength(const amgh_dmat_t* At, double theta, amgh_dmat_t** S, amgh_dmat_t** T,
                                  amgh_dmat_t** Sn, amgh_dmat_t** Tn) {
  RC_TRY(dmat_check(At));
  if (!S || !T || At->m != At->n) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(At->device));
  const int64_t n = At->n;
  const unsigned grid = (unsigned)std::max<int64_t>(1, (n + 255) / 256);
  int32_t* cnt = nullptr;
  RC_TRY(dev_alloc(&cnt, n + 1));
  amgh_dmat *Tm = nullptr, *Sm = nullptr, *Tnm = nullptr, *Snm = nullptr;
  int rc = AMGH_OK;
  if (n > 0)
    hipLaunchKernelGGL(strength_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)At->ptr, (const int32_t*)At->idx,
                       (const double*)At->val, n, theta, (const int32_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, cnt);
  int32_t* tp = nullptr;
  rc = dev_alloc(&tp, n + 1);
  int64_t total = 0;
  if (rc == AMGH_OK) rc = dev_exclusive_scan(cnt, tp, n, &total, nullptr);
  if (rc == AMGH_OK) rc = dmat_alloc(&Tm, At->device, n, n, total);
  if (rc == AMGH_OK) {
    hipFree(Tm->ptr);
    Tm->ptr = tp;
    tp = nullptr;
    if (n > 0)
      hipLaunchKernelGGL(strength_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)At->ptr,
                         (const int32_t*)At->idx, (const double*)At->val, n, theta, (const int32_t*)Tm->ptr, Tm->idx,
                         Tm->val, (int32_t*)nullptr);
    rc = dmat_transpose(Tm, &Sm, true, nullptr);
  }
  if (rc == AMGH_OK && Sn && Tn) {
    // Tn = T without diagonal / zeros (= remove_diag(S)' because S = T'), Sn = Tn'
    if (n > 0)
      hipLaunchKernelGGL(nodiag_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)Tm->ptr, (const int32_t*)Tm->idx,
                         (const double*)Tm->val, n, (const int32_t*)nullptr, (int32_t*)nullptr, cnt);
    int32_t* np = nullptr;
    rc = dev_alloc(&np, n + 1);
    int64_t tot2 = 0;
    if (rc == AMGH_OK) rc = dev_exclusive_scan(cnt, np, n, &tot2, nullptr);
    if (rc == AMGH_OK) rc = dmat_alloc(&Tnm, At->device, n, n, tot2, false);
    if (rc == AMGH_OK) {
      hipFree(Tnm->ptr);
      Tnm->ptr = np;
      np = nullptr;
      if (n > 0)
        hipLaunchKernelGGL(nodiag_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)Tm->ptr,
                           (const int32_t*)Tm->idx, (const double*)Tm->val, n, (const int32_t*)Tnm->ptr, Tnm->idx,
                           (int32_t*)nullptr);
      rc = dmat_transpose(Tnm, &Snm, false, nullptr);
    }
    hipFree(np);
  }
  hipFree(tp); hipFree(cnt);
  if (rc == AMGH_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = -1001;
  if (rc != AMGH_OK) { dmat_free(Tm); dmat_free(Sm); dmat_free(Tnm); dmat_free(Snm); return rc; }
  *S = Sm; *T = Tm;
  if (Sn && Tn) { *Sn = Snm; *Tn = Tnm; }
  return AMGH_OK;
}

// ---- strength.jl:77-122 : S = SymmetricStrength(theta)(A, bsr_flag) --------------------------------------------------------
// diags[i] = |sum of the stored diagonal entries of column i|  (strength.jl:94-104)
__global__ void sym_diag_kernel(const int32_t* ap, const int32_t* ai, const double* av, int64_t n, double* diags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = 0.0;
  for (int32_t j = ap[i]; j < ap[i + 1]; ++j)
    if (ai[j] == i) d += av[j];
  diags[i] = fabs(d);
}
// value an entry of column i takes before dropzeros!: 0 for an off-diagonal entry with v * v < theta^2 |a_ii| |a_rr|
// (strength.jl:106-117), |v| otherwise (:121; stored zeros go with the dropped ones)
__device__ __forceinline__ double sym_strength_value(int32_t row, int i, double v, double eps_aii, const double* diags) {
  if (row != i && v * v < eps_aii * diags[row]) return 0.0;
  return fabs(v);
}
__global__ void sym_strength_kernel(const int32_t* ap, const int32_t* ai, const double* av, int64_t n, double theta, const double* diags,
                                    const int32_t* sp, int32_t* si, double* sv, int32_t* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t a0 = ap[i], a1 = ap[i + 1];
  const double eps_aii = theta * theta * diags[i];
  if (!sp) {  // count pass
    int32_t c = 0;
    for (int32_t j = a0; j < a1; ++j) c += (sym_strength_value(ai[j], i, av[j], eps_aii, diags) != 0.0);
    cnt[i] = c;
    return;
  }
  // fill pass: kept entries, then scale_cols_by_largest_entry! (strength.jl:61-70: max over the stored values, from 0)
  int32_t o = sp[i];
  double big = 0.0;
  for (int32_t j = a0; j < a1; ++j) {
    const double nv = sym_strength_value(ai[j], i, av[j], eps_aii, diags);
    if (nv != 0.0) { si[o] = ai[j]; sv[o] = nv; big = fmax(big, nv); ++o; }
  }
  for (int32_t q = sp[i]; q < o; ++q) sv[q] = sv[q] / big;
}
__global__ void fill_ones_kernel(double* v, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) v[k] = 1.0;
}

int amgh_setup_symmetric_strength(const amgh_dmat_t* A, double theta, int bsr_flag, amgh_dmat_t** S) {
  RC_TRY(dmat_check(A));
  if (!S || A->m != A->n) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(A->device));
  const int64_t n = A->n;
  const unsigned grid = (unsigned)std::max<int64_t>(1, (n + 255) / 256);
  amgh_dmat* Sm = nullptr;
  if (bsr_flag && theta == 0.0) {   // strength.jl:81-84: the pattern of A, every value one
    RC_TRY(dmat_alloc(&Sm, A->device, n, n, A->nnz));
    int rc = AMGH_OK;
    if (hipMemcpyAsync(Sm->ptr, A->ptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, nullptr) != hipSuccess) rc = -1001;
    if (rc == AMGH_OK && A->nnz > 0 && hipMemcpyAsync(Sm->idx, A->idx, sizeof(int32_t) * (size_t)A->nnz, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) rc = -1001;
    if (rc == AMGH_OK && A->nnz > 0)
      hipLaunchKernelGGL(fill_ones_kernel, dim3((unsigned)std::min<int64_t>(65535, (A->nnz + 255) / 256)), dim3(256), 0, nullptr, Sm->val, A->nnz);
    if (rc == AMGH_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = -1001;
    if (rc != AMGH_OK) { dmat_free(Sm); return rc; }
    *S = Sm;
    return AMGH_OK;
  }
  double* diags = nullptr;
  int32_t *cnt = nullptr, *sp = nullptr;
  int rc = dev_alloc(&diags, n + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&cnt, n + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&sp, n + 1);
  int64_t total = 0;
  if (rc == AMGH_OK && n > 0) {
    hipLaunchKernelGGL(sym_diag_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)A->ptr, (const int32_t*)A->idx, (const double*)A->val, n, diags);
    hipLaunchKernelGGL(sym_strength_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)A->ptr, (const int32_t*)A->idx, (const double*)A->val, n,
                       theta, (const double*)diags, (const int32_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, cnt);
  }
  if (rc == AMGH_OK) rc = dev_exclusive_scan(cnt, sp, n, &total, nullptr);
  if (rc == AMGH_OK) rc = dmat_alloc(&Sm, A->device, n, n, total);
  if (rc == AMGH_OK) {
    hipFree(Sm->ptr);
    Sm->ptr = sp;
    sp = nullptr;
    if (n > 0)
      hipLaunchKernelGGL(sym_strength_kernel, dim3(grid), dim3(256), 0, nullptr, (const int32_t*)A->ptr, (const int32_t*)A->idx, (const double*)A->val, n,
                         theta, (const double*)diags, (const int32_t*)Sm->ptr, Sm->idx, Sm->val, (int32_t*)nullptr);
    if (hipStreamSynchronize(nullptr) != hipSuccess) rc = -1001;
  }
  hipFree(diags); hipFree(cnt); hipFree(sp);
  if (rc != AMGH_OK) { dmat_free(Sm); return rc; }
  *S = Sm;
  return AMGH_OK;
}

// ---- aggregation.jl:161-193 : T, Bc = fit_candidates(AggOp, B::Vector) --------------------------------------------------------
// column i of A = AggOp' lists the fine nodes of aggregate i (ascending); one thread walks it twice, in stored order: the
// sum of squares (norm_col, aggregation.jl:232-240), then B[row] * (1 / norm) — the host library's operations in its order
__global__ void fit_vector_kernel(const int32_t* ap, const int32_t* ai, double* av, int64_t ncol, const double* B, double tol, double* Bc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncol) return;
  double s = 0.0;
  for (int32_t j = ap[i]; j < ap[i + 1]; ++j) { const double v = B[ai[j]]; s += v * v; }
  const double norm_i = sqrt(s);
  const double threshold_i = tol * norm_i;
  double scale = 0.0, r = 0.0;
  if (norm_i > threshold_i) { scale = 1.0 / norm_i; r = norm_i; }
  Bc[i] = r;
  for (int32_t j = ap[i]; j < ap[i + 1]; ++j) av[j] = B[ai[j]] * scale;
}
int amgh_setup_fit_candidates_vector(const amgh_dmat_t* AggOp, const double* B, double tol, amgh_dmat_t** T, double* Bc) {
  RC_TRY(dmat_check(AggOp));
  if (!T || !B || !Bc) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(AggOp->device));
  const int64_t n_fine = AggOp->n, n_coarse = AggOp->m;
  amgh_dmat* A = nullptr;
  RC_TRY(dmat_transpose(AggOp, &A, true, nullptr));   // n_fine x n_coarse, rows ascending inside a column
  double *dB = nullptr, *dBc = nullptr;
  int rc = dev_alloc(&dB, std::max<int64_t>(n_fine, 1));
  if (rc == AMGH_OK) rc = dev_alloc(&dBc, std::max<int64_t>(n_coarse, 1));
  if (rc == AMGH_OK && n_fine > 0 && staged_copy(dB, B, sizeof(double) * (size_t)n_fine, hipMemcpyHostToDevice) != hipSuccess) rc = -1001;
  if (rc == AMGH_OK && n_coarse > 0) {
    hipLaunchKernelGGL(fit_vector_kernel, dim3((unsigned)((n_coarse + 255) / 256)), dim3(256), 0, nullptr, (const int32_t*)A->ptr, (const int32_t*)A->idx,
                       A->val, n_coarse, (const double*)dB, tol, dBc);
    if (hipStreamSynchronize(nullptr) != hipSuccess) rc = -1001;
    if (rc == AMGH_OK && hipMemcpy(Bc, dBc, sizeof(double) * (size_t)n_coarse, hipMemcpyDeviceToHost) != hipSuccess) rc = -1001;
  }
  hipFree(dB); hipFree(dBc);
  if (rc != AMGH_OK) { dmat_free(A); return rc; }
  *T = A;
  return AMGH_OK;
}

// P, R = direct_interpolation(At, T, splitting)  (classical.jl:57-189): R is nc x n in CSC (one column per fine node,
// = the CSR arrays of P), P = R' (copy).  splitting: host array, 1 = C node, 0 = F node.
int amgh_setup_direct_interpolation(const amgh_dmat_t* At, const amgh_dmat_t* T, const int32_t* splitting,
                                    amgh_dmat_t** R, amgh_dmat_t** P) {
  RC_TRY(dmat_check(At));
  RC_TRY(dmat_check(T));
  if (!splitting || !R || !P || At->m != At->n || T->n != At->n) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(At->device));
  const int64_t n = At->n;
  const unsigned grid = (unsigned)std::max<int64_t>(1, (n + 255) / 256);
  int32_t *split = nullptr, *map = nullptr, *cnt = nullptr, *bp = nullptr;
  amgh_dmat *Rm = nullptr, *Pm = nullptr;
  int rc = dev_upload(&split, splitting, n);
  if (rc == AMGH_OK) rc = dev_alloc(&map, n + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&cnt, n + 1);
  if (rc == AMGH_OK) rc = dev_alloc(&bp, n + 1);
  int64_t nc = 0, nnzc = 0;
  if (rc == AMGH_OK) rc = dev_exclusive_scan(split, map, n, &nc, nullptr);  // coarse index = prefix sum (classical.jl:180-186)
  InterpArgs a{};
  a.ap = At->ptr; a.ai = At->idx; a.av = At->val; a.tp = T->ptr; a.ti = T->idx; a.split = split; a.map = map;
  a.cnt = cnt; a.n = n;
  if (rc == AMGH_OK && n > 0) hipLaunchKernelGGL(interp_kernel, dim3(grid), dim3(256), 0, nullptr, a);
  if (rc == AMGH_OK) rc = dev_exclusive_scan(cnt, bp, n, &nnzc, nullptr);
  if (rc == AMGH_OK) rc = dmat_alloc(&Rm, At->device, nnzc > 0 ? nc : 0, n, nnzc);
  if (rc == AMGH_OK) {
    hipFree(Rm->ptr);
    Rm->ptr = bp;
    bp = nullptr;
    a.bp = Rm->ptr; a.bj = Rm->idx; a.bx = Rm->val;
    if (n > 0) hipLaunchKernelGGL(interp_kernel, dim3(grid), dim3(256), 0, nullptr, a);
    rc = dmat_transpose(Rm, &Pm, true, nullptr);
  }
  hipFree(split); hipFree(map); hipFree(cnt); hipFree(bp);
  if (rc == AMGH_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = -1001;
  if (rc != AMGH_OK) { dmat_free(Rm); dmat_free(Pm); return rc; }
  *R = Rm; *P = Pm;
  return AMGH_OK;
}

// C = X * Y (the two products of R * A * P, classical.jl:44).  AMGH_EUNSUPPORTED when a column of the product has more
// entries than the LDS table of a lane group holds (the caller then uses the host library for this product).
// ---- aggregation.jl:30-59 : P = JacobiProlongation(omega)(A, T) = T - (omega * D^-1 A) * T, D_i = sum_j |a_ij| -----------------
} // extern "C"
namespace {
// column i of At = row i of A with its entries in ascending column order: the order the host loop adds them in
__global__ void row_abs_inv_kernel(const int32_t* tp, const double* tv, int64_t n, double* dinv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = 0.0;
  for (int32_t j = tp[i]; j < tp[i + 1]; ++j) d += fabs(tv[j]);
  dinv[i] = d != 0.0 ? 1.0 / d : d;
}
__global__ void scale_rows_kernel(const int32_t* idx, const double* val, int64_t nnz, const double* dinv, double omega,
                                  double* out) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
    double v = val[k] * dinv[idx[k]];   // scale_rows! (aggregation.jl:49-59) ...
    v *= omega;                         // ... then rmul!(., omega): two roundings, as on the host
    out[k] = v;
  }
}
// C = X - Y column by column (both sorted by row), exact zeros not stored.  cp == nullptr: count pass.
__global__ void sparse_sub_kernel(const int32_t* xp, const int32_t* xi, const double* xv, const int32_t* yp,
                                  const int32_t* yi, const double* yv, int64_t n, const int32_t* cp, int32_t* ci,
                                  double* cv, int32_t* cnt) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int32_t a = xp[j], ae = xp[j + 1], b = yp[j], be = yp[j + 1], c = 0;
  const int32_t base = cp ? cp[j] : 0;
  while (a < ae || b < be) {
    int32_t row;
    double v;
    if (b >= be || (a < ae && xi[a] < yi[b])) { row = xi[a]; v = xv[a] - 0.0; ++a; }
    else if (a >= ae || yi[b] < xi[a]) { row = yi[b]; v = 0.0 - yv[b]; ++b; }
    else { row = xi[a]; v = xv[a] - yv[b]; ++a; ++b; }
    if (v != 0.0) {
      if (cp) { ci[base + c] = row; cv[base + c] = v; }
      ++c;
    }
  }
  if (!cp) cnt[j] = c;
}
}  // namespace
extern "C" {

int amgh_setup_jacobi_prolongation(const amgh_dmat_t* A, const amgh_dmat_t* T, double omega, amgh_dmat_t** P) {
  RC_TRY(dmat_check(A));
  RC_TRY(dmat_check(T));
  if (!P || A->m != A->n || T->m != A->n) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(A->device));
  const int64_t n = A->n, nc = T->n;
  amgh_dmat *At = nullptr, *DS = nullptr, *W = nullptr, *C = nullptr;
  double* dinv = nullptr;
  int32_t *cnt = nullptr, *cp = nullptr;
  int rc = dmat_transpose(A, &At, true, nullptr);
  if (rc == AMGH_OK) rc = dev_alloc(&dinv, n);
  if (rc == AMGH_OK && n > 0)
    hipLaunchKernelGGL(row_abs_inv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const int32_t*)At->ptr,
                       (const double*)At->val, n, dinv);
  if (rc == AMGH_OK) rc = dmat_alloc(&DS, A->device, n, n, A->nnz);
  if (rc == AMGH_OK) {
    if (hipMemcpyAsync(DS->ptr, A->ptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToDevice, nullptr) != hipSuccess ||
        hipMemcpyAsync(DS->idx, A->idx, sizeof(int32_t) * A->nnz, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) rc = -1001;
    if (rc == AMGH_OK && A->nnz > 0)
      hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)grid_for(A->nnz)), dim3(256), 0, nullptr, (const int32_t*)A->idx,
                         (const double*)A->val, A->nnz, (const double*)dinv, omega, DS->val);
  }
  if (rc == AMGH_OK) rc = dmat_spgemm(DS, T, &W, nullptr);      // AMGH_EUNSUPPORTED: a column outgrew the table
  if (rc == AMGH_OK) rc = dev_alloc(&cnt, nc + 1);
  if (rc == AMGH_OK && nc > 0)
    hipLaunchKernelGGL(sparse_sub_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, nullptr, (const int32_t*)T->ptr,
                       (const int32_t*)T->idx, (const double*)T->val, (const int32_t*)W->ptr, (const int32_t*)W->idx,
                       (const double*)W->val, nc, (const int32_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, cnt);
  int64_t total = 0;
  if (rc == AMGH_OK) rc = dev_alloc(&cp, nc + 1);
  if (rc == AMGH_OK) rc = dev_exclusive_scan(cnt, cp, nc, &total, nullptr);
  if (rc == AMGH_OK) rc = dmat_alloc(&C, A->device, n, nc, total);
  if (rc == AMGH_OK) {
    hipFree(C->ptr);
    C->ptr = cp;
    cp = nullptr;
    if (nc > 0)
      hipLaunchKernelGGL(sparse_sub_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, nullptr, (const int32_t*)T->ptr,
                         (const int32_t*)T->idx, (const double*)T->val, (const int32_t*)W->ptr, (const int32_t*)W->idx,
                         (const double*)W->val, nc, (const int32_t*)C->ptr, C->idx, C->val, (int32_t*)nullptr);
    if (hipStreamSynchronize(nullptr) != hipSuccess) rc = -1001;
  }
  dmat_free(At); dmat_free(DS); dmat_free(W);
  hipFree(dinv); hipFree(cnt); hipFree(cp);
  if (rc != AMGH_OK) { dmat_free(C); return rc; }
  *P = C;
  return AMGH_OK;
}

int amgh_setup_spgemm(const amgh_dmat_t* X, const amgh_dmat_t* Y, amgh_dmat_t** C) {
  RC_TRY(dmat_check(X));
  RC_TRY(dmat_check(Y));
  if (!C) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(X->device));
  return dmat_spgemm(X, Y, C, nullptr);
}

}  // extern "C"
