// amg_setup.cpp — host-side AMG setup phase (CPU, C++17 + OpenMP).
//
// Builds the Multilevel/Level hierarchy (A, P, R per level) that the HIP
// V-cycle in libamghip consumes.  Behavioural restatement of the reference's
// setup phase, written against its documented semantics; every routine cites
// the reference lines it follows (paths relative to /root/reference).  All
// matrices are compressed sparse COLUMN (colptr/rowval/nzval) exactly as the
// reference holds them, but 0-based with int32 indices.
//
// Not on the GPU hot path (sequential graph
// algorithms, run once).  See include/amgsetup.h for the C ABI.

#include "../../../include/amgsetup.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <sys/mman.h>

#ifdef _OPENMP
#include <omp.h>
#endif

struct amgs_mat {
  int64_t m = 0, n = 0;  // rows, cols
  std::vector<int32_t> colptr;  // n+1
  std::vector<int32_t> rowval;
  std::vector<double> nzval;
  int64_t nnz() const { return (int64_t)rowval.size(); }
};

namespace {

thread_local std::string g_err;

using Mat = amgs_mat;
using MatP = std::unique_ptr<Mat>;

constexpr int F_NODE = 0, C_NODE = 1, U_NODE = 2;  // splitting.jl:1-3

void check_nnz(int64_t nnz) {
  if (nnz >= (int64_t)std::numeric_limits<int32_t>::max())
    throw std::runtime_error("matrix exceeds int32 nnz range");
}

MatP make(int64_t m, int64_t n) {
  MatP A(new Mat);
  A->m = m;
  A->n = n;
  A->colptr.assign(n + 1, 0);
  return A;
}

int num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// dropzeros!(A): remove stored entries whose value is exactly zero.
// Count per column in parallel, prefix, compact in parallel (order inside a column is kept).
void dropzeros(Mat& A) {
  const int64_t n = A.n;
  std::vector<int32_t> np(n + 1, 0);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j) {
    int32_t c = 0;
    for (int32_t k = A.colptr[j]; k < A.colptr[j + 1]; ++k) c += (A.nzval[k] != 0.0);
    np[j + 1] = c;
  }
  for (int64_t j = 0; j < n; ++j) np[j + 1] += np[j];
  const int64_t w = np[n];
  if (w == A.nnz()) return;
  std::vector<int32_t> rv(w);
  std::vector<double> nz(w);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j) {
    int32_t o = np[j];
    for (int32_t k = A.colptr[j]; k < A.colptr[j + 1]; ++k)
      if (A.nzval[k] != 0.0) {
        rv[o] = A.rowval[k];
        nz[o] = A.nzval[k];
        ++o;
      }
  }
  A.colptr.swap(np);
  A.rowval.swap(rv);
  A.nzval.swap(nz);
}

// copy(A') — counting-sort transpose; rows inside each output column ascend.
MatP transpose_serial(const Mat& A) {
  MatP T = make(A.n, A.m);
  const int64_t nnz = A.nnz();
  T->rowval.resize(nnz);
  T->nzval.resize(nnz);
  std::vector<int32_t>& tp = T->colptr;
  for (int64_t k = 0; k < nnz; ++k) tp[A.rowval[k] + 1]++;
  for (int64_t i = 0; i < A.m; ++i) tp[i + 1] += tp[i];
  std::vector<int32_t> next(tp.begin(), tp.end() - 1);
  for (int64_t j = 0; j < A.n; ++j) {
    for (int32_t k = A.colptr[j]; k < A.colptr[j + 1]; ++k) {
      int32_t p = next[A.rowval[k]]++;
      T->rowval[p] = (int32_t)j;
      T->nzval[p] = A.nzval[k];
    }
  }
  return T;
}

// Parallel copy(A'): a stable two-level counting sort.  Source columns are cut into one contiguous chunk
// per thread (balanced by entries); entries are first scattered, chunk by chunk, into buckets of kBucketRows
// consecutive rows (stable: chunks are in column order), then every bucket is counting-sorted by row on its
// own (cache-resident).  Same output as transpose_serial, bit for bit.
MatP transpose(const Mat& A) {
  const int64_t nnz = A.nnz();
  const int T = num_threads();
  if (T <= 1 || nnz < (int64_t)1 << 18) return transpose_serial(A);
  constexpr int kShift = 13;  // 8192 rows per bucket
  const int64_t nb = (A.m + ((int64_t)1 << kShift) - 1) >> kShift;
  MatP Tm = make(A.n, A.m);
  Tm->rowval.resize(nnz);
  Tm->nzval.resize(nnz);
  // column chunks balanced by nnz
  std::vector<int64_t> cbeg(T + 1, A.n);
  cbeg[0] = 0;
  for (int t = 1; t < T; ++t) {
    const int64_t target = nnz * t / T;
    cbeg[t] = std::upper_bound(A.colptr.begin(), A.colptr.end(), (int32_t)target) - A.colptr.begin() - 1;
    if (cbeg[t] < cbeg[t - 1]) cbeg[t] = cbeg[t - 1];
    if (cbeg[t] > A.n) cbeg[t] = A.n;
  }
  std::vector<int64_t> cnt((size_t)T * nb, 0);
  std::vector<int64_t> bstart(nb + 1, 0);
  std::vector<int32_t> trow(nnz), tcol(nnz);
  std::vector<double> tval(nnz);
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < T; ++t) {
    int64_t* c = cnt.data() + (size_t)t * nb;
    for (int32_t k = A.colptr[cbeg[t]]; k < A.colptr[cbeg[t + 1]]; ++k) c[A.rowval[k] >> kShift]++;
  }
  {
    int64_t run = 0;
    for (int64_t b = 0; b < nb; ++b) {
      bstart[b] = run;
      for (int tt = 0; tt < T; ++tt) {
        const int64_t v = cnt[(size_t)tt * nb + b];
        cnt[(size_t)tt * nb + b] = run;
        run += v;
      }
    }
    bstart[nb] = run;
  }
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < T; ++t) {
    int64_t* c = cnt.data() + (size_t)t * nb;
    for (int64_t j = cbeg[t]; j < cbeg[t + 1]; ++j)
      for (int32_t k = A.colptr[j]; k < A.colptr[j + 1]; ++k) {
        const int32_t r = A.rowval[k];
        const int64_t o = c[r >> kShift]++;
        trow[o] = r;
        tcol[o] = (int32_t)j;
        tval[o] = A.nzval[k];
      }
  }
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t r0 = b << kShift, r1 = std::min<int64_t>(A.m, r0 + ((int64_t)1 << kShift));
    int32_t next[(1 << kShift) + 1];
    for (int64_t r = 0; r <= r1 - r0; ++r) next[r] = 0;
    for (int64_t k = bstart[b]; k < bstart[b + 1]; ++k) next[trow[k] - r0 + 1]++;
    int64_t run = bstart[b];
    for (int64_t r = 0; r < r1 - r0; ++r) {  // exclusive prefix -> output start of each row
      const int32_t v = next[r + 1];
      Tm->colptr[r0 + r] = (int32_t)run;
      next[r] = (int32_t)run;
      run += v;
    }
    for (int64_t k = bstart[b]; k < bstart[b + 1]; ++k) {
      const int32_t p = next[trow[k] - r0]++;
      Tm->rowval[p] = tcol[k];
      Tm->nzval[p] = tval[k];
    }
  }
  Tm->colptr[A.m] = (int32_t)nnz;
  return Tm;
}

// X*Y for CSC operands (SparseArrays spmatmul semantics): column j of the
// product accumulates, for k ascending over Y[:,j] and i ascending over X[:,k],
// acc[i] += X[i,k]*Y[k,j].  Structural zeros are kept; rows come out sorted.
MatP spgemm(const Mat& X, const Mat& Y) {
  if (X.n != Y.m) throw std::runtime_error("spgemm: dimension mismatch");
  const int64_t m = X.m, n = Y.n;
  MatP C = make(m, n);
  std::vector<int32_t> cnt(n, 0);
  // pass 1: structural count per column
#pragma omp parallel
  {
    std::vector<int32_t> mark(m, -1);
#pragma omp for schedule(dynamic, 1024)
    for (int64_t j = 0; j < n; ++j) {
      int32_t c = 0;
      for (int32_t p = Y.colptr[j]; p < Y.colptr[j + 1]; ++p) {
        int32_t k = Y.rowval[p];
        for (int32_t q = X.colptr[k]; q < X.colptr[k + 1]; ++q) {
          int32_t i = X.rowval[q];
          if (mark[i] != (int32_t)j) {
            mark[i] = (int32_t)j;
            ++c;
          }
        }
      }
      cnt[j] = c;
    }
  }
  int64_t total = 0;
  for (int64_t j = 0; j < n; ++j) {
    C->colptr[j] = (int32_t)total;
    total += cnt[j];
    check_nnz(total);
  }
  C->colptr[n] = (int32_t)total;
  C->rowval.resize(total);
  C->nzval.resize(total);
  // pass 2: numeric
#pragma omp parallel
  {
    std::vector<int32_t> mark(m, -1);
    std::vector<double> acc(m, 0.0);
#pragma omp for schedule(dynamic, 1024)
    for (int64_t j = 0; j < n; ++j) {
      int32_t base = C->colptr[j], c = 0;
      for (int32_t p = Y.colptr[j]; p < Y.colptr[j + 1]; ++p) {
        int32_t k = Y.rowval[p];
        double ykj = Y.nzval[p];
        for (int32_t q = X.colptr[k]; q < X.colptr[k + 1]; ++q) {
          int32_t i = X.rowval[q];
          if (mark[i] != (int32_t)j) {
            mark[i] = (int32_t)j;
            C->rowval[base + c++] = i;
            acc[i] = X.nzval[q] * ykj;
          } else {
            acc[i] += X.nzval[q] * ykj;
          }
        }
      }
      std::sort(C->rowval.begin() + base, C->rowval.begin() + base + c);
      for (int32_t t = 0; t < c; ++t) C->nzval[base + t] = acc[C->rowval[base + t]];
    }
  }
  return C;
}

// ---- gallery.jl:1-63 ---------------------------------------------------
// poisson((n1,..,nN)): 2N on the diagonal, -1 per axis neighbour, Dirichlet
// truncation, linear index first-axis-fastest (LinearIndices, gallery.jl:14).
MatP poisson(int ndim, const int64_t* dims) {
  if (ndim < 1 || ndim > 8) throw std::runtime_error("poisson: bad ndim");
  int64_t n = 1;
  std::vector<int64_t> stride(ndim);
  for (int d = 0; d < ndim; ++d) {
    if (dims[d] < 1) throw std::runtime_error("poisson: bad dims");
    stride[d] = n;
    n *= dims[d];
  }
  check_nnz(n * (2 * ndim + 1));
  MatP A = make(n, n);
  // count
  std::vector<int64_t> idx(ndim, 0);
  int64_t nnz = 0;
  // column j: neighbours sorted ascending = [-stride[N-1],...,-stride[0], 0,
  // +stride[0],...,+stride[N-1]] where in-bounds.
  A->rowval.reserve(n * (2 * ndim + 1));
  A->nzval.reserve(n * (2 * ndim + 1));
  for (int64_t j = 0; j < n; ++j) {
    for (int d = ndim - 1; d >= 0; --d)
      if (idx[d] > 0) {
        A->rowval.push_back((int32_t)(j - stride[d]));
        A->nzval.push_back(-1.0);
      }
    A->rowval.push_back((int32_t)j);
    A->nzval.push_back(2.0 * ndim);
    for (int d = 0; d < ndim; ++d)
      if (idx[d] + 1 < dims[d]) {
        A->rowval.push_back((int32_t)(j + stride[d]));
        A->nzval.push_back(-1.0);
      }
    nnz = (int64_t)A->rowval.size();
    A->colptr[j + 1] = (int32_t)nnz;
    for (int d = 0; d < ndim; ++d) {  // increment first-axis-fastest
      if (++idx[d] < dims[d]) break;
      idx[d] = 0;
    }
  }
  // a size-1 axis contributes nothing but the 2 on the diagonal (stencil_grid
  // bounds check), already handled.
  return A;
}

// ---- strength.jl:7-70 : Classical(theta)(At) ---------------------------
void scale_cols_by_largest_entry(Mat& A) {  // strength.jl:61-70, find_max :50-58
  const int64_t lim = std::min(A.m, A.n);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < lim; ++i) {  // n = size(A,1)
    double mx = 0.0;
    for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) mx = std::max(mx, A.nzval[j]);
    for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) A.nzval[j] /= mx;
  }
}

void classical_strength(const Mat& At, double theta, MatP& S, MatP& T) {
  T.reset(new Mat(At));
  const int64_t n = At.n;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    double mx = 0.0;  // find_max_off_diag, strength.jl:39-48
    for (int32_t j = T->colptr[i]; j < T->colptr[i + 1]; ++j)
      if (T->rowval[j] != i) mx = std::max(mx, std::fabs(T->nzval[j]));
    double thr = theta * mx;
    for (int32_t j = T->colptr[i]; j < T->colptr[i + 1]; ++j) {
      if (T->rowval[j] != i) {
        double v = T->nzval[j];
        T->nzval[j] = (std::fabs(v) >= thr) ? std::fabs(v) : 0.0;
      }
    }
  }
  dropzeros(*T);
  scale_cols_by_largest_entry(*T);
  S = transpose(*T);
}

// ---- strength.jl:77-122 : SymmetricStrength(theta)(A, bsr_flag) --------
MatP symmetric_strength(const Mat& A, double theta, bool bsr_flag) {
  MatP S(new Mat(A));
  if (bsr_flag && theta == 0.0) {  // strength.jl:81-84: pattern of A, all ones
    std::fill(S->nzval.begin(), S->nzval.end(), 1.0);
    return S;
  }
  const int64_t n = A.m;
  const int64_t lim = std::min(n, A.n);
  std::vector<double> diags(n, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < lim; ++i) {
    double d = 0.0;
    for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j)
      if (A.rowval[j] == i) d += A.nzval[j];
    diags[i] = std::fabs(d);
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < lim; ++i) {
    double eps_Aii = theta * theta * diags[i];
    for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) {
      int32_t row = A.rowval[j];
      double v = A.nzval[j];
      if (row != i && v * v < eps_Aii * diags[row]) S->nzval[j] = 0.0;
    }
  }
  dropzeros(*S);
  const int64_t snz = S->nnz();
  double* sv = S->nzval.data();
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < snz; ++k) sv[k] = std::fabs(sv[k]);
  scale_cols_by_largest_entry(*S);
  return S;
}

// ---- splitting.jl:8-159 : RS() -----------------------------------------
void remove_diag(Mat& a) {  // splitting.jl:8-18
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < a.n; ++i)
    for (int32_t j = a.colptr[i]; j < a.colptr[i + 1]; ++j)
      if (a.rowval[j] == i) a.nzval[j] = 0.0;
  dropzeros(a);
}

// RS_CF_splitting(S, T = copy(S')).  Index bookkeeping kept 1-based (arrays
// sized n+2) so the bucket arithmetic mirrors splitting.jl:25-159 literally;
// tie-breaking depends on it (ref_split_test.txt / thing.jl goldens).
void rs_cf_splitting_raw(int64_t n, const int32_t* Sp, const int32_t* Sj, const int32_t* Tp, const int32_t* Tj,
                         int32_t* splitting_out);
void rs_cf_splitting(const Mat& S, const Mat& T, int32_t* splitting_out) {
  rs_cf_splitting_raw(S.m, S.colptr.data(), S.rowval.data(), T.colptr.data(), T.rowval.data(), splitting_out);
}
// the sweep itself works on the two PATTERNS only (column pointers + row indices of S and of T = S')
//
// The sweep is one sequential chain of dependent random accesses (the bucket bookkeeping of splitting.jl:25-159 has
// to be followed step by step for the integer goldens), so its speed is set by cache misses per visited node: the
// per-node fields (lambda, position in the bucket array, C/F/U state) live in ONE 8-byte record, and the records of a
// node's neighbours are prefetched before the loop that walks them.
namespace {
struct SplitNode {
  int32_t lambda;
  uint32_t pos_state;  // position in index_to_node (30 bits) | state (2 bits: 0 = U, 1 = C, 2 = F)
};
constexpr uint32_t kStU = 0u, kStC = 1u, kStF = 2u;
inline uint32_t st_of(const SplitNode& v) { return v.pos_state >> 30; }
inline int64_t pos_of(const SplitNode& v) { return (int64_t)(v.pos_state & 0x3fffffffu); }
inline void set_state(SplitNode& v, uint32_t st) { v.pos_state = (v.pos_state & 0x3fffffffu) | (st << 30); }
inline void set_pos(SplitNode& v, int64_t p) { v.pos_state = (v.pos_state & 0xc0000000u) | (uint32_t)p; }

// The sweep is a chain of dependent RANDOM accesses over hundreds of megabytes (node records, the bucket array, the
// strength patterns): with 4 KiB pages nearly every access also misses the TLB.  Its working arrays live in anonymous
// mappings advised MADV_HUGEPAGE before their first touch (transparent huge pages are in `madvise` mode on the target
// hosts): 2 MiB pages, a page walk per 512 x as many bytes.
template <class T>
struct HugeBuf {
  T* p = nullptr;
  size_t bytes = 0;
  explicit HugeBuf(size_t count) {
    bytes = ((count * sizeof(T) + (size_t(2) << 20) - 1) >> 21) << 21;
    if (bytes == 0) bytes = size_t(2) << 20;
    void* q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (q == MAP_FAILED) throw std::bad_alloc();
    if (!std::getenv("AMGS_NO_HUGEPAGES")) madvise(q, bytes, MADV_HUGEPAGE);   // advice only: failure is harmless
    p = (T*)q;                                                               // (anonymous mappings start zeroed)
  }
  ~HugeBuf() { if (p) munmap(p, bytes); }
  HugeBuf(const HugeBuf&) = delete;
  HugeBuf& operator=(const HugeBuf&) = delete;
  T* data() { return p; }
  T& operator[](size_t i) { return p[i]; }
};

void rs_cf_splitting_packed(int64_t n, const int32_t* Sp, const int32_t* Sj, const int32_t* Tp, const int32_t* Tj,
                            int32_t* splitting_out) {
  HugeBuf<SplitNode> node(n + 2);
  HugeBuf<int32_t> interval_ptr(n + 3), interval_count(n + 3), index_to_node(n + 2);
  // (private huge-page copies of the patterns too: measured SLOWER — faulting in that many fresh 2 MiB pages costs more
  //  than the page walks it saves; AMGS_PATTERN_COPY=1 keeps the experiment available)
  std::unique_ptr<HugeBuf<int32_t>> cSp, cSj, cTp, cTj;
  if (n >= (int64_t(1) << 20) && !std::getenv("AMGS_NO_HUGEPAGES") && std::getenv("AMGS_PATTERN_COPY")) {
    auto dup = [](const int32_t* src, size_t cnt, std::unique_ptr<HugeBuf<int32_t>>& dst) {
      dst.reset(new HugeBuf<int32_t>(cnt));
      std::memcpy(dst->data(), src, cnt * sizeof(int32_t));
      return (const int32_t*)dst->data();
    };
    const size_t ns = (size_t)Sp[n], nt = (size_t)Tp[n];
    Sp = dup(Sp, (size_t)n + 1, cSp); Sj = dup(Sj, ns, cSj);
    Tp = dup(Tp, (size_t)n + 1, cTp); Tj = dup(Tj, nt, cTj);
  }
  for (int64_t i = 1; i <= n; ++i) {
    node[i].lambda = Sp[i] - Sp[i - 1];
    interval_count[node[i].lambda + 1] += 1;
  }
  {
    int64_t s = 0;
    for (int64_t k = 1; k <= n; ++k) {
      s += interval_count[k];
      interval_ptr[k + 1] = (int32_t)s;
    }
  }
  std::memset(interval_count.data(), 0, sizeof(int32_t) * (size_t)(n + 3));
  for (int64_t i = 1; i <= n; ++i) {
    int64_t li = node[i].lambda + 1;
    interval_count[li] += 1;
    int64_t index = interval_ptr[li] + interval_count[li];
    index_to_node[index] = (int32_t)i;
    set_pos(node[i], index);
    if (node[i].lambda == 0) set_state(node[i], kStF);
  }
  SplitNode* nd = node.data();
  int32_t* i2n = index_to_node.data();
  const bool dry = !std::getenv("AMGS_SPLIT_NODRY");
  for (int64_t top_index = n; top_index >= 1; --top_index) {
    // look-ahead on the nodes that are next in line (their place may still change: prefetches only)
    if (top_index > 8) {
      __builtin_prefetch(&nd[i2n[top_index - 8]], 1, 1);
      __builtin_prefetch(&Sp[i2n[top_index - 8] - 1], 0, 1);
      const int64_t i4 = i2n[top_index - 4];
      __builtin_prefetch(&Sj[Sp[i4 - 1]], 0, 1);
      __builtin_prefetch(&Tj[Tp[i4 - 1]], 0, 1);
      const int64_t i2 = i2n[top_index - 2];
      if (st_of(nd[i2]) != kStF)
        for (int32_t j = Sp[i2 - 1]; j < Sp[i2]; ++j) {
          __builtin_prefetch(&nd[Sj[j] + 1], 1, 1);
          __builtin_prefetch(&Tp[Sj[j]], 0, 1);
        }
    }
    int64_t i = i2n[top_index];
    int64_t li = nd[i].lambda + 1;
    interval_count[li] -= 1;
    if (st_of(nd[i]) == kStF) continue;
    set_state(nd[i], kStC);
    const int32_t s0 = Sp[i - 1], s1 = Sp[i];
    for (int32_t j = s0; j < s1; ++j) {
      __builtin_prefetch(&nd[Sj[j] + 1], 1, 1);
      __builtin_prefetch(&Tp[Sj[j]], 0, 1);
    }
    for (int32_t j = Tp[i - 1]; j < Tp[i]; ++j) __builtin_prefetch(&nd[Tj[j] + 1], 1, 1);
    if (dry) {
      // dry passes over what the real pass below will touch, one level of indirection at a time, so that the misses
      // of a whole neighbourhood are in flight together instead of one after the other
      for (int32_t j = s0; j < s1; ++j) {
        const int64_t row = (int64_t)Sj[j] + 1;
        if (st_of(nd[row]) == kStU) __builtin_prefetch(&Tj[Tp[row - 1]], 0, 1);
      }
      for (int32_t j = s0; j < s1; ++j) {
        const int64_t row = (int64_t)Sj[j] + 1;
        if (st_of(nd[row]) == kStU)
          for (int32_t k = Tp[row - 1]; k < Tp[row]; ++k) __builtin_prefetch(&nd[Tj[k] + 1], 1, 1);
      }
      for (int32_t j = s0; j < s1; ++j) {
        const int64_t row = (int64_t)Sj[j] + 1;
        if (st_of(nd[row]) == kStU)
          for (int32_t k = Tp[row - 1]; k < Tp[row]; ++k) {
            const SplitNode& nk = nd[(int64_t)Tj[k] + 1];
            if (st_of(nk) == kStU) {
              __builtin_prefetch(&i2n[pos_of(nk)], 1, 1);
              const int64_t lk = nk.lambda + 1;
              __builtin_prefetch(&nd[i2n[interval_ptr[lk] + interval_count[lk]]], 1, 1);
            }
          }
      }
    }
    for (int32_t j = s0; j < s1; ++j) {
      int64_t row = (int64_t)Sj[j] + 1;
      if (st_of(nd[row]) == kStU) {
        set_state(nd[row], kStF);
        const int32_t t0 = Tp[row - 1], t1 = Tp[row];
        for (int32_t k = t0; k < t1; ++k) __builtin_prefetch(&nd[Tj[k] + 1], 1, 1);
        for (int32_t k = t0; k < t1; ++k) {
          int64_t rowk = (int64_t)Tj[k] + 1;
          SplitNode& nk = nd[rowk];
          if (st_of(nk) == kStU) {
            if (nk.lambda >= n - 1) continue;
            int64_t lk = nk.lambda + 1;
            int64_t old_pos = pos_of(nk);
            int64_t new_pos = interval_ptr[lk] + interval_count[lk];
            int64_t swap_node = i2n[new_pos];
            i2n[old_pos] = (int32_t)swap_node;
            i2n[new_pos] = (int32_t)rowk;
            set_pos(nk, new_pos);
            set_pos(nd[swap_node], old_pos);
            nk.lambda += 1;
            interval_count[lk] -= 1;
            interval_count[lk + 1] += 1;
            interval_ptr[lk + 1] = (int32_t)(new_pos - 1);
          }
        }
      }
    }
    for (int32_t j = Tp[i - 1]; j < Tp[i]; ++j) {
      int64_t row = (int64_t)Tj[j] + 1;
      SplitNode& nr = nd[row];
      if (st_of(nr) == kStU) {
        if (nr.lambda == 0) continue;
        int64_t lj = nr.lambda + 1;
        int64_t old_pos = pos_of(nr);
        int64_t new_pos = interval_ptr[lj] + 1;
        int64_t swap_node = i2n[new_pos];
        i2n[old_pos] = (int32_t)swap_node;
        i2n[new_pos] = (int32_t)row;
        set_pos(nr, new_pos);
        set_pos(nd[swap_node], old_pos);
        nr.lambda -= 1;
        interval_count[lj] -= 1;
        interval_count[lj - 1] += 1;
        interval_ptr[lj] += 1;
      }
    }
  }
  for (int64_t i = 1; i <= n; ++i) {
    uint32_t st = st_of(nd[i]);
    splitting_out[i - 1] = st == kStC ? C_NODE : (st == kStF ? F_NODE : U_NODE);
  }
}
}  // namespace

void rs_cf_splitting_raw(int64_t n, const int32_t* Sp, const int32_t* Sj, const int32_t* Tp, const int32_t* Tj,
                         int32_t* splitting_out) {
  if (n + 2 < (int64_t(1) << 30) && !std::getenv("AMGS_SPLIT_PLAIN")) {
    rs_cf_splitting_packed(n, Sp, Sj, Tp, Tj, splitting_out);
    return;
  }
  // plain-array form of the same sweep (n >= 2^30, or AMGS_SPLIT_PLAIN for the A/B test)
  std::vector<int32_t> lambda(n + 2, 0), interval_ptr(n + 3, 0), interval_count(n + 3, 0);
  std::vector<int32_t> index_to_node(n + 2, 0), node_to_index(n + 2, 0);
  std::vector<int8_t> splitting(n + 2, U_NODE);

  for (int64_t i = 1; i <= n; ++i) {
    lambda[i] = Sp[i] - Sp[i - 1];
    interval_count[lambda[i] + 1] += 1;
  }
  // accumulate!(+, interval_ptr[2:end], interval_count[1:end-1])
  {
    int64_t s = 0;
    for (int64_t k = 1; k <= n; ++k) {
      s += interval_count[k];
      interval_ptr[k + 1] = s;
    }
  }
  std::fill(interval_count.begin(), interval_count.end(), 0);
  for (int64_t i = 1; i <= n; ++i) {
    int64_t li = lambda[i] + 1;
    interval_count[li] += 1;
    int64_t index = interval_ptr[li] + interval_count[li];
    index_to_node[index] = i;
    node_to_index[i] = index;
  }
  for (int64_t i = 1; i <= n; ++i)
    if (lambda[i] == 0) splitting[i] = F_NODE;

  for (int64_t top_index = n; top_index >= 1; --top_index) {
    int64_t i = index_to_node[top_index];
    int64_t li = lambda[i] + 1;
    interval_count[li] -= 1;
    if (splitting[i] == F_NODE) continue;
    // splitting[i] == U_NODE here (splitting.jl:95)
    splitting[i] = C_NODE;
    for (int32_t j = Sp[i - 1]; j < Sp[i]; ++j) {
      int64_t row = (int64_t)Sj[j] + 1;
      if (splitting[row] == U_NODE) {
        splitting[row] = F_NODE;
        for (int32_t k = Tp[row - 1]; k < Tp[row]; ++k) {
          int64_t rowk = (int64_t)Tj[k] + 1;
          if (splitting[rowk] == U_NODE) {
            if (lambda[rowk] >= n - 1) continue;
            int64_t lk = lambda[rowk] + 1;
            int64_t old_pos = node_to_index[rowk];
            int64_t new_pos = interval_ptr[lk] + interval_count[lk];
            int64_t swap_node = index_to_node[new_pos];
            index_to_node[old_pos] = swap_node;
            index_to_node[new_pos] = rowk;
            node_to_index[rowk] = new_pos;
            node_to_index[swap_node] = old_pos;
            lambda[rowk] += 1;
            interval_count[lk] -= 1;
            interval_count[lk + 1] += 1;
            interval_ptr[lk + 1] = new_pos - 1;
          }
        }
      }
    }
    for (int32_t j = Tp[i - 1]; j < Tp[i]; ++j) {
      int64_t row = (int64_t)Tj[j] + 1;
      if (splitting[row] == U_NODE) {
        if (lambda[row] == 0) continue;
        int64_t lj = lambda[row] + 1;
        int64_t old_pos = node_to_index[row];
        int64_t new_pos = interval_ptr[lj] + 1;
        int64_t swap_node = index_to_node[new_pos];
        index_to_node[old_pos] = swap_node;
        index_to_node[new_pos] = row;
        node_to_index[row] = new_pos;
        node_to_index[swap_node] = old_pos;
        lambda[row] -= 1;
        interval_count[lj] -= 1;
        interval_count[lj - 1] += 1;
        interval_ptr[lj] += 1;
      }
    }
  }
  for (int64_t i = 1; i <= n; ++i) splitting_out[i - 1] = splitting[i];
}

void rs_splitting(Mat& S, int32_t* splitting) {  // splitting.jl:20-23
  remove_diag(S);
  MatP T = transpose(S);
  rs_cf_splitting(S, *T, splitting);
}

// ---- classical.jl:57-189 : direct_interpolation ------------------------
// Returns R (n_c x n, one column per fine node); P = R'.
MatP direct_interpolation(const Mat& At, const Mat& Tin, const int32_t* splitting) {
  const int64_t n = At.m;
  // T .= At .* pattern(T)  (classical.jl:58-60): At's values on T's pattern,
  // numerically-zero products are not stored.
  Mat T;
  T.m = Tin.m;
  T.n = Tin.n;
  T.colptr.assign(T.n + 1, 0);
  auto masked = [&](int64_t i, int32_t* rv, double* nz) -> int32_t {  // entries of column i; rv == nullptr: count only
    int32_t a = At.colptr[i], ae = At.colptr[i + 1], c = 0;
    for (int32_t t = Tin.colptr[i]; t < Tin.colptr[i + 1]; ++t) {
      int32_t row = Tin.rowval[t];
      while (a < ae && At.rowval[a] < row) ++a;
      if (a < ae && At.rowval[a] == row) {
        double v = At.nzval[a] * 1.0;
        if (v != 0.0) {
          if (rv) {
            rv[c] = row;
            nz[c] = v;
          }
          ++c;
        }
      }
    }
    return c;
  };
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < Tin.n; ++i) T.colptr[i + 1] = masked(i, nullptr, nullptr);
  for (int64_t i = 0; i < Tin.n; ++i) T.colptr[i + 1] += T.colptr[i];
  T.rowval.resize(T.colptr[T.n]);
  T.nzval.resize(T.colptr[T.n]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < Tin.n; ++i) masked(i, T.rowval.data() + T.colptr[i], T.nzval.data() + T.colptr[i]);

  // pass 1 (classical.jl:71-89)
  std::vector<int32_t> Bp(n + 1, 0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    int32_t c = 0;
    if (splitting[i] == C_NODE) {
      c = 1;
    } else {
      for (int32_t j = T.colptr[i]; j < T.colptr[i + 1]; ++j)
        if (splitting[T.rowval[j]] == C_NODE) c += 1;
    }
    Bp[i + 1] = c;
  }
  int64_t nnzc = 0;
  for (int64_t i = 0; i < n; ++i) {
    nnzc += Bp[i + 1];
    check_nnz(nnzc);
    Bp[i + 1] = (int32_t)nnzc;
  }
  // pass 2 (classical.jl:92-189)
  std::vector<double> Bx(nnzc, 0.0);
  std::vector<int32_t> Bj(nnzc, 0);
  const double eps = std::numeric_limits<double>::epsilon();
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    if (splitting[i] == C_NODE) {
      Bj[Bp[i]] = (int32_t)i;
      Bx[Bp[i]] = 1.0;
      continue;
    }
    double sum_strong_pos = 0.0, sum_strong_neg = 0.0;
    for (int32_t j = T.colptr[i]; j < T.colptr[i + 1]; ++j) {
      int32_t row = T.rowval[j];
      double sval = T.nzval[j];
      if (splitting[row] == C_NODE) {
        if (sval < 0)
          sum_strong_neg += sval;
        else
          sum_strong_pos += sval;
      }
    }
    double sum_all_pos = 0.0, sum_all_neg = 0.0, diag = 0.0;
    for (int32_t j = At.colptr[i]; j < At.colptr[i + 1]; ++j) {
      int32_t row = At.rowval[j];
      double aval = At.nzval[j];
      if (row == i) {
        diag += aval;
      } else if (aval < 0) {
        sum_all_neg += aval;
      } else {
        sum_all_pos += aval;
      }
    }
    double alpha, beta;
    if (sum_strong_pos == 0) {
      beta = 0.0;
      if (diag >= 0) diag += sum_all_pos;
    } else {
      beta = sum_all_pos / sum_strong_pos;
    }
    if (sum_strong_neg == 0) {
      alpha = 0.0;
      if (diag < 0) diag += sum_all_neg;
    } else {
      alpha = sum_all_neg / sum_strong_neg;
    }
    double neg_coeff, pos_coeff;
    if (std::fabs(diag) <= eps) {  // isapprox(real(diag), 0, atol=eps)
      neg_coeff = 0.0;
      pos_coeff = 0.0;
    } else {
      neg_coeff = alpha / diag;
      pos_coeff = beta / diag;
    }
    int32_t nz = Bp[i];
    for (int32_t j = T.colptr[i]; j < T.colptr[i + 1]; ++j) {
      int32_t row = T.rowval[j];
      double sval = T.nzval[j];
      if (splitting[row] == C_NODE) {
        Bj[nz] = row;
        Bx[nz] = (sval < 0) ? std::fabs(neg_coeff * sval) : std::fabs(pos_coeff * sval);
        ++nz;
      }
    }
  }
  // coarse index map = exclusive prefix sum of splitting (classical.jl:180-186)
  std::vector<int32_t> map(n, 0);
  int32_t sum = 0;
  for (int64_t i = 0; i < n; ++i) {
    map[i] = sum;
    sum += splitting[i];
  }
  int32_t nc = 0;
#pragma omp parallel for schedule(static) reduction(max : nc)
  for (int64_t k = 0; k < nnzc; ++k) {
    Bj[k] = map[Bj[k]];
    nc = std::max(nc, Bj[k] + 1);  // isempty(Pj) ? 0 : maximum(Pj)
  }
  MatP R = make(nc, n);
  R->colptr = std::move(Bp);
  R->rowval = std::move(Bj);
  R->nzval = std::move(Bx);
  return R;
}

// ---- aggregate.jl:12-134 : StandardAggregation -------------------------
MatP standard_aggregation(const Mat& S) {
  const int64_t n = S.m;
  std::vector<int64_t> x(n, 0);
  int64_t next_aggregate = 1;
  // Pass 1
  for (int64_t i = 0; i < n; ++i) {
    if (x[i] != 0) continue;
    bool has_agg_neighbors = false, has_neighbors = false;
    for (int32_t j = S.colptr[i]; j < S.colptr[i + 1]; ++j) {
      int32_t row = S.rowval[j];
      if (row != i) {
        has_neighbors = true;
        if (x[row] != 0) {
          has_agg_neighbors = true;
          break;
        }
      }
    }
    if (!has_neighbors) {
      x[i] = -n;
    } else if (!has_agg_neighbors) {
      x[i] = next_aggregate;
      for (int32_t j = S.colptr[i]; j < S.colptr[i + 1]; ++j) {
        int32_t row = S.rowval[j];
        if (row != i) x[row] = next_aggregate;
      }
      next_aggregate += 1;
    }
  }
  // Pass 2
  for (int64_t i = 0; i < n; ++i) {
    if (x[i] != 0) continue;
    double s_best = 0.0;
    int64_t x_best = 0;
    for (int32_t j = S.colptr[i]; j < S.colptr[i + 1]; ++j) {
      int64_t x_row = x[S.rowval[j]];
      double s_candidate = S.nzval[j];
      if (x_row > 0 && s_candidate > s_best) {
        s_best = s_candidate;
        x_best = x_row;
      }
    }
    if (x_best > 0) x[i] = -x_best;
  }
  std::vector<uint8_t> unagg(n);
  for (int64_t i = 0; i < n; ++i) unagg[i] = (x[i] == 0);
  next_aggregate -= 1;
  for (int64_t i = 0; i < n; ++i) {
    int64_t xi = x[i];
    if (xi > 0)
      x[i] = xi - 1;
    else if (xi == -n)
      x[i] = -1;
    else if (xi < 0)
      x[i] = -xi - 1;
  }
  // Pass 3
  for (int64_t i = 0; i < n; ++i) {
    if (!unagg[i]) continue;
    x[i] = next_aggregate;
    for (int32_t j = S.colptr[i]; j < S.colptr[i + 1]; ++j) {
      int32_t row = S.rowval[j];
      if (unagg[row]) {
        x[row] = next_aggregate;
        unagg[row] = 0;
      }
    }
    unagg[i] = 0;
    next_aggregate += 1;
  }
  const int64_t N = next_aggregate;
  MatP Agg = make(N, n);  // N_agg x n, one column per fine node
  for (int64_t i = 0; i < n; ++i) {
    if (x[i] != -1) {  // isolated nodes get an empty column (aggregate.jl:120-126)
      Agg->rowval.push_back((int32_t)x[i]);
      Agg->nzval.push_back(1.0);
    }
    Agg->colptr[i + 1] = (int32_t)Agg->rowval.size();
  }
  return Agg;
}

// ---- aggregation.jl:161-193 : fit_candidates, B::Vector ----------------
MatP fit_candidates_vector(const Mat& AggOp, const double* B, double tol,
                           std::vector<double>& Bc) {
  MatP A = transpose(AggOp);  // n_fine x n_coarse
  const int64_t n_col = A->n;
  Bc.assign(n_col, 0.0);
  for (int64_t i = 0; i < n_col; ++i)
    for (int32_t j = A->colptr[i]; j < A->colptr[i + 1]; ++j) A->nzval[j] = B[A->rowval[j]];
  for (int64_t i = 0; i < n_col; ++i) {
    double s = 0.0;  // norm_col (aggregation.jl:232-240)
    for (int32_t j = A->colptr[i]; j < A->colptr[i + 1]; ++j) {
      double v = B[A->rowval[j]];
      s += v * v;
    }
    double norm_i = std::sqrt(s);
    double threshold_i = tol * norm_i;
    double scale;
    if (norm_i > threshold_i) {
      scale = 1.0 / norm_i;
      Bc[i] = norm_i;
    } else {
      scale = 0.0;
      Bc[i] = 0.0;
    }
    for (int32_t j = A->colptr[i]; j < A->colptr[i + 1]; ++j) A->nzval[j] *= scale;
  }
  return A;
}

// Dense Householder QR of an r x m block (column-major, ld = r), LAPACK
// dgeqr2/dlarfg conventions (beta = -sign(alpha)*norm).  On return `a` holds R
// in its upper triangle; Q (r x min(r,m), thin) is written to q.
void householder_qr(std::vector<double>& a, int r, int m, std::vector<double>& q) {
  const int k = std::min(r, m);
  std::vector<double> tau(k, 0.0);
  for (int j = 0; j < k; ++j) {
    double alpha = a[j + (size_t)j * r];
    double xnorm = 0.0;
    for (int i = j + 1; i < r; ++i) xnorm += a[i + (size_t)j * r] * a[i + (size_t)j * r];
    xnorm = std::sqrt(xnorm);
    if (xnorm == 0.0) {
      tau[j] = 0.0;
    } else {
      double beta = -std::copysign(std::hypot(alpha, xnorm), alpha);
      tau[j] = (beta - alpha) / beta;
      double scal = 1.0 / (alpha - beta);
      for (int i = j + 1; i < r; ++i) a[i + (size_t)j * r] *= scal;
      a[j + (size_t)j * r] = beta;
    }
    // apply H_j to trailing columns
    for (int c = j + 1; c < m; ++c) {
      double w = a[j + (size_t)c * r];
      for (int i = j + 1; i < r; ++i) w += a[i + (size_t)j * r] * a[i + (size_t)c * r];
      w *= tau[j];
      a[j + (size_t)c * r] -= w;
      for (int i = j + 1; i < r; ++i) a[i + (size_t)c * r] -= w * a[i + (size_t)j * r];
    }
  }
  // form thin Q = H_0 H_1 ... H_{k-1} [I_k; 0]
  q.assign((size_t)r * k, 0.0);
  for (int c = 0; c < k; ++c) q[c + (size_t)c * r] = 1.0;
  for (int j = k - 1; j >= 0; --j) {
    for (int c = 0; c < k; ++c) {
      double w = q[j + (size_t)c * r];
      for (int i = j + 1; i < r; ++i) w += a[i + (size_t)j * r] * q[i + (size_t)c * r];
      w *= tau[j];
      q[j + (size_t)c * r] -= w;
      for (int i = j + 1; i < r; ++i) q[i + (size_t)c * r] -= w * a[i + (size_t)j * r];
    }
  }
}

// ---- aggregation.jl:195-230 : fit_candidates, B::Matrix (QR per aggregate)
MatP fit_candidates_matrix(const Mat& AggOp, const double* B, int m, double tol,
                           std::vector<double>& Bc) {
  MatP A = transpose(AggOp);  // n_fine x n_agg
  const int64_t n_fine = A->m, n_agg = A->n;
  const int64_t n_coarse = (int64_t)m * n_agg;
  Bc.assign((size_t)n_coarse * m, 0.0);  // n_coarse x m column-major
  // collect triplets column by column; columns offset+local_j are produced in
  // increasing order and rows within an aggregate ascend, so output is CSC-sorted.
  MatP Qs = make(n_fine, n_coarse);
  std::vector<double> Mblk, Q;
  for (int64_t agg = 0; agg < n_agg; ++agg) {
    const int32_t* rows = A->rowval.data() + A->colptr[agg];
    const int nr = A->colptr[agg + 1] - A->colptr[agg];
    const int r = std::min(nr, m);
    const int64_t offset = agg * m;
    if (nr > 0) {
      Mblk.assign((size_t)nr * m, 0.0);
      for (int c = 0; c < m; ++c)
        for (int i = 0; i < nr; ++i) Mblk[i + (size_t)c * nr] = B[rows[i] + (size_t)c * n_fine];
      householder_qr(Mblk, nr, m, Q);
      for (int lj = 0; lj < r; ++lj) {
        for (int li = 0; li < nr; ++li) {
          double v = Q[li + (size_t)lj * nr];
          if (std::fabs(v) >= tol && v != 0.0) {  // store if >= tol; dropzeros! after
            Qs->rowval.push_back(rows[li]);
            Qs->nzval.push_back(v);
          }
        }
        Qs->colptr[offset + lj + 1] = (int32_t)Qs->rowval.size();
      }
      for (int c = 0; c < m; ++c)
        for (int i = 0; i < r; ++i)
          Bc[(offset + i) + (size_t)c * n_coarse] = (i <= c) ? Mblk[i + (size_t)c * nr] : 0.0;
    }
    for (int lj = r; lj < m; ++lj) Qs->colptr[offset + lj + 1] = (int32_t)Qs->rowval.size();
  }
  return Qs;
}

// ---- aggregation.jl:10-59 : JacobiProlongation(omega), LocalWeighting ---
// P = T - (omega * D^-1 A) * T,  D_ii = sum_j |a_ij|  (uses A, not S).
MatP sparse_sub(const Mat& X, const Mat& Y) {  // X - Y, exact zeros dropped
  MatP C = make(X.m, X.n);
  C->rowval.reserve(X.nnz() + Y.nnz());
  C->nzval.reserve(X.nnz() + Y.nnz());
  for (int64_t j = 0; j < X.n; ++j) {
    int32_t a = X.colptr[j], ae = X.colptr[j + 1];
    int32_t b = Y.colptr[j], be = Y.colptr[j + 1];
    while (a < ae || b < be) {
      int32_t row;
      double v;
      if (b >= be || (a < ae && X.rowval[a] < Y.rowval[b])) {
        row = X.rowval[a];
        v = X.nzval[a] - 0.0;
        ++a;
      } else if (a >= ae || Y.rowval[b] < X.rowval[a]) {
        row = Y.rowval[b];
        v = 0.0 - Y.nzval[b];
        ++b;
      } else {
        row = X.rowval[a];
        v = X.nzval[a] - Y.nzval[b];
        ++a;
        ++b;
      }
      if (v != 0.0) {
        C->rowval.push_back(row);
        C->nzval.push_back(v);
      }
    }
    C->colptr[j + 1] = (int32_t)C->rowval.size();
  }
  return C;
}

MatP jacobi_prolongation(const Mat& A, const Mat& T, double omega) {
  const int64_t n = A.m;
  std::vector<double> D(n, 0.0);
  for (int64_t i = 0; i < A.n; ++i)
    for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) D[A.rowval[j]] += std::fabs(A.nzval[j]);
  for (int64_t i = 0; i < n; ++i)
    if (D[i] != 0) D[i] = 1.0 / D[i];
  Mat DinvS(A);  // scale_rows (aggregation.jl:49-59) then rmul!(., omega)
  for (int64_t k = 0; k < DinvS.nnz(); ++k) {
    DinvS.nzval[k] *= D[DinvS.rowval[k]];
    DinvS.nzval[k] *= omega;
  }
  MatP W = spgemm(DinvS, T);
  return sparse_sub(T, *W);
}

// ---- smoother.jl:61-90 : gs!, Hermitian "fast" path (column i read as row i).
// Host-side copy used ONLY by improve_candidates during SA setup
// (aggregation.jl:75,135-136); the solve-phase sweeps run on the GPU.
void gs_sweep_host(const Mat& A, const double* b, double* x, bool forward) {
  const int64_t n = A.m;
  for (int64_t t = 0; t < n; ++t) {
    int64_t i = forward ? t : n - 1 - t;
    double rsum = 0.0, d = 0.0;
    for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) {
      int32_t row = A.rowval[j];
      double val = A.nzval[j];
      if (row == i)
        d = val;
      else
        rsum += val * x[row];
    }
    if (d != 0.0) x[i] = (b[i] - rsum) / d;
  }
}

// The same sweeps in parallel, EXACTLY: the lexicographic sweep is a sparse triangular solve, its result depends only on
// the dependency DAG.  With level[i] = 1 + max(level[j] : j adjacent to i, j < i) on the symmetrised pattern, the rows of
// one level read no row of their own level: a forward sweep visits the levels upward, a backward sweep downward, the rows
// of a level on all threads — every row runs the scalar loop of gs_sweep_host on the same operands in the same order,
// so the iterates are bitwise those of the sequential sweep (what the GPU smoothers do per launch, done here with
// OpenMP for improve_candidates: 8 sweeps over the level matrix, 1.3 s of the 3.8 s of smoothed_aggregation(256^3)).
struct GsLevels {
  std::vector<int32_t> lvl_ptr, rows;  // rows of level l: rows[lvl_ptr[l] .. lvl_ptr[l+1]), ascending inside a level
  int nlev() const { return (int)lvl_ptr.size() - 1; }
};
GsLevels gs_levels_host(const Mat& A) {
  const int64_t n = A.m;
  std::vector<int32_t> lev(n, 0);
  int32_t maxlev = -1;
  // First from the lower neighbours alone (one sequential read pass, no scattered writes) — complete when the pattern is
  // structurally symmetric, which a parallel pass then verifies: every upper neighbour must sit in a later level.
  for (int64_t i = 0; i < n; ++i) {
    int32_t li = 0;
    for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) {
      const int32_t c = A.rowval[j];
      if (c < i) li = std::max(li, lev[c] + 1);
    }
    lev[i] = li;
    maxlev = std::max(maxlev, li);
  }
  int64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (int64_t i = 0; i < n; ++i)
    for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) {
      const int32_t c = A.rowval[j];
      if (c > i && c < n && lev[c] <= lev[i]) ++bad;
    }
  if (bad) {  // not symmetric: the levels of the symmetrised pattern (as the GPU schedules compute them)
    std::fill(lev.begin(), lev.end(), 0);
    maxlev = -1;
    for (int64_t i = 0; i < n; ++i) {
      int32_t li = lev[i];
      for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) {
        const int32_t c = A.rowval[j];
        if (c < i) li = std::max(li, lev[c] + 1);
      }
      lev[i] = li;
      for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) {
        const int32_t c = A.rowval[j];
        if (c > i && c < n) lev[c] = std::max(lev[c], li + 1);
      }
      maxlev = std::max(maxlev, li);
    }
  }
  GsLevels G;
  G.lvl_ptr.assign(maxlev + 2, 0);
  for (int64_t i = 0; i < n; ++i) G.lvl_ptr[lev[i] + 1]++;
  for (int l = 0; l <= maxlev; ++l) G.lvl_ptr[l + 1] += G.lvl_ptr[l];
  G.rows.resize(n);
  std::vector<int32_t> next(G.lvl_ptr.begin(), G.lvl_ptr.end() - 1);
  for (int64_t i = 0; i < n; ++i) G.rows[next[lev[i]]++] = (int32_t)i;
  return G;
}
inline void gs_row_host(const Mat& A, const double* b, double* x, int64_t i) {
  double rsum = 0.0, d = 0.0;
  for (int32_t j = A.colptr[i]; j < A.colptr[i + 1]; ++j) {
    const int32_t row = A.rowval[j];
    const double val = A.nzval[j];
    if (row == i)
      d = val;
    else
      rsum += val * x[row];
  }
  if (d != 0.0) x[i] = (b[i] - rsum) / d;
}
// iters symmetric sweeps on each of the nB columns of B (b = 0): improve_candidates (aggregation.jl:135-136)
void improve_candidates_host(const Mat& A, double* B, int nB, int iters) {
  const int64_t n = A.m;
  std::vector<double> zero(n, 0.0);
  int threads = 1;
#ifdef _OPENMP
  threads = omp_get_max_threads();
#endif
  const bool par = threads > 1 && n >= 100000 && !std::getenv("AMGS_SEQUENTIAL_GS");
  GsLevels G;
  if (par) G = gs_levels_host(A);
  if (!par || (int64_t)G.nlev() * 256 > n) {   // few rows per level: the barriers would cost more than the rows
    for (int c = 0; c < nB; ++c)
      for (int it = 0; it < iters; ++it) {
        gs_sweep_host(A, zero.data(), B + (size_t)c * n, true);
        gs_sweep_host(A, zero.data(), B + (size_t)c * n, false);
      }
    return;
  }
  const int nlev = G.nlev();
  const double* b = zero.data();
#pragma omp parallel
  {
    for (int c = 0; c < nB; ++c) {
      double* x = B + (size_t)c * n;
      for (int it = 0; it < iters; ++it)
        for (int dir = 0; dir < 2; ++dir)
          for (int q = 0; q < nlev; ++q) {
            const int l = dir == 0 ? q : nlev - 1 - q;
            const int32_t r0 = G.lvl_ptr[l], r1 = G.lvl_ptr[l + 1];
#pragma omp for schedule(static)
            for (int32_t p = r0; p < r1; ++p) gs_row_host(A, b, x, G.rows[p]);
            // (implicit barrier: the next level reads what this one wrote)
          }
    }
  }
}

}  // namespace

struct amgs_hier {
  std::vector<MatP> A, P, R;
  MatP final_A;
};

namespace {

// OpenMP's thread count is a per-host-thread setting: a host thread other than the one that loaded the library (a
// worker of the Python setup pipeline) would start with the runtime's default — every CPU of the machine, whatever the
// container's quota.  Each entry point therefore adopts the library-wide cap the first time a host thread enters.
int g_threads_cap = 0;
inline void adopt_thread_cap() {
#ifdef _OPENMP
  static thread_local bool done = false;
  if (!done) {
    done = true;
    if (g_threads_cap > 0) omp_set_num_threads(g_threads_cap);
  }
#endif
}

struct Guard {
  template <class F>
  static auto ptr(F f) -> decltype(f()) {
    adopt_thread_cap();
    try {
      return f();
    } catch (const std::exception& e) {
      g_err = e.what();
      return nullptr;
    }
  }
  template <class F>
  static int rc(F f) {
    adopt_thread_cap();
    try {
      f();
      return 0;
    } catch (const std::exception& e) {
      g_err = e.what();
      return -1;
    }
  }
};

struct Timer {
  bool on = std::getenv("AMGS_TIMING") != nullptr;
  double t0 = now();
  static double now() {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
  }
  void lap(const char* label, int64_t n) {
    if (!on) return;
    const double t = now();
    std::fprintf(stderr, "[amgsetup] n=%lld %-14s %.3f s\n", (long long)n, label, t - t0);
    t0 = t;
  }
};

// classical.jl:6-55
amgs_hier* ruge_stuben(const Mat& A0, const amgs_options& o) {
  std::unique_ptr<amgs_hier> h(new amgs_hier);
  MatP A(new Mat(A0));
  if (A->m != A->n) throw std::runtime_error("ruge_stuben: matrix must be square");
  while ((int)h->A.size() + 1 < o.max_levels && A->m > o.max_coarse) {
    MatP Atown;
    const Mat* At = A.get();
    if (!o.hermitian) {
      Atown = transpose(*A);
      At = Atown.get();
    }
    Timer tm;  // AMGS_TIMING=1 prints the same labels the reference's @timeit_debug uses (classical.jl:25-48)
    MatP S, T;
    classical_strength(*At, o.theta, S, T);
    tm.lap("strength", A->m);
    std::vector<int32_t> splitting(A->m);
    rs_splitting(*S, splitting.data());
    tm.lap("splitting", A->m);
    MatP R = direct_interpolation(*At, *T, splitting.data());
    if (R->m == 0) break;  // size(P,2) == 0
    MatP P = transpose(*R);
    tm.lap("interpolation", A->m);
    MatP RA = spgemm(*R, *A);
    MatP RAP = spgemm(*RA, *P);
    tm.lap("RAP", A->m);
    h->A.push_back(std::move(A));
    h->P.push_back(std::move(P));
    h->R.push_back(std::move(R));
    A = std::move(RAP);
  }
  h->final_A = std::move(A);
  return h.release();
}

// aggregation.jl:66-157
amgs_hier* smoothed_aggregation(const Mat& A0, const double* B0, int nB, const amgs_options& o) {
  std::unique_ptr<amgs_hier> h(new amgs_hier);
  MatP A(new Mat(A0));
  if (A->m != A->n) throw std::runtime_error("smoothed_aggregation: matrix must be square");
  const int64_t n0 = A->m;
  bool vector_path = (B0 == nullptr) || (o.sa_B_is_vector && nB == 1);
  if (B0 == nullptr) nB = 1;
  std::vector<double> B((size_t)n0 * nB, 1.0);
  if (B0) std::copy(B0, B0 + (size_t)n0 * nB, B.begin());
  bool bsr_flag = false;
  while ((int)h->A.size() + 1 < o.max_levels && A->m > o.max_coarse) {
    const int64_t n = A->m;
    Timer tm;  // AMGS_TIMING=1: the labels of aggregation.jl:126-145
    MatP S;
    if (o.hermitian) {
      S = symmetric_strength(*A, o.theta, bsr_flag);
    } else {
      MatP At = transpose(*A);
      S = symmetric_strength(*At, o.theta, bsr_flag);
    }
    tm.lap("strength", n);
    MatP AggOp = standard_aggregation(*S);
    tm.lap("aggregation", n);
    if (AggOp->m == 0) break;
    // improve_candidates(A, B, 0): always the Hermitian fast path
    // (aggregation.jl:135-136, smoother.jl:34-38); symmetric sweep x iter.
    // gs! loops the columns of B inside each directional pass (smoother.jl:77);
    // columns are independent, so sweeping column-by-column is equivalent.
    improve_candidates_host(*A, B.data(), nB, o.sa_improve_iters);
    tm.lap("improve candidates", n);
    std::vector<double> Bc;
    MatP T = vector_path ? fit_candidates_vector(*AggOp, B.data(), 1e-10, Bc)
                         : fit_candidates_matrix(*AggOp, B.data(), nB, 1e-10, Bc);
    tm.lap("fit candidates", n);
    MatP P = jacobi_prolongation(*A, *T, o.sa_omega);
    tm.lap("smooth prolongator", n);
    if (P->n == 0) break;
    MatP R = transpose(*P);
    MatP RA = spgemm(*R, *A);
    MatP RAP = spgemm(*RA, *P);
    tm.lap("RAP", n);
    h->A.push_back(std::move(A));
    h->P.push_back(std::move(P));
    h->R.push_back(std::move(R));
    A = std::move(RAP);
    B = std::move(Bc);
    bsr_flag = true;
  }
  h->final_A = std::move(A);
  return h.release();
}

}  // namespace

extern "C" {

const char* amgs_last_error(void) { return g_err.c_str(); }

#ifdef _OPENMP
// Library load: unless OMP_NUM_THREADS says otherwise, do not run more OpenMP threads than the container's CPU
// quota (cgroup v2 cpu.max / v1 cfs quota) — a 256-CPU host with a 16-CPU quota ran the setup 3x slower with 256
// throttled threads than with 16 — and never more than 64 (the setup is memory-bound well before that).
__attribute__((constructor)) static void amgs_default_threads() {
  if (std::getenv("OMP_NUM_THREADS")) return;
  int cap = std::min(omp_get_max_threads(), 64);
  long quota = -1, period = -1;
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0};
    if (std::fscanf(f, "%63s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atol(q);
    std::fclose(f);
  } else {
    if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (std::fscanf(g, "%ld", &quota) != 1) quota = -1;
      std::fclose(g);
    }
    if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (std::fscanf(g, "%ld", &period) != 1) period = -1;
      std::fclose(g);
    }
  }
  if (quota > 0 && period > 0) cap = std::min<long>(cap, std::max<long>(1, (quota + period - 1) / period));
  g_threads_cap = cap;
  omp_set_num_threads(cap);
}
#endif

int amgs_set_threads(int nthreads) {
#ifdef _OPENMP
  adopt_thread_cap();
  if (nthreads > 0) {
    g_threads_cap = nthreads;
    omp_set_num_threads(nthreads);
  }
  return omp_get_max_threads();
#else
  (void)nthreads;
  return 1;
#endif
}

amgs_mat* amgs_mat_create(int64_t m, int64_t n, const int32_t* colptr, const int32_t* rowval,
                          const double* nzval) {
  return Guard::ptr([&]() -> amgs_mat* {
    if (m < 0 || n < 0 || !colptr) throw std::runtime_error("amgs_mat_create: bad argument");
    MatP A = make(m, n);
    std::copy(colptr, colptr + n + 1, A->colptr.begin());
    int64_t nnz = colptr[n];
    if (colptr[0] != 0 || nnz < 0) throw std::runtime_error("amgs_mat_create: colptr must be 0-based");
    for (int64_t j = 0; j < n; ++j)
      if (colptr[j + 1] < colptr[j]) throw std::runtime_error("amgs_mat_create: colptr not monotone");
    A->rowval.assign(rowval, rowval + nnz);
    A->nzval.assign(nzval, nzval + nnz);
    for (int64_t k = 0; k < nnz; ++k)
      if (rowval[k] < 0 || rowval[k] >= m) throw std::runtime_error("amgs_mat_create: row index out of range");
    return A.release();
  });
}
// an m x n matrix with room for nnz stored entries whose arrays the CALLER fills in place (amgs_mat_colptr / _rowval /
// _nzval): what the GPU half of the setup downloads into, without a staging copy.  The caller owns their consistency.
amgs_mat* amgs_mat_alloc(int64_t m, int64_t n, int64_t nnz) {
  return Guard::ptr([&]() -> amgs_mat* {
    if (m < 0 || n < 0 || nnz < 0 || nnz > (int64_t)INT32_MAX) throw std::runtime_error("amgs_mat_alloc: bad argument");
    MatP A = make(m, n);
    // (hundreds of megabytes that are written once, front to back: on 2 MiB pages — transparent huge pages are in `madvise`
    // mode on the target hosts — the first touch costs a page fault per 512 x as many bytes)
    auto huge = [](void* p, size_t bytes) {
      const uintptr_t a = ((uintptr_t)p + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1), e = ((uintptr_t)p + bytes) & ~(((uintptr_t)2 << 20) - 1);
      if (e > a && !std::getenv("AMGS_NO_HUGEPAGES")) madvise((void*)a, e - a, MADV_HUGEPAGE);   // advice only
    };
    A->rowval.reserve((size_t)nnz);
    huge(A->rowval.data(), sizeof(int32_t) * (size_t)nnz);
    A->rowval.resize((size_t)nnz);
    A->nzval.reserve((size_t)nnz);
    huge(A->nzval.data(), sizeof(double) * (size_t)nnz);
    A->nzval.resize((size_t)nnz);
    return A.release();
  });
}
void amgs_mat_free(amgs_mat* A) { delete A; }
int64_t amgs_mat_rows(const amgs_mat* A) { return A->m; }
int64_t amgs_mat_cols(const amgs_mat* A) { return A->n; }
int64_t amgs_mat_nnz(const amgs_mat* A) { return A->nnz(); }
const int32_t* amgs_mat_colptr(const amgs_mat* A) { return A->colptr.data(); }
const int32_t* amgs_mat_rowval(const amgs_mat* A) { return A->rowval.data(); }
const double* amgs_mat_nzval(const amgs_mat* A) { return A->nzval.data(); }
amgs_mat* amgs_mat_transpose(const amgs_mat* A) {
  return Guard::ptr([&]() -> amgs_mat* { return transpose(*A).release(); });
}
amgs_mat* amgs_mat_spgemm(const amgs_mat* X, const amgs_mat* Y) {
  return Guard::ptr([&]() -> amgs_mat* { return spgemm(*X, *Y).release(); });
}
int amgs_mat_is_symmetric(const amgs_mat* A) {
  if (A->m != A->n) return 0;
  MatP T = transpose(*A);
  return T->colptr == A->colptr && T->rowval == A->rowval && T->nzval == A->nzval;
}

amgs_mat* amgs_poisson(int ndim, const int64_t* dims) {
  return Guard::ptr([&]() -> amgs_mat* { return poisson(ndim, dims).release(); });
}

int amgs_classical_strength(const amgs_mat* At, double theta, amgs_mat** S, amgs_mat** T) {
  return Guard::rc([&]() {
    MatP s, t;
    classical_strength(*At, theta, s, t);
    *S = s.release();
    *T = t.release();
  });
}
amgs_mat* amgs_symmetric_strength(const amgs_mat* A, double theta, int bsr_flag) {
  return Guard::ptr([&]() -> amgs_mat* { return symmetric_strength(*A, theta, bsr_flag != 0).release(); });
}
int amgs_rs_splitting(amgs_mat* S, int32_t* splitting) {
  return Guard::rc([&]() {
    if (S->m != S->n) throw std::runtime_error("rs_splitting: S must be square");
    rs_splitting(*S, splitting);
  });
}
int amgs_rs_cf_splitting_patterns(int64_t n, const int32_t* Sp, const int32_t* Sj, const int32_t* Tp, const int32_t* Tj,
                                  int32_t* splitting) {
  return Guard::rc([&]() {
    if (n < 0 || !Sp || !Tp || !splitting) throw std::runtime_error("rs_cf_splitting: bad arguments");
    rs_cf_splitting_raw(n, Sp, Sj, Tp, Tj, splitting);
  });
}
amgs_mat* amgs_direct_interpolation(const amgs_mat* At, const amgs_mat* T, const int32_t* splitting) {
  return Guard::ptr([&]() -> amgs_mat* { return direct_interpolation(*At, *T, splitting).release(); });
}
amgs_mat* amgs_standard_aggregation(const amgs_mat* S) {
  return Guard::ptr([&]() -> amgs_mat* { return standard_aggregation(*S).release(); });
}
amgs_mat* amgs_fit_candidates(const amgs_mat* AggOp, const double* B, int nB, int vector_path, double tol,
                              double** Bc, int64_t* n_coarse) {
  return Guard::ptr([&]() -> amgs_mat* {
    std::vector<double> bc;
    MatP Q;
    if (vector_path) {
      if (nB != 1) throw std::runtime_error("fit_candidates: vector path needs nB == 1");
      Q = fit_candidates_vector(*AggOp, B, tol, bc);
    } else {
      Q = fit_candidates_matrix(*AggOp, B, nB, tol, bc);
    }
    *n_coarse = Q->n;
    *Bc = (double*)std::malloc(sizeof(double) * std::max<size_t>(bc.size(), 1));
    std::copy(bc.begin(), bc.end(), *Bc);
    return Q.release();
  });
}
amgs_mat* amgs_jacobi_prolongation(const amgs_mat* A, const amgs_mat* T, double omega) {
  return Guard::ptr([&]() -> amgs_mat* { return jacobi_prolongation(*A, *T, omega).release(); });
}
int amgs_improve_candidates(const amgs_mat* A, double* B, int nB, int iters) {
  return Guard::rc([&]() {
    if (A->m != A->n) throw std::runtime_error("improve_candidates: matrix must be square");
    improve_candidates_host(*A, B, nB, iters);
  });
}
int amgs_set_threads_here(int nthreads) {
#ifdef _OPENMP
  adopt_thread_cap();
  if (nthreads > 0) omp_set_num_threads(nthreads);
  return omp_get_max_threads();
#else
  (void)nthreads;
  return 1;
#endif
}
void amgs_free(void* p) { std::free(p); }

void amgs_default_options_rs(amgs_options* o) {
  o->theta = 0.25;  // Classical(0.25), classical.jl:8
  o->max_levels = 10;
  o->max_coarse = 10;
  o->hermitian = 1;
  o->sa_omega = 4.0 / 3.0;
  o->sa_improve_iters = 4;
  o->sa_B_is_vector = 1;
}
void amgs_default_options_sa(amgs_options* o) {
  amgs_default_options_rs(o);
  o->theta = 0.0;  // SymmetricStrength(), strength.jl:75
}

amgs_hier* amgs_ruge_stuben(const amgs_mat* A, const amgs_options* o) {
  return Guard::ptr([&]() -> amgs_hier* { return ruge_stuben(*A, *o); });
}
amgs_hier* amgs_smoothed_aggregation(const amgs_mat* A, const double* B, int nB, const amgs_options* o) {
  return Guard::ptr([&]() -> amgs_hier* { return smoothed_aggregation(*A, B, nB, *o); });
}
void amgs_hier_free(amgs_hier* h) { delete h; }
int amgs_hier_num_levels(const amgs_hier* h) { return (int)h->A.size(); }
const amgs_mat* amgs_hier_get(const amgs_hier* h, int level, int which) {
  int L = (int)h->A.size();
  if (level == L && which == 0) return h->final_A.get();
  if (level < 0 || level >= L) return nullptr;
  switch (which) {
    case 0: return h->A[level].get();
    case 1: return h->P[level].get();
    case 2: return h->R[level].get();
  }
  return nullptr;
}

}  // extern "C"
