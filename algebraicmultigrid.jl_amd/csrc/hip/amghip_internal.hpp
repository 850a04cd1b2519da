// amghip_internal.hpp — types and helpers shared by the host side of libamghip: error macros, device
// allocation helpers, the Gauss-Seidel schedule (GsSchedule), the stand-alone CSR operator (amgh_csr), the
// launch-shape tunables.  Included by amghip.hip only (one translation unit).
#pragma once
#include "amghip_kernels.hpp"
#include "gs_blocks.hpp"
#include "gs_flow.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "../../../include/amghip.h"

using namespace amgh;

#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return -(1000 + (int)e_);     \
  } while (0)
#define RC_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != AMGH_OK) return rc_; \
  } while (0)

namespace {

// byte accounting (amgh_device_bytes*): one real, one matrix entry (int32 column + real value) of THIS instance
constexpr int64_t kRealB = (int64_t)sizeof(real);
constexpr int64_t kEntB = 4 + (int64_t)sizeof(real);

template <class T>
int dev_alloc(T** p, int64_t count) {
  *p = nullptr;
  if (count <= 0) count = 1;
  hipError_t e = hipMalloc((void**)p, sizeof(T) * (size_t)count);
  if (e == hipErrorOutOfMemory) return AMGH_ENOMEM;
  if (e != hipSuccess) return -(1000 + (int)e);
  return AMGH_OK;
}
// Large transfers between pageable host memory and HBM.  hipMemcpy on pageable memory moves ~4 GB/s here (one host
// thread staging through the runtime's bounce buffer, and — for a freshly allocated destination — taking every page
// fault itself); the matrices of a 256^3 hierarchy are several GB each way.  staged_copy splits the range over a few
// host threads, each with its own stream and two pinned 8 MB buffers: the DMA of one piece runs while the thread
// copies (and faults in) the previous one.
struct StageLane {
  hipStream_t stream = nullptr;
  void* buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
};
struct StageSet {
  int device = -1;
  std::vector<StageLane> lanes;
};
constexpr size_t kStagePiece = size_t(8) << 20;
constexpr int kStageLanes = 8;
inline std::mutex& stage_mutex() { static std::mutex m; return m; }
inline std::vector<StageSet*>& stage_pool() { static std::vector<StageSet*> p; return p; }

inline StageSet* stage_acquire(int device) {
  {
    std::lock_guard<std::mutex> lk(stage_mutex());
    auto& pool = stage_pool();
    for (size_t i = 0; i < pool.size(); ++i)
      if (pool[i]->device == device) { StageSet* s = pool[i]; pool.erase(pool.begin() + i); return s; }
  }
  StageSet* s = new StageSet;
  s->device = device;
  s->lanes.resize(kStageLanes);
  bool ok = true;
  for (StageLane& L : s->lanes) {
    ok = ok && hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 2; ++k) {
      ok = ok && hipHostMalloc(&L.buf[k], kStagePiece, hipHostMallocDefault) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&L.ev[k], hipEventDisableTiming) == hipSuccess;
    }
  }
  if (!ok) {   // no pinned memory to be had: the caller falls back to plain hipMemcpy
    for (StageLane& L : s->lanes) {
      for (int k = 0; k < 2; ++k) { if (L.buf[k]) hipHostFree(L.buf[k]); if (L.ev[k]) hipEventDestroy(L.ev[k]); }
      if (L.stream) hipStreamDestroy(L.stream);
    }
    delete s;
    return nullptr;
  }
  return s;
}
inline void stage_release(StageSet* s) {
  std::lock_guard<std::mutex> lk(stage_mutex());
  stage_pool().push_back(s);
}

// dst <- src, `bytes` bytes, one side pageable host memory and the other device memory on the CURRENT device.
// Synchronous (like hipMemcpy): the data is there on return.
inline hipError_t staged_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  if (bytes < 4 * kStagePiece) return hipMemcpy(dst, src, bytes, kind);
  int device = 0;
  hipError_t e0 = hipGetDevice(&device);
  if (e0 != hipSuccess) return e0;
  StageSet* set = stage_acquire(device);
  if (!set) { (void)hipGetLastError(); return hipMemcpy(dst, src, bytes, kind); }
  const size_t npiece = (bytes + kStagePiece - 1) / kStagePiece;
  const int T = (int)std::min<size_t>(kStageLanes, npiece / 2);
  std::vector<hipError_t> err(T, hipSuccess);
  auto lane_fn = [&](int t) {
    hipSetDevice(device);
    StageLane& L = set->lanes[t];
    const size_t p0 = npiece * t / T, p1 = npiece * (t + 1) / T;
    auto piece_len = [&](size_t p) { return std::min(kStagePiece, bytes - p * kStagePiece); };
    hipError_t e = hipSuccess;
    if (kind == hipMemcpyDeviceToHost) {
      if (p0 < p1) {
        e = hipMemcpyAsync(L.buf[0], (const char*)src + p0 * kStagePiece, piece_len(p0), kind, L.stream);
        if (e == hipSuccess) e = hipEventRecord(L.ev[0], L.stream);
      }
      for (size_t p = p0; p < p1 && e == hipSuccess; ++p) {
        const int k = (int)((p - p0) & 1);
        if (p + 1 < p1) {
          e = hipMemcpyAsync(L.buf[k ^ 1], (const char*)src + (p + 1) * kStagePiece, piece_len(p + 1), kind, L.stream);
          if (e == hipSuccess) e = hipEventRecord(L.ev[k ^ 1], L.stream);
          if (e != hipSuccess) break;
        }
        e = hipEventSynchronize(L.ev[k]);
        if (e == hipSuccess) std::memcpy((char*)dst + p * kStagePiece, L.buf[k], piece_len(p));
      }
    } else {
      for (size_t p = p0; p < p1 && e == hipSuccess; ++p) {
        const int k = (int)((p - p0) & 1);
        if (p >= p0 + 2) e = hipEventSynchronize(L.ev[k]);   // the DMA that last read this buffer
        if (e != hipSuccess) break;
        std::memcpy(L.buf[k], (const char*)src + p * kStagePiece, piece_len(p));
        e = hipMemcpyAsync((char*)dst + p * kStagePiece, L.buf[k], piece_len(p), kind, L.stream);
        if (e == hipSuccess) e = hipEventRecord(L.ev[k], L.stream);
      }
    }
    hipError_t es = hipStreamSynchronize(L.stream);
    err[t] = e != hipSuccess ? e : es;
  };
  {
    std::vector<std::thread> th;
    int started = 0;
    for (; started < T - 1; ++started) {
      try { th.emplace_back(lane_fn, started); } catch (const std::system_error&) { break; }
    }
    for (int t = started; t < T; ++t) lane_fn(t);
    for (auto& x : th) x.join();
  }
  stage_release(set);
  for (hipError_t e : err) if (e != hipSuccess) return e;
  return hipSuccess;
}

// zero a (possibly very large) device range: in pieces of 1 GiB (a single multi-GB memset has been seen to fail with
// "invalid argument" — the slot arrays of a 512^3 fine level are 5.6 and 11 GB)
inline hipError_t dev_zero(void* p, size_t bytes, hipStream_t st) {
  const size_t piece = size_t(1) << 30;
  for (size_t o = 0; o < bytes; o += piece) {
    const hipError_t e = hipMemsetAsync((char*)p + o, 0, std::min(piece, bytes - o), st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// AMGH_VERBOSE: report (and clear) a launch error that is still pending — kernel launches with an invalid configuration
// do not fail any later synchronous call, they only show up in hipGetLastError
inline void dbg_pending(const char* where) {
  if (!getenv("AMGH_VERBOSE")) return;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) fprintf(stderr, "[amghip] pending HIP error at %s: %s\n", where, hipGetErrorString(e));
}

template <class T>
int dev_upload(T** p, const T* src, int64_t count) {
  RC_TRY(dev_alloc(p, count));
  if (count > 0) HIP_TRY(staged_copy(*p, src, sizeof(T) * (size_t)count, hipMemcpyHostToDevice));
  return AMGH_OK;
}
inline int grid_for(int64_t n, int threads = 256) {
  int64_t g = (n + threads - 1) / threads;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, 256 * 8));
}

// Gauss-Seidel dependency schedule of one square operator.
struct GsSchedule {
  int nlev = 0;
  std::vector<int32_t> lvl_ptr;  // host, nlev+1
  int32_t* d_lvl_ptr = nullptr;
  // level-permuted copy of the smoother's matrix
  int32_t* rowptr = nullptr;
  int32_t* col = nullptr;
  real* val = nullptr;
  int32_t* perm = nullptr;
  int32_t* dpos = nullptr;
  real* diag = nullptr;
  bool diag_shared = false;   // (merged Gauss-Seidel child: the parent's array — same rows in the same order)
  bool tiny_ok = false;       // the whole operator fits LDS (gs_chain_tiny_kernel): rows, nonzeros, levels and columns within kTiny*
  // wavefront of blocks (gs_blocks.hpp): the rows in BLOCK order (perm), one packed record per block, a launch per depth
  // of the quotient DAG.  When set, it is the only layout of this schedule (no level layout, no merged children).
  struct Bw {
    bw::Desc* blocks = nullptr; unsigned char* rec = nullptr; int32_t* ext_col = nullptr;
    std::vector<int32_t> launch_ptr;
    size_t lds_max = 0; int maxk = 0;
    int64_t rec_bytes = 0, rec_entries = 0, sum_depth = 0;
    double est_seconds = 0.0;
    // the whole sweep as ONE launch (gs_bw_chain_kernel): blocks chained by flags instead of kernel boundaries
    int32_t *dep_ptr = nullptr, *dep = nullptr, *sdep_ptr = nullptr, *sdep = nullptr;
    unsigned int* flags = nullptr; unsigned long long* head = nullptr;
    int32_t* err = nullptr;   // the process-wide error word (bw_err_word(): pinned host memory, not owned)
    int32_t nblocks = 0;
    bool on = false;          // this schedule IS a wavefront of blocks (rec may be absent when only the dataflow layout was kept)
    // the sweep as a DATAFLOW (gs_flow.hpp): rows published as they are computed, records streamed into registers
    struct FlowDev {
      bw::FlowDesc* fd = nullptr; unsigned char* srec = nullptr; uint32_t* aux = nullptr;
      int32_t* fl_mb = nullptr; uint16_t* fl_slot = nullptr; void* mbox = nullptr;
      int64_t nmail = 0; size_t lds_max = 0; int64_t bytes = 0;
      int64_t mail_stride = 0;   // bytes of one right-hand-side column's mailboxes (the spare cells included)
      int mcols = 0;             // columns mbox holds
      bool on = false;
      bool late_ok = false;      // the records' entries lie split around the padding (bw::Plan::late_ok): the relayed walk may sum the far half above the hand-over
      // a row-sharded operator's extended fetch lists (bw::FlowX): what a sweep pipelined across the ranks reads
      uint32_t* xaux = nullptr; int32_t* xfl_mb = nullptr; uint16_t* xfl_slot = nullptr; int32_t* xlist = nullptr;
      std::vector<int32_t> h_xfl_mb, h_row_cell_f, h_row_cell_b;   // host copies: the halo entries are patched with the neighbours' cells
      bool xon = false;
      // the DICTIONARY layout (bw::FlowDict): column records + every block's distinct value rows — what the relayed
      // single-column sweep streams instead of srec where the operator's rows repeat (stencils and their Galerkin products)
      unsigned char* crec = nullptr; unsigned char* dict = nullptr; int32_t* dict_ent = nullptr;
      size_t dict_lds = 0; bool dict_on = false;
    } flow;
  } bw;
  // the same operator as one record walked by a single wave (gs_wave_kernel): built when it fits (rows of at most
  // kWaveMaxK off-diagonal entries, at most kWaveMaxSteps steps, the LDS budget)
  unsigned char* ww_rec = nullptr; int ww_S = 0, ww_maxk = 0, ww_steps = 0; size_t ww_lds = 0;
  unsigned char* wq_rec = nullptr; int wq_E = 0, wq_steps = 0; size_t wq_lds = 0;   // ... with four lanes per row (gs_waveq_kernel): 4 n mini-rows of wq_E entries
  i4_t* rowmeta = nullptr;  // per permuted row {start, end, diagonal position, original row}
  i4_t* desc = nullptr;     // per dependency level {first row, end row, first nnz, end nnz}
  real* bp = nullptr;     // right-hand side in dependency-level order (scratch)
  real* xp = nullptr;     // x in dependency-level order (scratch, ncols entries per right-hand-side column)
  int cols_alloc = 1;       // right-hand-side columns bp / xp currently hold
  real* xil = nullptr;    // blocks of 2 / 4 / 8 right-hand sides on merged groups: [x ; s] INTERLEAVED (xstride positions x bs values), beside xp (gs_slot_il_kernel); allocated by the first such sweep
  int64_t xil_cap = 0;      // ... doubles allocated
  int xil_cols = 0;         // ... columns of the sweep that last wrote all of xil's x part (0: stale — something else wrote xp since)
  int bp_cols = 0;          // columns of b gathered into bp by the last level-ordered sweep (0: none)
  int32_t* permx = nullptr; // perm extended by the identity over halo columns
  std::vector<int32_t> h_perm;  // host copy of perm (level-ordered row -> original row), for level-ordered P / R copies
  int64_t n = 0, ncols = 0;
  int64_t bytes = 0;
  bool compacted = false;   // memory-lean child: the CSR copy holds only the rows no slot launch covers
  int64_t csr_bytes = 0, slot_bytes = 0;  // device bytes of the CSR copy (+ row / level descriptors) and of the slot arrays
  int64_t nnz = 0;          // entries of this system's rows (composite rows for a merged child)
  int64_t slot_total = 0;   // entries the slot arrays hold, zero padding included
  int64_t tri_nnz = 0;      // (on the parent, per direction [fwd, bwd]) entries of the pre-pass triangles
  int64_t tri_nnz_b = 0;
  struct Seg {  // dependency levels [l0, l1); launch shape
    int l0, l1; bool chain; int rows; int slot0, nslots;
    int sell_k = 0, sell_chunk0 = 0, sell_nchunks = 0;  // SELL-like layout of this group (0: none)
  };
  std::vector<Seg> segs;
  // slot layout of the wide levels (gs_slot_kernel)
  int32_t* wcol = nullptr; real* wval = nullptr; int32_t* slot_row = nullptr; i4_t* wmeta = nullptr;
  int slot_entries = kSlot;  // kSlot (gs_slot_kernel) or kBigSlot (gs_bigslot_kernel: long composite rows)
  // SELL-like copy of the merged groups (gs_sell_kernel)
  int32_t* scol = nullptr; real* sval = nullptr; i2_t* schunk = nullptr;
  int64_t sell_bytes = 0, sell_total = 0;
  // block-inverse path (small, densely coupled operators; see gs_block_kernel)
  struct Outer {
    int32_t* rowptr = nullptr; int32_t* col = nullptr; real* val = nullptr; real* tinv = nullptr;
    int32_t* near_ptr = nullptr; i2_t* near_pi = nullptr; real* near_val = nullptr;  // see gs_block_pipe_kernel
    // entries that reference blocks swept LATER (and the in-block other triangle): they read old x only, so
    // b - O_next x is one full-chip residual launch before the sequential sweep
    int32_t* nx_rowptr = nullptr; int32_t* nx_col = nullptr; real* nx_val = nullptr;
    // entries that reference EARLIER superblocks: final once that superblock is done, applied to the rows of a
    // superblock by one parallel launch before its sequential sweep
    int32_t* sp_rowptr = nullptr; int32_t* sp_col = nullptr; real* sp_val = nullptr;
  };
  Outer blk_f, blk_b;
  real* blk_diag = nullptr;
  real* blk_s = nullptr;  // b - O_next x (n entries per right-hand-side column)
  int nblk = 0;  // 0 = block path not used for this operator
  int super = 0; // blocks per superblock (0: the whole operator is one superblock)
  double blk_cond = 0.0;
  // small operators: the triangles of LARGE diagonal blocks (the whole matrix up to kDenseTriMax rows, else dti_B rows)
  // inverted densely.  A sweep walks the blocks: s = b - (everything outside the block's triangle) x on the block's
  // rows — a launch over the whole chip — then x_blk = T_blk^-1 s, one triangular GEMV: two launches per block
  // instead of B / 128 sequential steps of one workgroup.
  real* dti_f = nullptr;  // (D + L)_blk^-1 of every block, each nb x nb row-major, one after the other (natural order)
  real* dti_b = nullptr;  // (D + U)_blk^-1: the SAME array — lower triangles hold the forward inverses, upper the backward ones
  int dti_B = 0;          // rows per block (the last one may be shorter)
  std::vector<int64_t> dti_off;  // first element of block k in dti_f / dti_b
  double dti_cond = 0.0;  // largest inf-norm condition estimate of an in-block triangle
  // merged-level sweeps (gs_schedule.hpp, merge_build): per direction a child schedule over GROUPS of consecutive
  // dependency levels whose rows were made independent by substitution, plus the other triangle as a plain CSR
  // for the pre-pass s = b - T x.  Children share perm / bp / xp with this schedule; xp is then laid out as
  // [x in level order (ncols) ; s (n)] per right-hand-side column (xstride doubles apart).
  GsSchedule* mf = nullptr;
  GsSchedule* mb = nullptr;
  struct Tri { int32_t* rowptr = nullptr; int32_t* col = nullptr; real* val = nullptr; };
  Tri dtri_f, dtri_b;     // pre-pass matrices of the dense sweeps, natural order: A without the in-block lower (forward) / upper (backward) triangle and diagonal
  Tri tri_f, tri_b;       // forward pre-pass: entries of later levels + halo; backward: earlier levels + halo
  int merge_f = 1, merge_b = 1;  // dependency levels per group
  int64_t xstride = 0;    // doubles per column of xp (ncols, or ncols + n with merged children)
  // SOR as merged sweeps: (D/w + L) x = b - U x_old + ((1-w)/w) D x_old is a triangular solve with a scaled
  // diagonal, so the same construction applies, per relaxation factor w (composite coefficients carry powers of w).
  // Built on the first SOR sweep with that w; two factors are kept (pre- and post-smoother may differ).
  struct SorSet { real omega = 0.0; GsSchedule* f = nullptr; GsSchedule* b = nullptr; Tri tf, tb; bool built = false; };
  SorSet sor[2];
  int sor_next = 0;
  real s_key = 0.0;     // relaxation factor of the merged sweep that produced the current s (1 = Gauss-Seidel)
  bool diag_nonzero = false;  // no row keeps its x (zero diagonal): alternating merged sweeps may derive s from the last one
  int s_dir = -1;         // direction (0 fwd, 1 bwd) whose merged sweep last ran on the current xp / s, -1: none
  void free_dev() {
    for (GsSchedule** c : {&mf, &mb})
      if (*c) { (*c)->free_dev(); delete *c; *c = nullptr; }
    for (Tri* t : {&tri_f, &tri_b}) { hipFree(t->rowptr); hipFree(t->col); hipFree(t->val); *t = Tri(); }
    for (SorSet& ss : sor) {
      for (GsSchedule** c : {&ss.f, &ss.b})
        if (*c) { (*c)->free_dev(); delete *c; *c = nullptr; }
      for (Tri* t : {&ss.tf, &ss.tb}) { hipFree(t->rowptr); hipFree(t->col); hipFree(t->val); *t = Tri(); }
      ss = SorSet();
    }
    for (Outer* o : {&blk_f, &blk_b}) {
      hipFree(o->rowptr); hipFree(o->col); hipFree(o->val); hipFree(o->tinv);
      hipFree(o->near_ptr); hipFree(o->near_pi); hipFree(o->near_val);
      hipFree(o->nx_rowptr); hipFree(o->nx_col); hipFree(o->nx_val);
      hipFree(o->sp_rowptr); hipFree(o->sp_col); hipFree(o->sp_val);
      *o = Outer();
    }
    hipFree(dti_f); if (dti_b != dti_f) hipFree(dti_b);
    dti_f = dti_b = nullptr;
    for (Tri* t : {&dtri_f, &dtri_b}) { hipFree(t->rowptr); hipFree(t->col); hipFree(t->val); *t = Tri(); }
    hipFree(blk_diag); blk_diag = nullptr;
    hipFree(blk_s); blk_s = nullptr;
    hipFree(wcol); hipFree(wval); hipFree(slot_row); hipFree(wmeta); wcol = slot_row = nullptr; wval = nullptr; wmeta = nullptr;
    hipFree(scol); hipFree(sval); hipFree(schunk); scol = nullptr; sval = nullptr; schunk = nullptr;
    hipFree(ww_rec); ww_rec = nullptr;
    hipFree(wq_rec); wq_rec = nullptr;
    hipFree(xil); xil = nullptr; xil_cap = 0; xil_cols = 0;
    hipFree(bw.blocks); hipFree(bw.rec); hipFree(bw.ext_col);
    hipFree(bw.dep_ptr); hipFree(bw.dep); hipFree(bw.sdep_ptr); hipFree(bw.sdep); hipFree(bw.flags); hipFree(bw.head);
    hipFree(bw.flow.fd); hipFree(bw.flow.srec); hipFree(bw.flow.aux); hipFree(bw.flow.fl_mb); hipFree(bw.flow.fl_slot); hipFree(bw.flow.mbox);
    hipFree(bw.flow.xaux); hipFree(bw.flow.xfl_mb); hipFree(bw.flow.xfl_slot); hipFree(bw.flow.xlist);
    hipFree(bw.flow.crec); hipFree(bw.flow.dict); hipFree(bw.flow.dict_ent);
    bw = Bw();
    hipFree(d_lvl_ptr); hipFree(rowptr); hipFree(col); hipFree(val);
    hipFree(perm); hipFree(dpos); if (!diag_shared) hipFree(diag);
    hipFree(rowmeta); hipFree(desc); hipFree(bp); hipFree(xp); hipFree(permx);
    d_lvl_ptr = rowptr = col = perm = dpos = nullptr; val = diag = bp = xp = nullptr; rowmeta = desc = nullptr; permx = nullptr;
  }
};

}  // namespace

#include "dev_scan.hpp"

// Value-coded columns of an operator (StreamArgs::ccol): built once where the operator has at most 256 distinct values and
// fewer than 2^24 columns (code_values, csr_ops.hpp); the SpMV-type launches of big operators then stream 4 bytes per entry.
struct CodedCols {
  uint32_t* ccol = nullptr; real* vtab = nullptr; int n = 0;
  int64_t bytes = 0;
  void free_dev() { hipFree(ccol); hipFree(vtab); ccol = nullptr; vtab = nullptr; n = 0; bytes = 0; }
};

struct amgh_csr {
  CodedCols cc;
  bool xcd_map = false;   // SpMV launches of this operator with the XCD-contiguous mapping of the workgroups (amgh_finalize times both)
  int device = 0;
  int64_t nrows = 0, ncols = 0, nnz = 0;
  int32_t* rowptr = nullptr;
  int32_t* col = nullptr;
  real* val = nullptr;
  // smoother metadata in natural row order (Jacobi), built on demand
  int32_t* dpos = nullptr;
  real* diag = nullptr;
  GsSchedule* gs = nullptr;
  int gs_nrhs_hint = 0;   // right-hand-side columns the sweeps over this operator will carry (0 = unknown), see gs_build
  // a sweep pipelined across the ranks of a row-sharded level (amghip_dist.hpp), set around a csr_gs_sweep call: the sweep's
  // mailbox tag (0 = not pipelined), the neighbouring rank's mailboxes, the workgroups to launch (0 = one per block)
  uint32_t pipe_epoch = 0; const void* pipe_rmbox = nullptr; int pipe_grid = 0;
  int64_t bytes = 0;
};

namespace {

// launch-shape tunables of the per-level Gauss-Seidel launches (amgh_debug_set_tunable)
int g_gs_block_target = 256;   // aim for at least this many workgroups per wide level
int g_gs_min_rows = 4;         // but never fewer rows per workgroup than this
int g_gs_nnz_per_wg = 256;     // and about this many nonzeros per workgroup (one per thread)
int g_gs_threads = 256;
int g_gs_block_pipe = 1;        // software-pipelined block sweep (gs_block_pipe_kernel)
int g_gs_super = 8;             // block-inverse sweeps: blocks per superblock (0 = one launch for the whole operator); read at schedule build
int g_gs_coarse_lo = 1;         // ... and hand the coarse vectors of a level over in the next level's order (read at amgh_push_level)
int g_gs_keep_lo = 1;           // keep x in level order between pre- and post-smoother (level-ordered residual, R, P)
int g_gs_flip = 1;              // alternating merged sweeps: s of the next sweep from the last one (no matrix pass)
int g_gs_bigslot = 1;           // allow long-row slots (composite rows up to 2048 entries) when merging
int g_gs_merge = 16;             // merged-level sweeps: largest group of dependency levels tried (1 = off); read at schedule build
int g_gs_zone_t0_ns = 3000;      // ... its per-launch price (ns) and the floor below which a launch does not get cheaper (ns; 0 = none)
int g_gs_zone_floor_ns = 0;
int g_gs_zone = 1;               // groups of different depth along a sweep (deep where levels are small); read at schedule build
int g_gs_merge_force_maxn = 0;   // ... only on operators with at most this many rows (0 = all)
int g_gs_merge_force = 0;        // measurement hook: groups of exactly this many levels wherever they can be built (0 = cost model); read at schedule build
int g_gs_xcd_map = 1;           // XCD-contiguous slot -> workgroup mapping in gs_slot_kernel
int g_gs_slots = 1;             // wide levels from the slot layout (0 = CSR stream kernel)
int g_jacobi_zero = 1;          // Jacobi on x = 0 as a vector kernel (0 = the full sweep); read at every sweep
int g_rhs_il = 1;               // blocks of 2 / 4 / 8 / 16 right-hand sides: restriction and prolongation gather an interleaved copy of their input (0 = column by column); read at every cycle
int g_gs_tiny = 1;              // an operator that fits LDS entirely: 1 = gs_wave_kernel where its record was built, else gs_chain_tiny_kernel; 2 = gs_chain_tiny_kernel; 0 = gs_chain_kernel; read at every sweep
int g_gs_wave_quad = 1;         // the single-wave walk with four lanes per row where its record was built (gs_waveq_kernel: a row's additions as four interleaved partial sums); 0 = one lane per row (the scalar loop's bits); read at schedule build (0: not built) and at every sweep
int g_gs_bw = 1;                // wavefront of blocks for single-right-hand-side hierarchies (gs_blocks.hpp): 0 off, 1 where the cost model prefers it, 2 always (tests); read at schedule build
int g_gs_bw_rows = 512;         // ... rows per block aimed at
int g_gs_bw_flow = 1;           // the wavefront of blocks as a dataflow (gs_flow.hpp) where the pattern is structurally symmetric: 1 = on, 0 = off (chained / launched sweeps); build: the layout is only built when on; read at every sweep too
int g_gs_bw_spin = 0;           // polls before a wait of the dataflow / chained sweep gives up (0 = the default, ~seconds); test hook
int g_gs_bw_skip_pub = -1;      // test hook: the block with this ticket publishes nothing in dataflow sweeps (a forced protocol error); -1 = none
int g_gs_bw_nc = -1;            // columns of a block of right-hand sides one workgroup of the dataflow sweep carries (walker waves beside its one fetcher; -1 = 3 on the dictionary layout — records a quarter the size: the fetcher's shared polls count for more than a second record stream —, 2 on plain records: bs = 8 at 256^3 46.5 / 47.4 / 46.8 ms with 3 / 2 / 4 columns on the dictionary, 52.8 / 52.5 / 52.9 on plain records; 0 = as many as are instantiated: 4 for rows of <= 6 entries, else 3); 256^3, bs = 8, smoothers of the two block-ordered levels: 1: 16.3 + 10.0, 2: 12.0 + 10.1, 3: 12.0 + 10.4, 4: 13.0 + 10.4 ms; read at every sweep
int g_gs_bw_nrhs = 1;           // hierarchies built for blocks of right-hand sides get the dataflow layout too (0 = single-column hierarchies only); read at schedule build
int g_gs_bw_chain = 1;          // the wavefront of blocks as one launch per sweep, blocks chained by flags (0: one launch per depth of the quotient graph)
int g_stream_code = 1;          // SpMV-type launches (residual, restriction, prolongation of the level-ordered cycle) of operators of >= 2^18 rows stream value-coded columns where the operator has <= 256 distinct values (CodedCols; bitwise the same sums); read at amgh_finalize (0: not built) and at every launch
int g_trim_coded = 1;           // trimmed / lean footprint: operators of the level-ordered cycle that have value-coded columns keep ONLY them (amgh_finalize releases their 12-byte columns and values); 0 = keep both; read at amgh_finalize
int g_gs_bw_inorder = 0;        // 1: the relayed single-column sweep sums every row in stored entry order (the scalar loop's bits); 0 (default): where the records allow it (GsSchedule::Bw::FlowDev::late_ok) the products with the sweep's FAR side — x values that cannot change any more — are summed above the hand-over and the near half is added below it: the same Gauss-Seidel iterate, one reassociation per row (<= 1 ulp-level differences), a shorter dependent tail per step; read at every sweep
int g_gs_bw_dict = 1;           // the relayed single-column sweep reads the dictionary layout where a schedule carries one (bw::FlowDict: half the bytes of a 7-point level's sweep; bitwise the same); read at schedule build (0: not built) and at every sweep
int g_gs_bw_relay = 3;          // walker waves a single-column dataflow sweep relays a block's walk between (gs_relay.hpp: the one instantiated count, BW_RELAY_W; 0: one walker, gs_bw_flow_kernel — bitwise the same); read at every sweep
int g_gs_bw_grid = 0;           // workgroups of a relayed single-column sweep (fewer than blocks: the persistent form of gs_relay.hpp; 0: one per block); read at every sweep
int g_gs_bw_grid_long = 512;      // ... of levels with rows of more than 6 entries ON PLAIN RECORDS (critical-path bound: fewer resident blocks, faster hand-offs; the dictionary layout launches a workgroup per block); read at every sweep
int g_gs_flow_xzero = 1;        // a dataflow sweep that starts a smooth! call on x = 0 reads no x (0: fill + read as any other sweep — bitwise the same); read at every sweep
int g_gs_dup_launch = 0;        // measurement hook: every merged-group / level launch of a sweep issued 1 + this many times (idempotent); read at every sweep
int g_gs_bw_two_min_rows = 200000;  // ... operators with TWO offset classes (2-D grids) take the wavefront of blocks from this many rows where the cost model agrees (0 = never); round 4: 6 000 000; with the relayed dataflow sweep 512^2 / 1024^2 / 2048^2 Poisson V-cycles 4.82 -> 3.92 / 9.35 -> 8.50 / 21.6 -> 18.3 ms, 4096^2 39.1 -> 39.9 (profiles/r05_block_layout_threshold.log); read at schedule build
int g_gs_bw_min_rows = 30000;   // ... operators below this many rows keep the level schedules in mode 1 (half as many for rows of at most 7 entries).  Round 3 (chained kernel): 3 000 000; with the relayed dataflow sweep and its own cost model (Plan::est_flow_seconds) the block layout wins wherever that model says so: 48^3 ... 160^3 Poisson hierarchies -19 ... -28 % per V-cycle with the second level on it too (tools/minrows_sweep.py, profiles/r05_block_layout_threshold.log)
int g_gs_sample = 1;            // candidate group sizes of the merged sweeps from a sample of the groups (0 = every candidate built in full); read at schedule build
int g_gs_sell = 1;              // merged groups from the SELL-like layout where it was built (0 = slot kernels); build: read at schedule build too
int g_gs_lean = -1;             // footprint policy: -1 = AMGH_LEAN environment variable (unset: trim), 0 = full (every copy kept), 1 = lean, 2 = trim; read at schedule build
int g_gs_ept = 0;               // entries per thread of merged slot launches (0 = 2 when a group has more than 1024 slots, else 1)
int g_gs_lpr = 0;               // lanes per row in the row sums of merged slot launches (0 = by row length, 1 = one thread per row)
int g_gs_il = 1;                // blocks of 2 / 4 / 8 right-hand sides: merged groups gather from an interleaved copy of the sweep's vector (gs_slot_il_kernel, gs_sell_il_kernel; 0 = one gather per column); read at every sweep
int g_pcg_fused = 1;            // amgh_pcg: 1 = the recurrence between two cycles in 8 launches (second stages of the dot products fused with the scalar steps, the updates with the norm); 0 = one launch per operation (15) — bitwise the same iterates
int g_tail_dense_rows = 6144;   // the collapsed coarse tail: the first level with at most this many rows and everything below it become ONE dense operator (0 = off); read at amgh_finalize
int g_tail_dense = 1;           // ... and is applied where it has been built (0 = the per-level cycle: what the operator was built from); read at every cycle
int g_tail_dense_batch = 64;    // ... built from the library's own cycle on this many columns of the identity at a time (64 = the largest block of right-hand sides: 256^3 174 -> ~140 ms, C1 17.8 -> 12.0 ms against 32; profiles/r06_tail_dense.log); read at amgh_finalize
int g_stream_xcd = 1;           // big SpMV-type operators whose timing at amgh_finalize asked for it run with the XCD-contiguous workgroup mapping (0 = never); read at amgh_finalize and at every launch
int g_gs_dti_pre = 8;           // pre-pass of a dense-triangle block: rows per workgroup (8; 16 = the shape of rounds 2-5, bitwise the same); read at every sweep
int g_gs_tri_rb1 = 0;           // ... and under a SINGLE column: rows per workgroup of the dense triangle inverses (0 = one row, tri_gemv_kernel; 2 / 4 / 8: tri_gemm_kernel<1, RB> — bitwise the same); read at every sweep
int g_gs_tri_rb = 1;            // dense triangle inverses under a block of right-hand sides: 4 rows per workgroup (tri_gemm_kernel; 0 = one row, tri_gemv_kernel — bitwise the same); read at every sweep
int g_gs_dense_blk = 4096;       // ... rows per dense block above kDenseTriMax rows; read at schedule build
int g_gs_dense_tri = 1;          // small operators: sweeps through the dense inverse of the whole triangle (0 = block-inverse / exact order); build + sweep
int g_gs_block_inverse = 1;     // block-inverse sweeps for small densely coupled operators (0 = exact order everywhere)        // workgroup size of the per-level launches (64 or 256)

// bumped whenever a schedule buffer that captured hipGraphs may point to is reallocated or freed (xp / bp growth,
// SOR child eviction): handles drop their cached graph execs when it has moved on
std::atomic<unsigned long long> g_sched_epoch{0};   // (levels may be prepared on several host threads)

unsigned long long* g_chain_tim = nullptr;  // diagnostics buffer (amgh_debug_chain_timing)

constexpr int kChainWidth = 1024;  // dependency levels at most this wide are chained

__global__ void find_diag_kernel(const int32_t* rowptr, const int32_t* col, const real* val, int n,
                                 int32_t* dpos, real* diag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int dp = -1;
  real d = 0.0;
  // the reference keeps the LAST matching entry (d = ifelse(i == row, val, d))
  for (int j = rowptr[i]; j < rowptr[i + 1]; ++j)
    if (col[j] == i) { dp = j; d = val[j]; }
  dpos[i] = dp;
  diag[i] = d;
}

int csr_upload(amgh_csr* op, int device, int64_t nrows, int64_t ncols, const int32_t* rowptr,
               const int32_t* col, const real* val) {
  if (nrows < 0 || ncols < 0 || !rowptr) return AMGH_EINVAL;
  if (nrows >= INT32_MAX || ncols >= INT32_MAX) return AMGH_EUNSUPPORTED;
  const int64_t nnz = rowptr[nrows];
  if (rowptr[0] != 0 || nnz < 0) return AMGH_EINVAL;
  if (nnz > 0 && (!col || !val)) return AMGH_EINVAL;
  op->device = device;
  op->nrows = nrows;
  op->ncols = ncols;
  op->nnz = nnz;
  RC_TRY(dev_upload(&op->rowptr, rowptr, nrows + 1));
  RC_TRY(dev_upload(&op->col, col, nnz));
  RC_TRY(dev_upload(&op->val, val, nnz));
  op->bytes = (nrows + 1) * 4 + nnz * kEntB;
  return AMGH_OK;
}

void csr_free(amgh_csr* op) {
  if (!op) return;
  hipFree(op->rowptr); hipFree(op->col); hipFree(op->val);
  hipFree(op->dpos); hipFree(op->diag);
  op->cc.free_dev();
  if (op->gs) { op->gs->free_dev(); delete op->gs; }
  op->rowptr = op->col = op->dpos = nullptr; op->val = op->diag = nullptr; op->gs = nullptr;
}

int csr_ensure_diag(amgh_csr* op, hipStream_t st) {
  if (op->dpos) return AMGH_OK;
  const int64_t n = std::min(op->nrows, op->ncols);
  RC_TRY(dev_alloc(&op->dpos, op->nrows));
  RC_TRY(dev_alloc(&op->diag, op->nrows));
  if (n < op->nrows) {
    HIP_TRY(hipMemsetAsync(op->dpos, 0xff, sizeof(int32_t) * op->nrows, st));
    HIP_TRY(hipMemsetAsync(op->diag, 0, sizeof(real) * op->nrows, st));
  }
  if (n > 0)
    hipLaunchKernelGGL(find_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, op->rowptr, op->col,
                       op->val, (int)n, op->dpos, op->diag);
  HIP_TRY(hipGetLastError());
  op->bytes += op->nrows * kEntB;
  return AMGH_OK;
}


}  // namespace
