// amghip_dist.hpp — the row-sharded cycle behind the C ABI (include/amghip.h, "Row-sharded hierarchy").
//
// BASELINE.json config C4 / SURVEY.md section 8e: the fine levels of the hierarchy are partitioned by contiguous 1-D row
// ranges, one rank per GPU; every operator application is preceded by an exchange of the halo entries of its input
// vector; the coarse levels are collapsed onto the rank that owns all their rows (an ordinary amgh_t).  The reference is
// single-process (multilevel.jl:214-239 is what gets sharded), so there is no reference counterpart of the exchange.
//
// Halo exchange = NEIGHBOUR send/recv, not an all-gather: the halo region of a vector is laid out [local | halo sorted
// by global index], i.e. grouped by owner, so what a peer sends lands directly in place (no unpack kernel); the send
// side is one gather kernel into a packed buffer.  Transports:
//   RCCL   one process per GPU; ncclSend/ncclRecv grouped per exchange on a communication stream, overlapped with the
//          interior rows (rows that read no halo column) of the consuming operator; librccl is dlopen'ed on first use so
//          that libamghip.so loads on machines without it.
//   LOCAL  N ranks as N handles of ONE process (threads), peer copies between their streams; ranks may share a device.
//          This is what the single-GPU test box runs (virtual ranks) and what a single-process multi-GPU host can use.
//   IPC    one process per rank, no library in between (amghip_ipc.hpp): the packed send buffers are peer-mapped with
//          hipIpc memory handles, the hand-off is a pair of sequence flags per (vector, peer) in a POSIX shared-memory
//          segment that the streams themselves write and wait on (hipStreamWriteValue64 / hipStreamWaitValue64) — no
//          host round trip per exchange; the consumer pulls its halo entries with one device-to-device copy per peer.
//          Two processes may share a GPU (which RCCL refuses), so the single-GPU test box runs this path with real
//          processes; device = -1 builds the halo plans only (host memory, no GPU: the CPU multi-process test).
// Gauss-Seidel / SOR cannot be both lexicographic and parallel across a row partition: on sharded levels the sweep is
// exact inside a shard with the halo frozen per directional sweep (processor-block hybrid); Jacobi, residual,
// restriction and prolongation are exactly the single-GPU arithmetic.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <condition_variable>
#include <mutex>

namespace {

// ---- halo plan of one distributed vector -------------------------------------------------------------------------
struct PeerSpan { int64_t off = 0, cnt = 0; };

struct VecPlan {
  int id = -1;                       // position in the handle's list of plans (the same on every rank)
  int64_t r0 = 0, r1 = 0;            // my rows of the vector
  std::vector<int64_t> halo;         // sorted unique global indices outside [r0, r1) this rank reads
  std::vector<PeerSpan> recv;        // per peer: where its entries sit in the halo region (offsets relative to the halo)
  std::vector<PeerSpan> send;        // per peer: span of the packed send buffer
  std::vector<int32_t> h_send_idx;   // local indices gathered into the send buffer (all peers back to back)
  int32_t* d_send_idx = nullptr;     // ... on the device
  int64_t nsend = 0;
  real* d_sendbuf = nullptr;         // `copies` buffers of nsend entries
  int copies = 1;
  bool any = false;                  // any rank moves anything (same decision everywhere)
  int64_t nloc() const { return r1 - r0; }
  int64_t nhalo() const { return (int64_t)halo.size(); }
  void free_dev() { hipFree(d_send_idx); hipFree(d_sendbuf); d_send_idx = nullptr; d_sendbuf = nullptr; }
};

// ---- transports -------------------------------------------------------------------------------------------------

struct Transport {
  int rank = 0, nranks = 1;
  virtual ~Transport() {}
  virtual bool async() const { return false; }
  // setup: every rank contributes a list of int64, every rank gets all lists
  virtual int allgatherv_host(const std::vector<int64_t>& mine, std::vector<std::vector<int64_t>>& all) = 0;
  // a plan's send buffer exists: whatever the transport keeps per plan (collective; IPC maps the peers' buffers)
  virtual int send_copies() const { return 1; }
  virtual int plan_attach(VecPlan&) { return AMGH_OK; }
  virtual void plan_detach(VecPlan&) {}
  // data path of one exchange of plan pl, in this order:
  //   pack_target()    which copy of the send buffer the pack kernel fills (may enqueue waits on `st`)
  //   exchange_begin() send sendbuf[send[p].off .. +cnt) to peer p, receive recv[p].cnt entries from p into
  //                    recvbuf + recv[p].off.  `st` is the stream the packed send buffer was produced on and on which
  //                    the received data will be consumed; it may return before the data has arrived.
  //                    overlap = the caller has work for `st` that does not read the halo (interior rows): the transfer
  //                    then runs on the transport's own stream between two events; otherwise it is enqueued on `st`
  //                    itself — a second stream and two events cost more than the exchange (tools/ipc_pingpong.hip),
  //   exchange_finish() makes `st` wait for it.
  virtual int pack_target(VecPlan& pl, hipStream_t, real** sendbuf) { *sendbuf = pl.d_sendbuf; return AMGH_OK; }
  virtual int exchange_begin(VecPlan& pl, const real* sendbuf, real* recvbuf, hipStream_t st, bool overlap) = 0;
  virtual int exchange_finish(VecPlan& pl, hipStream_t st) = 0;
  // the same exchange on HOST buffers (device = -1 with a host tail: the C code's cycle executed in host memory — the CPU
  // multi-process test): every rank's packed entries travel through allgatherv_host as bit patterns, prefixed by the
  // offsets and counts of its per-peer spans (slow, exact)
  virtual int exchange_host(VecPlan& pl, const real* sendbuf, real* recvbuf) {
    const size_t N = (size_t)nranks;
    std::vector<int64_t> mine(2 * N + (size_t)pl.nsend);
    for (size_t p = 0; p < N; ++p) { mine[p] = pl.send[p].off; mine[N + p] = pl.send[p].cnt; }
    for (int64_t i = 0; i < pl.nsend; ++i) { const double v = (double)sendbuf[i]; std::memcpy(&mine[2 * N + (size_t)i], &v, 8); }
    std::vector<std::vector<int64_t>> all;
    RC_TRY(allgatherv_host(mine, all));
    for (size_t p = 0; p < N; ++p) {
      if ((int)p == rank || pl.recv[p].cnt <= 0) continue;
      const std::vector<int64_t>& v = all[p];
      if (v.size() < 2 * N) return AMGH_ESTATE;
      const int64_t off = v[(size_t)rank], cnt = v[N + (size_t)rank];
      if (cnt != pl.recv[p].cnt || off < 0 || 2 * N + (size_t)(off + cnt) > v.size()) return AMGH_ESTATE;
      for (int64_t i = 0; i < cnt; ++i) { double x; std::memcpy(&x, &v[2 * N + (size_t)(off + i)], 8); recvbuf[pl.recv[p].off + i] = (real)x; }
    }
    return AMGH_OK;
  }
  virtual int allreduce(double* v, int n, bool max_op) = 0;  // host values, in place
  virtual int barrier() = 0;
  // host wait for a stream whose work may depend on other ranks (a transport that can, bounds the wait)
  virtual int wait_stream(hipStream_t st) {
    HIP_TRY(hipStreamSynchronize(st));
    return AMGH_OK;
  }
};

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.lib ? &api : nullptr;
  tried = true;
  // whichever librccl the process already has (SONAME match), else the ROCm installation's
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (api.lib) break;
  }
  if (!api.lib) return nullptr;
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); ok = ok && p; return p; };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
  api.CommGetAsyncError = (decltype(api.CommGetAsyncError))sym("ncclCommGetAsyncError");
  api.Send = (decltype(api.Send))sym("ncclSend");
  api.Recv = (decltype(api.Recv))sym("ncclRecv");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
  api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(api.lib); api.lib = nullptr; return nullptr; }
  return &api;
}

// where a collective setup step rejected its input (AMGH_VERBOSE=1)
#define DIST_EINVAL(what)                                                                              \
  do {                                                                                                 \
    fprintf(stderr, "[amghip dist] invalid argument: %s (line %d)\n", what, __LINE__);               \
    return AMGH_EINVAL;                                                                                \
  } while (0)

#define NCCL_TRY(expr)                                              \
  do {                                                              \
    ncclResult_t r_ = (expr);                                       \
    if (r_ != ncclSuccess) return -(2000 + (int)r_);                \
  } while (0)

struct RcclTransport : Transport {
  RcclApi* api = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t cs = nullptr;      // communication stream
  // one (ready, done) event pair per halo plan: an exchange of one vector never re-records what another one waits on
  struct EvPair { hipEvent_t ready = nullptr, done = nullptr; bool on_cs = false; };
  std::vector<EvPair> evs;
  int64_t* d_i64 = nullptr;      // setup scratch
  int64_t d_i64_cap = 0;
  double* d_scal = nullptr;
  bool broken = false;
  // A rank that fails between two collectives must not leave its peers blocked in ncclRecv / ncclAllReduce for ever:
  // aborting the communicator makes their pending operations fail (they return -(2000 + ncclResult_t)).
  int fail(int rc) {
    if (!broken && comm && api) { api->CommAbort(comm); comm = nullptr; }
    broken = true;
    return rc;
  }
  ~RcclTransport() override {
    if (comm && api) api->CommDestroy(comm);
    if (cs) hipStreamDestroy(cs);
    for (EvPair& e : evs) { if (e.ready) hipEventDestroy(e.ready); if (e.done) hipEventDestroy(e.done); }
    hipFree(d_i64); hipFree(d_scal);
  }
  bool async() const override { return true; }
  int init(const void* id, int rank_, int nranks_) {
    api = rccl_api();
    if (!api) return AMGH_EUNSUPPORTED;
    rank = rank_; nranks = nranks_;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    NCCL_TRY(api->CommInitRank(&comm, nranks, uid, rank));
    HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    RC_TRY(dev_alloc(&d_scal, 64));
    return AMGH_OK;
  }
  int ensure_i64(int64_t n) {
    if (n <= d_i64_cap) return AMGH_OK;
    hipFree(d_i64); d_i64 = nullptr;
    RC_TRY(dev_alloc(&d_i64, n));
    d_i64_cap = n;
    return AMGH_OK;
  }
  int allgatherv_host(const std::vector<int64_t>& mine, std::vector<std::vector<int64_t>>& all) override {
    // counts, then the lists padded to the longest
    RC_TRY(ensure_i64(2 * (int64_t)nranks + 2));
    int64_t cnt = (int64_t)mine.size();
    HIP_TRY(hipMemcpyAsync(d_i64, &cnt, 8, hipMemcpyHostToDevice, cs));
    NCCL_TRY(api->AllGather(d_i64, d_i64 + 1, 1, ncclInt64, comm, cs));
    std::vector<int64_t> counts(nranks);
    HIP_TRY(hipMemcpyAsync(counts.data(), d_i64 + 1, sizeof(int64_t) * nranks, hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipStreamSynchronize(cs));
    int64_t mx = 1;
    for (int64_t c : counts) mx = std::max(mx, c);
    RC_TRY(ensure_i64(mx * (nranks + 1)));
    if (cnt) HIP_TRY(hipMemcpyAsync(d_i64, mine.data(), sizeof(int64_t) * cnt, hipMemcpyHostToDevice, cs));
    NCCL_TRY(api->AllGather(d_i64, d_i64 + mx, (size_t)mx, ncclInt64, comm, cs));
    std::vector<int64_t> flat((size_t)mx * nranks);
    HIP_TRY(hipMemcpyAsync(flat.data(), d_i64 + mx, sizeof(int64_t) * mx * nranks, hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipStreamSynchronize(cs));
    all.assign(nranks, {});
    for (int p = 0; p < nranks; ++p) all[p].assign(flat.begin() + (size_t)p * mx, flat.begin() + (size_t)p * mx + counts[p]);
    return AMGH_OK;
  }
  int exchange_begin(VecPlan& pl, const real* sendbuf, real* recvbuf, hipStream_t st, bool overlap) override {
    if (broken) return AMGH_ESTATE;
    const ncclDataType_t dt = sizeof(real) == 8 ? ncclDouble : ncclFloat;
    if (pl.id < 0) return AMGH_ESTATE;
    if ((size_t)pl.id >= evs.size()) evs.resize(pl.id + 1);
    EvPair& ev = evs[pl.id];
    ev.on_cs = overlap;
    hipStream_t ts = overlap ? cs : st;   // (nothing to overlap: the grouped send / recv sits on the caller's stream)
    if (overlap) {
      if (!ev.ready) {
        HIP_TRY(hipEventCreateWithFlags(&ev.ready, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev.done, hipEventDisableTiming));
      }
      hipError_t e = hipEventRecord(ev.ready, st);
      if (e == hipSuccess) e = hipStreamWaitEvent(cs, ev.ready, 0);
      if (e != hipSuccess) return fail(-(1000 + (int)e));
    }
    ncclResult_t r = api->GroupStart();
    for (int p = 0; p < nranks && r == ncclSuccess; ++p) {
      if (p == rank) continue;
      if (pl.send[p].cnt > 0) r = api->Send(sendbuf + pl.send[p].off, (size_t)pl.send[p].cnt, dt, p, comm, ts);
      if (r == ncclSuccess && pl.recv[p].cnt > 0)
        r = api->Recv(recvbuf + pl.recv[p].off, (size_t)pl.recv[p].cnt, dt, p, comm, ts);
    }
    const ncclResult_t r2 = api->GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return fail(-(2000 + (int)r));
    if (overlap) {
      const hipError_t e = hipEventRecord(ev.done, cs);
      if (e != hipSuccess) return fail(-(1000 + (int)e));
    }
    return AMGH_OK;
  }
  int exchange_finish(VecPlan& pl, hipStream_t st) override {
    if (pl.id < 0 || (size_t)pl.id >= evs.size()) return AMGH_ESTATE;
    if (evs[pl.id].on_cs) HIP_TRY(hipStreamWaitEvent(st, evs[pl.id].done, 0));
    return AMGH_OK;
  }
  int allreduce(double* v, int n, bool max_op) override {
    if (n > 64) return AMGH_EINVAL;
    if (broken) return AMGH_ESTATE;
    HIP_TRY(hipMemcpyAsync(d_scal, v, sizeof(double) * n, hipMemcpyHostToDevice, cs));
    const ncclResult_t r = api->AllReduce(d_scal, d_scal, (size_t)n, ncclDouble, max_op ? ncclMax : ncclSum, comm, cs);
    if (r != ncclSuccess) return fail(-(2000 + (int)r));
    HIP_TRY(hipMemcpyAsync(v, d_scal, sizeof(double) * n, hipMemcpyDeviceToHost, cs));
    RC_TRY(wait_stream(cs));
    return AMGH_OK;
  }
  // the communicator reports asynchronous failures (a peer that died, a link error): poll it while waiting so that a
  // rank whose peer is gone returns an error instead of blocking in hipStreamSynchronize for ever
  int wait_stream(hipStream_t st) override {
    for (int64_t spins = 0;; ++spins) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) return AMGH_OK;
      if (q != hipErrorNotReady) return -(1000 + (int)q);
      if (broken) return AMGH_ESTATE;
      if ((spins & 1023) == 1023 && comm && api->CommGetAsyncError) {
        ncclResult_t ar = ncclSuccess;
        if (api->CommGetAsyncError(comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
          return fail(-(2000 + (int)ar));
      }
      if (spins > 64) std::this_thread::yield();
    }
  }
  int barrier() override {
    double z = 0.0;
    return allreduce(&z, 1, false);
  }
};

}  // namespace

// N ranks of one process: a rendezvous area shared by their handles
struct amgh_local_group {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  bool broken = false;
  // published per rank for the current collective
  std::vector<const real*> sendbuf;
  std::vector<const std::vector<PeerSpan>*> sendspans;
  std::vector<int> device;
  std::vector<const std::vector<int64_t>*> lists;
  std::vector<std::vector<double>> vals;
  // returns false when the group was aborted (a rank failed): nobody hangs
  bool wait() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const uint64_t g = gen;
    if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); return true; }
    cv.wait(lk, [&] { return gen != g || broken; });
    return !broken;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    broken = true;
    cv.notify_all();
  }
};

namespace {

struct LocalTransport : Transport {
  amgh_local_group* g = nullptr;
  int device = 0;
  int allgatherv_host(const std::vector<int64_t>& mine, std::vector<std::vector<int64_t>>& all) override {
    g->lists[rank] = &mine;
    if (!g->wait()) return AMGH_ESTATE;
    all.assign(nranks, {});
    for (int p = 0; p < nranks; ++p) all[p] = *g->lists[p];
    if (!g->wait()) return AMGH_ESTATE;
    return AMGH_OK;
  }
  int exchange_begin(VecPlan& pl, const real* sendbuf, real* recvbuf, hipStream_t st, bool) override {
    const std::vector<PeerSpan>&send = pl.send, &recv = pl.recv;
    if (hipStreamSynchronize(st) != hipSuccess) { g->abort(); return AMGH_ESTATE; }  // my packed entries are complete
    g->sendbuf[rank] = sendbuf;
    g->sendspans[rank] = &send;
    g->device[rank] = device;
    if (!g->wait()) return AMGH_ESTATE;
    for (int p = 0; p < nranks; ++p) {
      if (p == rank || recv[p].cnt <= 0) continue;
      const PeerSpan& sp = (*g->sendspans[p])[rank];
      if (sp.cnt != recv[p].cnt) { g->abort(); return AMGH_ESTATE; }
      const real* src = g->sendbuf[p] + sp.off;
      hipError_t e = g->device[p] == device
                         ? hipMemcpyAsync(recvbuf + recv[p].off, src, sizeof(real) * sp.cnt, hipMemcpyDeviceToDevice, st)
                         : hipMemcpyPeerAsync(recvbuf + recv[p].off, device, src, g->device[p], sizeof(real) * sp.cnt, st);
      if (e != hipSuccess) { g->abort(); return -(1000 + (int)e); }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { g->abort(); return AMGH_ESTATE; }
    if (!g->wait()) return AMGH_ESTATE;  // nobody repacks a send buffer a peer is still reading
    return AMGH_OK;
  }
  int exchange_finish(VecPlan&, hipStream_t) override { return AMGH_OK; }
  int allreduce(double* v, int n, bool max_op) override {
    g->vals[rank].assign(v, v + n);
    if (!g->wait()) return AMGH_ESTATE;
    for (int i = 0; i < n; ++i) {
      double acc = g->vals[0][i];  // rank order: every rank computes the same bits
      for (int p = 1; p < nranks; ++p) acc = max_op ? std::max(acc, g->vals[p][i]) : acc + g->vals[p][i];
      v[i] = acc;
    }
    if (!g->wait()) return AMGH_ESTATE;
    return AMGH_OK;
  }
  int barrier() override { return g->wait() ? AMGH_OK : AMGH_ESTATE; }
};

// one local operator block: rows = my rows, columns = [my entries of the input vector | its halo]
struct DistOp {
  amgh_csr op;
  std::vector<int32_t> h_rowptr, h_col;   // host execution (device = -1): the block in host memory, columns in [local | halo]
  std::vector<real> h_val;
  int64_t h_ncols = 0;
  bool present = false;
  int32_t i0 = 0, i1 = 0;  // rows [i0, i1) read no halo column (interior): they can run while the halo is in flight
};

struct HostBlock {  // local rows with GLOBAL column indices, kept until finalize
  std::vector<int32_t> rowptr, col;
  std::vector<real> val;
  int64_t nrows = 0;
  bool present = false;
};

struct DistLevel {
  int64_t n_glob = 0, nc_glob = 0;
  std::vector<int64_t> cuts, ccuts;   // row partition of this level / of the next one
  HostBlock hA, hS, hP, hR;
  DistOp A, S, P, R;
  amgh_smoother_t pre{}, post{};
  real *x = nullptr, *b = nullptr, *res = nullptr, *tmp = nullptr;  // x: [local | halo], res: [local | halo of R's input]
  VecPlan rplan;                       // halo plan of res (read by R)
  DistOp* smat() { return S.present ? &S : &A; }
  // Gauss-Seidel / SOR PIPELINED across the ranks (dist_pipe_setup): the neighbours' mailboxes as this rank maps them
  struct Pipe {
    bool on = false;                   // every rank has the dataflow layout with extended lists on this level, each halo side comes from one neighbour
    const void *rprev = nullptr, *rnext = nullptr;   // mailboxes of rank - 1 (polled in forward sweeps) / rank + 1 (backward sweeps)
    void *ipc_prev = nullptr, *ipc_next = nullptr;   // ... when they were opened from IPC handles (closed at destroy)
    int grid = 0;                      // workgroups per launch (ranks sharing one device: the persistent form, every rank's workgroups resident)
    uint32_t epoch = 0;                // sweeps so far: the mailbox tag, the same on every rank
  } pipe;
};

void sort_unique(std::vector<int64_t>& v) {
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
}

}  // namespace

struct amgh_dist {
  int device = 0;
  Transport* tr = nullptr;
  hipStream_t stream = nullptr;
  std::vector<DistLevel*> levels;
  std::vector<VecPlan> xplan;          // one per sharded level + one for the first collapsed level (read by the last P)
  real *xt = nullptr, *bt = nullptr; // vectors of the first collapsed level: [all of it on the owner | halo elsewhere]
  amgh_t* tail = nullptr;              // the collapsed levels (on the rank that owns them), not owned
  bool finalized = false;
  bool host_only = false;              // device < 0: halo plans in host memory only, no GPU call anywhere
  amgh_coarse_fn host_tail = nullptr;  // ... unless the collapsed levels are given as a host callback (amgh_dist_set_host_tail): then the
  void* host_tail_user = nullptr;      // whole sharded cycle EXECUTES in host memory (vectors, exchanges, sweeps: plain loops)
  bool host_exec = false;
  std::vector<real> h_pack;            // host execution: the packed send entries of the exchange under way
  bool gs_exact = true;                // Gauss-Seidel / SOR across the shards: exact lexicographic order or the hybrid (amgh_dist_set_gs_mode)
  bool tail_agreed = false;            // every rank has seen (collectively, at the first data-path call) that the owner holds the collapsed levels
  bool pipe_mail_failed = false;       // the mailbox protocol probe between neighbouring ranks failed (dist_pipe_mail_probe): the levels sweep in turns
  bool pipe_serial = false;            // ranks of this process sharing the device were found NOT to run concurrently (dist_pipe_probe): the levels sweep in turns
  bool gs_pipe = true;                 // ... exact order as ONE pipelined sweep where the level allows it (DistLevel::Pipe::on), else the ranks in turn
  int nplans = 0;
  real *partial = nullptr, *scal = nullptr;
  int64_t ex_count = 0, ex_bytes = 0;  // halo exchanges / bytes sent by this rank since the last reset
  int overlap = 1;
};

namespace {

int plan_build(amgh_dist* d, VecPlan& pl, const std::vector<int64_t>& cuts, std::vector<int64_t> needs) {
  Transport* tr = d->tr;
  const int N = tr->nranks, me = tr->rank;
  pl.id = d->nplans++;
  pl.r0 = cuts[me]; pl.r1 = cuts[me + 1];
  sort_unique(needs);
  pl.halo = needs;
  for (int64_t g : pl.halo)
    if (g >= pl.r0 && g < pl.r1) DIST_EINVAL("halo entry inside the local range");
  pl.recv.assign(N, PeerSpan());
  pl.send.assign(N, PeerSpan());
  {  // halo entries are sorted, owners are contiguous ranges: one span per owner
    int64_t pos = 0;
    for (int p = 0; p < N; ++p) {
      const int64_t lo = pos;
      while (pos < (int64_t)pl.halo.size() && pl.halo[pos] < cuts[p + 1]) ++pos;
      pl.recv[p].off = lo;
      pl.recv[p].cnt = pos - lo;
    }
    if (pos != (int64_t)pl.halo.size()) DIST_EINVAL("column index beyond the vector");
  }
  std::vector<std::vector<int64_t>> all;
  RC_TRY(tr->allgatherv_host(pl.halo, all));
  std::vector<int32_t>& send_idx = pl.h_send_idx;
  send_idx.clear();
  pl.any = false;
  for (int p = 0; p < N; ++p) {
    pl.any = pl.any || !all[p].empty();
    pl.send[p].off = (int64_t)send_idx.size();
    if (p != me)
      for (int64_t g : all[p])
        if (g >= pl.r0 && g < pl.r1) send_idx.push_back((int32_t)(g - pl.r0));
    pl.send[p].cnt = (int64_t)send_idx.size() - pl.send[p].off;
  }
  pl.nsend = (int64_t)send_idx.size();
  if (d->host_only) return AMGH_OK;
  pl.copies = tr->send_copies();
  RC_TRY(dev_upload(&pl.d_send_idx, send_idx.data(), pl.nsend));
  RC_TRY(dev_alloc(&pl.d_sendbuf, pl.copies * pl.nsend));
  return tr->plan_attach(pl);
}

// global column -> position in [local | halo]
inline int32_t plan_localize(const VecPlan& pl, int64_t g) {
  if (g >= pl.r0 && g < pl.r1) return (int32_t)(g - pl.r0);
  const auto it = std::lower_bound(pl.halo.begin(), pl.halo.end(), g);
  return (int32_t)(pl.nloc() + (it - pl.halo.begin()));
}

void block_needs(const HostBlock& hb, int64_t c0, int64_t c1, std::vector<int64_t>& needs) {
  if (!hb.present) return;
  for (int32_t c : hb.col)
    if (c < c0 || c >= c1) needs.push_back(c);
}

int block_upload(amgh_dist* d, DistOp& dop, HostBlock& hb, const VecPlan& pl) {
  if (!hb.present) return AMGH_OK;
  const int64_t n = hb.nrows;
  std::vector<int32_t> lc(hb.col.size());
  const int64_t nloc = pl.nloc();
  for (size_t k = 0; k < hb.col.size(); ++k) lc[k] = plan_localize(pl, hb.col[k]);
  // interior rows: the longest middle range of rows without a halo column
  int32_t i0 = 0, i1 = (int32_t)n;
  const int64_t half = n / 2;
  for (int64_t i = 0; i < n; ++i) {
    bool halo = false;
    for (int32_t j = hb.rowptr[i]; j < hb.rowptr[i + 1] && !halo; ++j) halo = lc[j] >= nloc;
    if (!halo) continue;
    if (i < half) i0 = (int32_t)i + 1;
    else { i1 = (int32_t)i; break; }
  }
  dop.i0 = std::min(i0, i1); dop.i1 = i1;
  if (!d->host_only)
    RC_TRY(csr_upload(&dop.op, d->device, n, nloc + pl.nhalo(), hb.rowptr.data(), lc.data(), hb.val.data()));
  else if (d->host_exec) {
    dop.h_rowptr = hb.rowptr; dop.h_col = lc; dop.h_val = hb.val; dop.h_ncols = nloc + pl.nhalo();
    dop.op.nrows = n; dop.op.ncols = nloc + pl.nhalo(); dop.op.nnz = (int64_t)hb.val.size();
  }
  dop.present = true;
  hb = HostBlock();
  return AMGH_OK;
}

int take_block(HostBlock& hb, int64_t nrows, const int32_t* rowptr, const int32_t* col, const real* val) {
  if (!rowptr) return AMGH_OK;
  if (rowptr[0] != 0) return AMGH_EINVAL;
  const int64_t nnz = rowptr[nrows];
  if (nnz < 0 || (nnz > 0 && (!col || !val))) return AMGH_EINVAL;
  hb.nrows = nrows;
  hb.rowptr.assign(rowptr, rowptr + nrows + 1);
  hb.col.assign(col, col + nnz);
  hb.val.assign(val, val + nnz);
  hb.present = true;
  return AMGH_OK;
}

// ---- data path ---------------------------------------------------------------------------------------------------
int halo_begin(amgh_dist* d, VecPlan& pl, real* vec, bool overlap = false) {
  if (!pl.any || d->tr->nranks == 1) return AMGH_OK;
  if (d->host_exec) {   // host execution: pack, exchange, done (nothing is in flight behind this call)
    d->h_pack.resize((size_t)std::max<int64_t>(1, pl.nsend));
    for (int64_t i = 0; i < pl.nsend; ++i) d->h_pack[(size_t)i] = vec[pl.h_send_idx[(size_t)i]];
    ++d->ex_count;
    d->ex_bytes += (int64_t)sizeof(real) * pl.nsend;
    return d->tr->exchange_host(pl, d->h_pack.data(), vec + pl.nloc());
  }
  real* sendbuf = nullptr;
  RC_TRY(d->tr->pack_target(pl, d->stream, &sendbuf));
  if (pl.nsend > 0) {
    hipLaunchKernelGGL(gather_perm_kernel, dim3(grid_for(pl.nsend)), dim3(256), 0, d->stream, (const real*)vec,
                       (const int32_t*)pl.d_send_idx, sendbuf, (int)pl.nsend, (int64_t)0, (int64_t)0);
    HIP_TRY(hipGetLastError());
  }
  ++d->ex_count;
  d->ex_bytes += (int64_t)sizeof(real) * pl.nsend;
  return d->tr->exchange_begin(pl, sendbuf, vec + pl.nloc(), d->stream, overlap);
}
int halo_finish(amgh_dist* d, VecPlan& pl) {
  if (!pl.any || d->tr->nranks == 1 || d->host_exec) return AMGH_OK;
  return d->tr->exchange_finish(pl, d->stream);
}
int halo_exchange(amgh_dist* d, VecPlan& pl, real* vec) {
  RC_TRY(halo_begin(d, pl, vec));
  return halo_finish(d, pl);
}

int rows_apply(const amgh_csr* op, int mode, const real* x, const real* b, real* y, real omega, int32_t ra,
               int32_t rb, hipStream_t st) {
  if (rb <= ra) return AMGH_OK;
  StreamArgs a{};
  a.rowptr = op->rowptr; a.col = op->col; a.val = op->val;
  a.x = x; a.y = y; a.b = b; a.dpos = op->dpos; a.diag = op->diag; a.omega = omega;
  a.row_begin = ra; a.row_end = rb;
  a.ldx = op->ncols; a.ldy = op->nrows; a.ldb = op->nrows;
  switch (mode) {
    case M_SPMV: return launch_stream<M_SPMV>(a, st);
    case M_RESID: return launch_stream<M_RESID>(a, st);
    case M_ADD: return launch_stream<M_ADD>(a, st);
    case M_JACOBI: return launch_stream<M_JACOBI>(a, st);
  }
  return AMGH_EINVAL;
}

// ---- host execution (device = -1 + amgh_dist_set_host_tail): the operators as plain loops over the block in host memory, sums in
// stored entry order (the order of the stream kernels and of the scalar loops) ----
int rows_apply_host(const DistOp& dop, int mode, const real* x, const real* b, real* y, real omega) {
  const int64_t n = (int64_t)dop.h_rowptr.size() - 1;
  for (int64_t i = 0; i < n; ++i) {
    real acc = 0, dg = 0;
    for (int32_t j = dop.h_rowptr[(size_t)i]; j < dop.h_rowptr[(size_t)i + 1]; ++j) {
      const int32_t c = dop.h_col[(size_t)j];
      if (mode == M_JACOBI && c == i) { dg = dop.h_val[(size_t)j]; continue; }
      acc += dop.h_val[(size_t)j] * x[c];
    }
    switch (mode) {
      case M_SPMV: y[i] = acc; break;
      case M_RESID: y[i] = b[i] - acc; break;
      case M_ADD: y[i] += acc; break;
      case M_JACOBI: y[i] = dg != (real)0 ? ((real)1 - omega) * x[i] + omega * ((b[i] - acc) / dg) : x[i]; break;   // smoother.jl:113-141
      default: return AMGH_EINVAL;
    }
  }
  return AMGH_OK;
}
// one directional Gauss-Seidel / SOR sweep over the local rows in index order (smoother.jl:61-90, :193-221), halo entries as they stand
void gs_sweep_host(const DistOp& dop, bool backward, bool sor, real omega, real* x, const real* b) {
  const int64_t n = (int64_t)dop.h_rowptr.size() - 1;
  for (int64_t s = 0; s < n; ++s) {
    const int64_t i = backward ? n - 1 - s : s;
    real acc = 0, dg = 0;
    for (int32_t j = dop.h_rowptr[(size_t)i]; j < dop.h_rowptr[(size_t)i + 1]; ++j) {
      const int32_t c = dop.h_col[(size_t)j];
      if (c == i) dg = dop.h_val[(size_t)j]; else acc += dop.h_val[(size_t)j] * x[c];
    }
    if (dg == (real)0) continue;
    x[i] = sor ? ((real)1 - omega) * x[i] + (omega / dg) * (b[i] - acc) : (b[i] - acc) / dg;
  }
}

// y = op(vec) with the halo of vec exchanged first; interior rows run while the halo is in flight when the transport
// is asynchronous and the block is big enough for three launches to pay
int dist_apply(amgh_dist* d, DistOp& dop, int mode, VecPlan& pl, real* vec, const real* b, real* y, real omega,
               bool skip_exchange = false) {
  const amgh_csr* op = &dop.op;
  const int32_t n = (int32_t)op->nrows;
  if (d->host_exec) {
    if (!skip_exchange) RC_TRY(halo_exchange(d, pl, vec));
    return dop.present ? rows_apply_host(dop, mode, vec, b, y, omega) : AMGH_OK;
  }
  if (skip_exchange) return rows_apply(op, mode, vec, b, y, omega, 0, n, d->stream);
  const bool split = d->overlap && d->tr->async() && pl.any && d->tr->nranks > 1 && (dop.i1 - dop.i0) >= 65536;
  RC_TRY(halo_begin(d, pl, vec, split));
  if (split) {
    RC_TRY(rows_apply(op, mode, vec, b, y, omega, dop.i0, dop.i1, d->stream));
    RC_TRY(halo_finish(d, pl));
    RC_TRY(rows_apply(op, mode, vec, b, y, omega, 0, dop.i0, d->stream));
    return rows_apply(op, mode, vec, b, y, omega, dop.i1, n, d->stream);
  }
  RC_TRY(halo_finish(d, pl));
  return rows_apply(op, mode, vec, b, y, omega, 0, n, d->stream);
}

// smooth!(x, smoother, b) on a sharded level.  xzero: x (local part and halo) is zero on every rank, the first exchange
// would move zeros.  b_kept: the previous smooth! call of this cycle swept the same b with Gauss-Seidel / SOR.
int dist_smooth(amgh_dist* d, int l, const amgh_smoother_t& s, bool xzero, bool* b_kept) {
  DistLevel* L = d->levels[l];
  VecPlan& pl = d->xplan[l];
  DistOp* M = L->smat();
  const int64_t n = pl.nloc();
  bool fresh = xzero;
  if (d->host_exec) {   // host execution: the same control flow — exchanges, turns — over plain loops
    for (int it = 0; it < s.iter; ++it) {
      if (s.kind == AMGH_SMOOTH_JACOBI) {
        if (!fresh) RC_TRY(halo_exchange(d, pl, L->x));
        fresh = false;
        if (n > 0) { RC_TRY(rows_apply_host(*M, M_JACOBI, L->x, L->b, L->tmp, s.omega)); std::memcpy(L->x, L->tmp, sizeof(real) * (size_t)n); }
      } else if (s.kind == AMGH_SMOOTH_GS || s.kind == AMGH_SMOOTH_SOR) {
        const bool sor = s.kind == AMGH_SMOOTH_SOR;
        for (int dir = 0; dir < 2; ++dir) {
          const bool run = dir == 0 ? (s.sweep == AMGH_SWEEP_FORWARD || s.sweep == AMGH_SWEEP_SYMMETRIC)
                                    : (s.sweep == AMGH_SWEEP_BACKWARD || s.sweep == AMGH_SWEEP_SYMMETRIC);
          if (!run) continue;
          if (!fresh) RC_TRY(halo_exchange(d, pl, L->x));
          fresh = false;
          if (!d->gs_exact || d->tr->nranks == 1) {
            if (n > 0) gs_sweep_host(*M, dir == 1, sor, s.omega, L->x, L->b);
          } else {
            const int P = d->tr->nranks;
            for (int turn = 0; turn < P; ++turn) {
              const int q = dir == 0 ? turn : P - 1 - turn;
              if (q == d->tr->rank && n > 0) gs_sweep_host(*M, dir == 1, sor, s.omega, L->x, L->b);
              if (turn + 1 < P) RC_TRY(halo_exchange(d, pl, L->x));
            }
          }
        }
      } else if (s.kind != AMGH_SMOOTH_NONE) {
        return AMGH_EINVAL;
      }
    }
    *b_kept = false;
    return AMGH_OK;
  }
  for (int it = 0; it < s.iter; ++it) {
    if (s.kind == AMGH_SMOOTH_JACOBI) {
      if (n > 0) RC_TRY(csr_ensure_diag(&M->op, d->stream));
      if (fresh && g_jacobi_zero) {   // x = 0 on every rank: no exchange, no matrix pass
        if (n > 0) RC_TRY(csr_jacobi_zero(&M->op, s.omega, L->b, L->tmp, d->stream));
      } else {
        RC_TRY(dist_apply(d, *M, M_JACOBI, pl, L->x, L->b, L->tmp, s.omega, fresh));
      }
      fresh = false;
      if (n > 0) HIP_TRY(hipMemcpyAsync(L->x, L->tmp, sizeof(real) * n, hipMemcpyDeviceToDevice, d->stream));
    } else if (s.kind == AMGH_SMOOTH_GS || s.kind == AMGH_SMOOTH_SOR) {
      const bool sor = s.kind == AMGH_SMOOTH_SOR;
      for (int dir = 0; dir < 2; ++dir) {
        const bool run = dir == 0 ? (s.sweep == AMGH_SWEEP_FORWARD || s.sweep == AMGH_SWEEP_SYMMETRIC)
                                  : (s.sweep == AMGH_SWEEP_BACKWARD || s.sweep == AMGH_SWEEP_SYMMETRIC);
        if (!run) continue;
        if (!fresh) RC_TRY(halo_exchange(d, pl, L->x));
        fresh = false;
        if (!d->gs_exact || d->tr->nranks == 1) {
          // processor-block hybrid: every shard sweeps at once, exact inside, the halo frozen for this directional sweep
          if (n > 0)
            RC_TRY(csr_gs_sweep(&M->op, dir == 1, sor, s.omega, L->x, L->b, d->stream, true, true, 1, false, *b_kept));
        } else if (d->gs_pipe && L->pipe.on) {
          // exact lexicographic order across the shards as ONE sweep: every rank launches its dataflow sweep at once; a block
          // whose rows read the halo of the near side (lower ranks going forward, higher ranks going backward) polls those
          // rows' mailboxes in the neighbouring rank's array — they are published the moment they are computed, exactly as
          // between the blocks of one rank — and starts on them: the wavefront runs through the shards as through one level.
          // The far side's halo is what the exchange above delivered (old values, as the order demands)
          if (n > 0) {
            amgh_csr* op = &M->op;
            ++L->pipe.epoch;
            if ((L->pipe.epoch & 0x7fffffffu) == 0u) ++L->pipe.epoch;
            op->pipe_epoch = 0x80000000u | (L->pipe.epoch & 0x7fffffffu);   // (tags of the launches' own counts stay below 2^31)
            op->pipe_rmbox = dir == 0 ? L->pipe.rprev : L->pipe.rnext;
            op->pipe_grid = L->pipe.grid;
            const int rc = csr_gs_sweep(op, dir == 1, sor, s.omega, L->x, L->b, d->stream, true, true, 1, false, *b_kept);
            op->pipe_epoch = 0; op->pipe_rmbox = nullptr; op->pipe_grid = 0;
            RC_TRY(rc);
          }
        } else {
          // exact lexicographic order across the shards (smoother.jl:61-90 on the whole level): the ranks sweep IN TURN —
          // upward going forward, downward going backward — and every turn's boundary values travel before the next turn,
          // so that a shard reads new values of everything before it in the sweep and old values of everything behind it
          const int P = d->tr->nranks;
          for (int turn = 0; turn < P; ++turn) {
            const int q = dir == 0 ? turn : P - 1 - turn;
            if (q == d->tr->rank && n > 0)
              RC_TRY(csr_gs_sweep(&M->op, dir == 1, sor, s.omega, L->x, L->b, d->stream, true, true, 1, false, *b_kept));
            if (turn + 1 < P) RC_TRY(halo_exchange(d, pl, L->x));
          }
        }
        *b_kept = true;
      }
    } else if (s.kind != AMGH_SMOOTH_NONE) {
      return AMGH_EINVAL;
    }
  }
  if (!((s.kind == AMGH_SMOOTH_GS || s.kind == AMGH_SMOOTH_SOR) && s.iter > 0)) *b_kept = false;
  return AMGH_OK;
}

int dist_cycle(amgh_dist* d, int l, int cyc, bool xzero);

int dist_cycle_next(amgh_dist* d, int l, int cyc) {  // __solve_next! (multilevel.jl:200-212)
  RC_TRY(dist_cycle(d, l, cyc, true));
  if (cyc == AMGH_CYCLE_W) return dist_cycle(d, l, AMGH_CYCLE_W, false);
  if (cyc == AMGH_CYCLE_F) return dist_cycle(d, l, AMGH_CYCLE_V, false);
  return AMGH_OK;
}

// __solve! (multilevel.jl:214-239) on level l of the sharded hierarchy
int dist_cycle(amgh_dist* d, int l, int cyc, bool xzero) {
  const int lc = (int)d->levels.size();
  if (l == lc) {  // the collapsed levels: single-GPU cycle on their owner, nothing elsewhere
    if (d->host_exec) {   // host execution: the owner's callback IS the collapsed levels (x = 0 on entry, one application per visit)
      const int64_t nt = d->xplan[lc].nloc();
      if (nt > 0 && d->host_tail && d->host_tail(d->host_tail_user, d->bt, d->xt, nt) != 0) return AMGH_ESTATE;
      return AMGH_OK;
    }
    if (!d->tail) return AMGH_OK;
    if (d->tail->levels.empty()) return coarse_solve(d->tail, d->xt, d->bt);
    return cycle(d->tail, 0, d->xt, d->bt, cyc, xzero);
  }
  DistLevel* L = d->levels[l];
  VecPlan& xp = d->xplan[l];
  VecPlan& xpc = d->xplan[l + 1];
  real* xc = l + 1 == lc ? d->xt : d->levels[l + 1]->x;
  real* bc = l + 1 == lc ? d->bt : d->levels[l + 1]->b;
  bool b_kept = false;
  RC_TRY(dist_smooth(d, l, L->pre, xzero, &b_kept));
  RC_TRY(dist_apply(d, L->A, M_RESID, xp, L->x, L->b, L->res, 0.0));                                // res = b - A x
  RC_TRY(dist_apply(d, L->R, M_SPMV, L->rplan, L->res, nullptr, bc, 0.0));                          // b_c = R res
  const int64_t ncx = xpc.nloc() + xpc.nhalo();
  if (ncx > 0 && d->host_exec) std::memset(xc, 0, sizeof(real) * (size_t)ncx);
  else if (ncx > 0) HIP_TRY(hipMemsetAsync(xc, 0, sizeof(real) * ncx, d->stream));                             // coarse_x .= 0
  RC_TRY(dist_cycle_next(d, l + 1, cyc));
  RC_TRY(dist_apply(d, L->P, M_ADD, xpc, xc, nullptr, L->x, 0.0));                                   // x += P x_c
  RC_TRY(dist_smooth(d, l, L->post, false, &b_kept));
  return AMGH_OK;
}

int dist_apply_cycle(amgh_dist* d, int cyc, bool xzero) {
  if (d->levels.empty() && d->host_exec) return dist_cycle(d, 0, cyc, xzero);
  if (d->levels.empty()) {
    if (!d->tail) return AMGH_OK;
    if (d->tail->levels.empty()) return coarse_solve(d->tail, d->xt, d->bt);
    return cycle(d->tail, 0, d->xt, d->bt, cyc, xzero);
  }
  return dist_cycle(d, 0, cyc, xzero);
}

real* dist_x0(amgh_dist* d) { return d->levels.empty() ? d->xt : d->levels[0]->x; }
real* dist_b0(amgh_dist* d) { return d->levels.empty() ? d->bt : d->levels[0]->b; }

// sum over all ranks of x . y over the local entries (host result, same bits on every rank)
int dist_dot(amgh_dist* d, const real* x, const real* y, int64_t n, real* out) {
  real v = 0.0;
  if (d->host_exec) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc += (double)x[i] * (double)y[i];
    RC_TRY(d->tr->allreduce(&acc, 1, false));
    *out = (real)acc;
    return AMGH_OK;
  }
  if (n > 0) {
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kRedBlocks, (n + kThreads - 1) / kThreads));
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(kThreads), 0, d->stream, x, y, n, d->partial);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, d->stream, d->partial, nb, d->scal, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&v, d->scal, sizeof(real), hipMemcpyDeviceToHost, d->stream));
  }
  RC_TRY(d->tr->wait_stream(d->stream));
  double vd = (double)v;   // partial sums of the ranks are added in double whatever the eltype
  RC_TRY(d->tr->allreduce(&vd, 1, false));
  *out = (real)vd;
  return AMGH_OK;
}

// || b - A x || over all ranks (multilevel.jl:188-190); res of level 0 is the scratch, as in the reference
int dist_resnorm(amgh_dist* d, real* out) {
  real s = 0.0;
  if (d->levels.empty() && d->host_exec) return AMGH_EUNSUPPORTED;   // (host execution is for sharded levels over a host tail)
  if (d->levels.empty()) {
    real v = 0.0;
    if (d->tail) {
      amgh_t* h = d->tail;
      real* res = h->levels.empty() ? h->res_final : h->levels[0]->res;
      RC_TRY(fine_residual(h, d->xt, d->bt, res));
      RC_TRY(vec_dot(h, res, res, fine_n(h), h->scal, 0));
      HIP_TRY(hipMemcpyAsync(&v, h->scal, sizeof(real), hipMemcpyDeviceToHost, h->stream));
      RC_TRY(d->tr->wait_stream(h->stream));
    }
    double vd = (double)v;
    RC_TRY(d->tr->allreduce(&vd, 1, false));
    s = (real)vd;
  } else {
    DistLevel* L = d->levels[0];
    RC_TRY(dist_apply(d, L->A, M_RESID, d->xplan[0], L->x, L->b, L->res, 0.0));
    RC_TRY(dist_dot(d, L->res, L->res, d->xplan[0].nloc(), &s));
  }
  *out = std::sqrt(s);
  return AMGH_OK;
}

int dist_check(amgh_dist* d) {
  if (!d) return AMGH_EINVAL;
  if (!d->finalized) return AMGH_ESTATE;
  if (d->host_only && !d->host_exec) return AMGH_EUNSUPPORTED;  // plans only: there is no data path without a device or a host tail
  if (!d->tail_agreed) {
    // The owner of the collapsed levels may pass them after amgh_dist_finalize (amgh_dist_set_tail, not collective).  Whether it
    // HAS is settled collectively at the first data-path call: an owner without its tail fails this call on EVERY rank — alone it
    // would return while the others enter the cycle's exchanges and wait for it (RCCL has no timeout).
    double missing = (!d->host_only && !d->tail && d->xplan.back().nloc() > 0) ? 1.0 : 0.0;
    RC_TRY(d->tr->allreduce(&missing, 1, true));
    if (missing != 0.0) return AMGH_ESTATE;
    d->tail_agreed = true;
  }
  return AMGH_OK;
}

void dist_free(amgh_dist* d) {
  for (DistLevel* L : d->levels) {
    if (d->host_exec) { free(L->x); free(L->b); free(L->res); free(L->tmp); }
    if (!d->host_only) {
      for (DistOp* o : {&L->A, &L->S, &L->P, &L->R}) csr_free(&o->op);
      hipFree(L->x); hipFree(L->b); hipFree(L->res); hipFree(L->tmp);
      if (L->pipe.ipc_prev) (void)hipIpcCloseMemHandle(L->pipe.ipc_prev);
      if (L->pipe.ipc_next) (void)hipIpcCloseMemHandle(L->pipe.ipc_next);
      d->tr->plan_detach(L->rplan);
      L->rplan.free_dev();
    }
    delete L;
  }
  d->levels.clear();
  if (!d->host_only) {
    for (VecPlan& p : d->xplan) { d->tr->plan_detach(p); p.free_dev(); }
    hipFree(d->xt); hipFree(d->bt); hipFree(d->partial); hipFree(d->scal);
    if (d->stream) hipStreamDestroy(d->stream);
  }
  if (d->host_exec) { free(d->xt); free(d->bt); }
  delete d->tr;
  delete d;
}

int dist_new(amgh_dist** dp, int device, Transport* tr) {
  amgh_dist* d = new amgh_dist;
  d->device = device;
  d->tr = tr;
  d->host_only = device < 0;
  if (!d->host_only) {
    hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete tr; delete d; return -(1000 + (int)e); }
  }
  if (const char* ev = getenv("AMGH_DIST_OVERLAP")) d->overlap = atoi(ev);
  *dp = d;
  return AMGH_OK;
}

}  // namespace

#include "amghip_ipc.hpp"

const char* amgh_rccl_error_string(int code) {
  RcclApi* api = rccl_api();
  return api ? api->GetErrorString((ncclResult_t)code) : "librccl not loaded";
}

extern "C" {

int amgh_dist_rccl_available(void) { return rccl_api() ? 1 : 0; }

int amgh_dist_unique_id(void* id128) {
  if (!id128) return AMGH_EINVAL;
  RcclApi* api = rccl_api();
  if (!api) return AMGH_EUNSUPPORTED;
  ncclUniqueId uid;
  NCCL_TRY(api->GetUniqueId(&uid));
  std::memcpy(id128, &uid, sizeof uid);
  return AMGH_OK;
}

int amgh_dist_create_rccl(amgh_dist_t** dp, int device, int rank, int nranks, const void* id128) {
  if (!dp || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return AMGH_EINVAL;
  *dp = nullptr;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(device));
  RcclTransport* tr = new RcclTransport;
  int rc = tr->init(id128, rank, nranks);
  if (rc != AMGH_OK) { delete tr; return rc; }
  return dist_new(dp, device, tr);
}

int amgh_local_group_create(amgh_local_group_t** gp, int nranks) {
  if (!gp || nranks < 1 || nranks > 1024) return AMGH_EINVAL;
  amgh_local_group* g = new amgh_local_group;
  g->n = nranks;
  g->sendbuf.assign(nranks, nullptr);
  g->sendspans.assign(nranks, nullptr);
  g->device.assign(nranks, 0);
  g->lists.assign(nranks, nullptr);
  g->vals.assign(nranks, {});
  *gp = g;
  return AMGH_OK;
}
void amgh_local_group_destroy(amgh_local_group_t* g) { delete g; }
void amgh_local_group_abort(amgh_local_group_t* g) { if (g) g->abort(); }

int amgh_dist_create_local(amgh_dist_t** dp, int device, int rank, amgh_local_group_t* g) {
  if (!dp || !g || rank < 0 || rank >= g->n) return AMGH_EINVAL;
  *dp = nullptr;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return AMGH_EINVAL;
  HIP_TRY(hipSetDevice(device));
  LocalTransport* tr = new LocalTransport;
  tr->g = g; tr->rank = rank; tr->nranks = g->n; tr->device = device;
  return dist_new(dp, device, tr);
}

// IPC transport (one process per rank, shared memory + hipIpc): every rank passes the same `shm_name` ("/name", as
// shm_open takes it); rank 0 creates the segment, the call returns when all `nranks` processes have attached
// (collective).  device < 0: plans only (no GPU).
int amgh_dist_create_ipc(amgh_dist_t** dp, int device, int rank, int nranks, const char* shm_name) {
  if (!dp || !shm_name || shm_name[0] != '/' || std::strlen(shm_name) > 200 || nranks < 1 || nranks > kIpcMaxRanks ||
      rank < 0 || rank >= nranks)
    return AMGH_EINVAL;
  *dp = nullptr;
  if (device >= 0) {
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device >= ndev) return AMGH_EINVAL;
    HIP_TRY(hipSetDevice(device));
  }
  IpcTransport* tr = new IpcTransport;
  const int rc = tr->init(shm_name, rank, nranks, device < 0 ? -1 : device);
  if (rc != AMGH_OK) { delete tr; return rc; }
  return dist_new(dp, device < 0 ? -1 : device, tr);
}

int amgh_dist_set_gs_mode(amgh_dist_t* d, int mode) {
  if (!d || mode < 0 || mode > 2) return AMGH_EINVAL;
  d->gs_exact = mode != 0;
  d->gs_pipe = mode != 2;   // 1: pipelined where the level allows it (the default), 2: the ranks strictly in turn
  return AMGH_OK;
}
// 1 when Gauss-Seidel / SOR sweeps of sharded level `level` run as one sweep pipelined across the ranks under mode 1
int amgh_dist_gs_pipelined(const amgh_dist_t* d, int level) {
  if (!d || !d->finalized || level < 0 || level >= (int)d->levels.size()) return -1;
  return d->levels[level]->pipe.on ? 1 : 0;
}
// 1 when levels that could have been pipelined sweep in turns because the ranks sharing this device inside one process do not
// run concurrently (their streams share a hardware queue)
int amgh_dist_pipe_serialized(const amgh_dist_t* d) { return (d && d->finalized) ? (d->pipe_serial ? 1 : 0) : -1; }
int amgh_dist_pipe_protocol_failed(const amgh_dist_t* d) { return (d && d->finalized) ? (d->pipe_mail_failed ? 1 : 0) : -1; }

void amgh_dist_destroy(amgh_dist_t* d) {
  if (!d) return;
  if (!d->host_only) hipSetDevice(d->device);
  if (d->stream) d->tr->wait_stream(d->stream);
  // a transport whose peers read this rank's send buffers in place (IPC): a slower peer may still have copies of the last
  // exchanges queued — nobody unmaps or frees before everybody's stream has drained (a broken transport returns at once;
  // a peer that never arrives ends the wait with the transport's timeout and breaks it for everybody)
  if (d->tr && d->tr->async() && !d->host_only) (void)d->tr->barrier();
  if (d->tail && d->tail->stream == d->stream) {  // the borrowed tail goes back to its own stream (ours is about to die)
    d->tail->stream = d->tail->own_stream;
    d->tail->own_stream = nullptr;
    d->tail->ext_stream = false;
  }
  dist_free(d);
}

int amgh_dist_push_level(amgh_dist_t* d, int64_t n_global, int64_t nc_global, const int64_t* row_cuts,
                         const int64_t* crow_cuts, const int32_t* A_rowptr, const int32_t* A_col, const real* A_val,
                         const int32_t* S_rowptr, const int32_t* S_col, const real* S_val, const int32_t* P_rowptr,
                         const int32_t* P_col, const real* P_val, const int32_t* R_rowptr, const int32_t* R_col,
                         const real* R_val, const amgh_smoother_t* pre, const amgh_smoother_t* post) {
  if (!d || !row_cuts || !crow_cuts || !A_rowptr || !P_rowptr || !R_rowptr || n_global <= 0 || nc_global < 0)
    return AMGH_EINVAL;
  if (!smoother_valid(pre) || !smoother_valid(post)) return AMGH_EINVAL;
  if (d->finalized) return AMGH_ESTATE;
  if (n_global >= INT32_MAX || nc_global >= INT32_MAX) return AMGH_EUNSUPPORTED;
  const int N = d->tr->nranks, me = d->tr->rank;
  if (row_cuts[0] != 0 || row_cuts[N] != n_global || crow_cuts[0] != 0 || crow_cuts[N] != nc_global)
    DIST_EINVAL("row_cuts / crow_cuts do not span the level");
  for (int p = 0; p < N; ++p)
    if (row_cuts[p] > row_cuts[p + 1] || crow_cuts[p] > crow_cuts[p + 1]) return AMGH_EINVAL;
  if (!d->levels.empty()) {
    const DistLevel* prev = d->levels.back();
    if (prev->nc_glob != n_global) return AMGH_EINVAL;
    for (int p = 0; p <= N; ++p)
      if (prev->ccuts[p] != row_cuts[p]) return AMGH_EINVAL;
  }
  DistLevel* L = new DistLevel;
  L->n_glob = n_global; L->nc_glob = nc_global;
  L->cuts.assign(row_cuts, row_cuts + N + 1);
  L->ccuts.assign(crow_cuts, crow_cuts + N + 1);
  L->pre = *pre; L->post = *post;
  const int64_t nloc = row_cuts[me + 1] - row_cuts[me], ncloc = crow_cuts[me + 1] - crow_cuts[me];
  int rc = take_block(L->hA, nloc, A_rowptr, A_col, A_val);
  if (rc == AMGH_OK) rc = take_block(L->hS, nloc, S_rowptr, S_col, S_val);
  if (rc == AMGH_OK) rc = take_block(L->hP, nloc, P_rowptr, P_col, P_val);
  if (rc == AMGH_OK) rc = take_block(L->hR, ncloc, R_rowptr, R_col, R_val);
  if (rc != AMGH_OK) { delete L; return rc; }
  d->levels.push_back(L);
  return AMGH_OK;
}

int amgh_dist_set_tail(amgh_dist_t* d, amgh_t* tail) {
  if (!d) return AMGH_EINVAL;
  if (tail && (d->host_only || !tail->finalized || tail->nrhs != 1 || tail->device != d->device)) return AMGH_EINVAL;
  if (d->finalized) {
    // late binding (not collective): the owner built the collapsed levels beside amgh_dist_finalize — the shards' plans and
    // schedules and the tail's schedules are seconds of independent work — and passes them now, once
    if (!tail || d->tail || d->xplan.back().nloc() != fine_n(tail)) return AMGH_ESTATE;
    d->tail = tail;
    d->tail->ext_stream = true;
    d->tail->own_stream = d->tail->own_stream ? d->tail->own_stream : d->tail->stream;
    d->tail->stream = d->stream;
    return AMGH_OK;
  }
  d->tail = tail;
  return AMGH_OK;
}

// The collapsed levels as a HOST callback (device = -1 handles; before amgh_dist_finalize, on every rank — the owner's fn is the
// one that gets called, with x = 0 semantics of one visit: x = fn(b)).  With it the handle EXECUTES the sharded cycle in host
// memory: amgh_dist_precond_apply_d / amgh_dist_solve_d then take host pointers.
int amgh_dist_set_host_tail(amgh_dist_t* d, amgh_coarse_fn fn, void* user) {
  if (!d || !d->host_only) return AMGH_EINVAL;
  if (d->finalized) return AMGH_ESTATE;
  d->host_tail = fn; d->host_tail_user = user;
  d->host_exec = true;
  return AMGH_OK;
}

// Collective: every rank calls it after the same sequence of amgh_dist_push_level calls.
namespace {
uint64_t fnv1a(const char* s) { uint64_t h = 1469598103934665603ull; for (; *s; ++s) { h ^= (unsigned char)*s; h *= 1099511628211ull; } return h; }

// Collective: which sharded levels sweep Gauss-Seidel / SOR as one pipeline across the ranks.  A level qualifies when EVERY
// rank holds the dataflow layout with extended lists for its shard and reads its lower halo from rank - 1 only, its upper
// halo from rank + 1 only.  The ranks then exchange, per level, the mailbox of every row a neighbour reads (in the order of
// the halo exchange's send lists = the order of the receiver's halo) and a handle of their mailbox array; the halo entries
// of the extended fetch lists are pointed at the neighbour's cells and the arrays are mapped (same process: the pointer;
// another process: hipIpcOpenMemHandle).  Any failure anywhere switches the level back to the ranks sweeping in turn.
// Do the sweep streams of the ranks that share this device AND this process really run side by side?  A process's streams are
// spread over a handful of hardware queues; two ranks whose streams landed on one queue run their kernels one after the other,
// and a pipelined sweep — every rank's workgroups waiting for the neighbour's — would only end in its bounded polls.  Every such
// rank raises a flag from a kernel on its sweep stream and waits (bounded: ~50 ms) for the flags of ALL the others — not only its
// neighbours': the sweep is one dependency chain through the ranks, and rank 0 queued in front of rank 2 is as fatal going backward.
__global__ void pipe_probe_kernel(unsigned* mine, const unsigned* const* peers, int npeers, unsigned tag, long long ticks, unsigned* result) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  __hip_atomic_store(mine, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long t0 = wall_clock64();
  bool ok = false;
  for (;;) {
    bool all = true;
    for (int p = 0; p < npeers; ++p) all = all && __hip_atomic_load(peers[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == tag;
    if (all) { ok = true; break; }
    if (wall_clock64() - t0 > ticks) break;
    __builtin_amdgcn_s_sleep(32);
  }
  *result = ok ? 1u : 2u;
}
// collective (every rank calls it); *ok = 0 where some pair of same-process ranks on one device does not run concurrently
int dist_pipe_probe(amgh_dist* d, int64_t where, bool* ok) {
  Transport* tr = d->tr;
  const int N = tr->nranks, me = tr->rank;
  *ok = true;
  unsigned* flag = nullptr;   // [0]: my flag, [1]: the result
  RC_TRY(dev_alloc(&flag, 2));
  int rc = hipMemset(flag, 0, 8) == hipSuccess ? AMGH_OK : -1001;
  std::vector<int64_t> mine{(int64_t)getpid(), where, (int64_t)(uintptr_t)flag};
  std::vector<std::vector<int64_t>> all;
  if (rc == AMGH_OK) rc = tr->allgatherv_host(mine, all);
  double fail = rc == AMGH_OK ? 0.0 : 1.0;
  if (rc == AMGH_OK) {
    auto peer = [&](int p) -> const unsigned* {
      if (p < 0 || p >= N || all[(size_t)p].size() < 3 || all[(size_t)p][0] != mine[0] || all[(size_t)p][1] != where) return nullptr;
      return (const unsigned*)(uintptr_t)all[(size_t)p][2];
    };
    std::vector<const unsigned*> peers;
    for (int p = 0; p < N; ++p) if (p != me && peer(p)) peers.push_back(peer(p));
    const unsigned** d_peers = nullptr;
    if (!peers.empty()) {
      if (hipMalloc((void**)&d_peers, sizeof(void*) * peers.size()) != hipSuccess ||
          hipMemcpy((void*)d_peers, peers.data(), sizeof(void*) * peers.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); fail = 1.0; }
    }
    rc = tr->barrier();   // (every rank's flag is zeroed and known; the launches below leave the hosts within microseconds of each other)
    if (rc == AMGH_OK && !peers.empty() && fail == 0.0) {
      hipLaunchKernelGGL(pipe_probe_kernel, dim3(1), dim3(64), 0, d->stream, flag, (const unsigned* const*)d_peers, (int)peers.size(), 0x50495045u,
                         (long long)5000000, flag + 1);   // 50 ms of the 100 MHz clock
      unsigned res = 0;
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess ||
          hipMemcpy(&res, flag + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); fail = 1.0; }
      else if (res != 1u) fail = 1.0;
    }
    if (d_peers) hipFree((void*)d_peers);
    if (rc != AMGH_OK) fail = 1.0;
  }
  const int rc2 = tr->allreduce(&fail, 1, true);   // (also: nobody frees a flag a neighbour's kernel may still read)
  hipFree(flag);
  if (rc2 != AMGH_OK) return rc2;
  *ok = fail == 0.0;
  return AMGH_OK;
}

// The mailbox protocol ITSELF between neighbouring ranks, before any level depends on it: on memory allocated and mapped exactly
// as the levels' mailbox arrays are (hipMalloc; the neighbour's through its pointer — same process, peer access enabled when it
// lives on another device — or hipIpcOpenMemHandle) and with the instructions the sweeps use: a walker's write-through store
// (Mail<real>::store, sc1) polled by the neighbour's fetcher (Mail<real>::load_sys, sc0 sc1).  Two ranks on two devices is the case
// this exists for: an agent-scope store to coarse-grained memory is not PROMISED to become visible to another agent's system-scope
// loads mid-kernel; where it does not (or the mapping fails), the polls below run into their bound and every level sweeps with the
// ranks in turn.  kMailProbeRounds rounds in lockstep, a fresh cell per round (a cell is written once per epoch, as in the sweeps),
// values checked.  One lane per rank.
constexpr int kMailProbeRounds = 64;
__global__ void pipe_mail_probe_kernel(void* mine, const void* prev, const void* next, int me, int rounds, long long ticks, unsigned* result, int mute) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  typedef bw::Mail<real> M;
  const __amdgpu_buffer_rsrc_t rs_me = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_pv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(prev ? prev : (const void*)mine), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_nx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(next ? next : (const void*)mine), 0, 0x7ffffff0, 0x00020000);
  const long long t0 = wall_clock64();
  unsigned res = 1u;
  for (int r = 1; r <= rounds && res == 1u; ++r) {
    // side 0: the cells rank - 1 reads, side 1: the cells rank + 1 reads
    if (!mute) {
      M::store(rs_me, (unsigned)((0 * rounds + (r - 1)) * M::kBytes), (real)(1000 * me + r), (unsigned)r);
      M::store(rs_me, (unsigned)((1 * rounds + (r - 1)) * M::kBytes), (real)(1000 * me + r), (unsigned)r);
    }
    bool got_p = prev == nullptr, got_n = next == nullptr;
    while (!(got_p && got_n)) {
      if (!got_p) {
        const typename M::cell c = M::load_sys(rs_pv, (unsigned)((1 * rounds + (r - 1)) * M::kBytes));
        if (M::valid(c, (unsigned)r)) { got_p = true; if (M::value(c) != (real)(1000 * (me - 1) + r)) res = 3u; }
      }
      if (!got_n) {
        const typename M::cell c = M::load_sys(rs_nx, (unsigned)((0 * rounds + (r - 1)) * M::kBytes));
        if (M::valid(c, (unsigned)r)) { got_n = true; if (M::value(c) != (real)(1000 * (me + 1) + r)) res = 3u; }
      }
      if (wall_clock64() - t0 > ticks) { res = 2u; break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  *result = res;
}
// collective; *ok = false when some pair of neighbouring ranks cannot run the protocol (mapping, peer access, visibility, bound)
int dist_pipe_mail_probe(amgh_dist* d, int64_t where, bool* ok) {
  Transport* tr = d->tr;
  const int N = tr->nranks, me = tr->rank;
  *ok = true;
  const size_t bytes = (size_t)2 * kMailProbeRounds * bw::Mail<real>::kBytes + 16;
  void* cells = nullptr;
  double fail = 0.0;
  if (hipMalloc(&cells, bytes) != hipSuccess || hipMemset(cells, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); fail = 1.0; }
  hipIpcMemHandle_t hnd;
  std::memset(&hnd, 0, sizeof hnd);
  if (cells && hipIpcGetMemHandle(&hnd, cells) != hipSuccess) { (void)hipGetLastError(); std::memset(&hnd, 0, sizeof hnd); }
  std::vector<int64_t> mine{(int64_t)getpid(), where, (int64_t)(uintptr_t)cells};
  { int64_t hw[8]; std::memcpy(hw, &hnd, 64); for (int k = 0; k < 8; ++k) mine.push_back(hw[k]); }
  mine.push_back((int64_t)d->device);
  std::vector<std::vector<int64_t>> all;
  RC_TRY(tr->allgatherv_host(mine, all));
  const void* nb[2] = {nullptr, nullptr};
  void* opened[2] = {nullptr, nullptr};
  for (int side = 0; side < 2 && fail == 0.0; ++side) {
    const int p = side == 0 ? me - 1 : me + 1;
    if (p < 0 || p >= N) continue;
    const std::vector<int64_t>& v = all[(size_t)p];
    if (v.size() < 12 || v[2] == 0) { fail = 1.0; break; }
    if (v[0] == (int64_t)getpid()) {
      if (v[1] != where) {   // the same process, another device: the raw pointer is only usable with peer access
        int can = 0;
        const int pdev = (int)v[11];
        if (hipDeviceCanAccessPeer(&can, d->device, pdev) != hipSuccess || !can) { (void)hipGetLastError(); fail = 1.0; break; }
        const hipError_t e = hipDeviceEnablePeerAccess(pdev, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); fail = 1.0; break; }
        (void)hipGetLastError();
      }
      nb[side] = (const void*)(uintptr_t)v[2];
    } else {
      hipIpcMemHandle_t h2;
      std::memcpy(&h2, v.data() + 3, 64);
      if (hipIpcOpenMemHandle(&opened[side], h2, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); opened[side] = nullptr; fail = 1.0; break; }
      nb[side] = opened[side];
    }
  }
  unsigned* result = nullptr;
  if (dev_alloc(&result, 2) != AMGH_OK || hipMemset(result, 0, 8) != hipSuccess) fail = 1.0;
  // (every rank launches or none does: a rank that could not map its neighbour would leave that neighbour polling into its bound)
  RC_TRY(tr->allreduce(&fail, 1, true));
  if (fail == 0.0) {
    int rc = tr->barrier();
    if (rc == AMGH_OK) {
      long long ticks = 20000000;   // 200 ms of the 100 MHz clock
      if (const char* e = getenv("AMGH_MAIL_PROBE_MS")) ticks = std::max(1ll, atoll(e)) * 100000ll;
      int mute = 0;   // test hook: this rank publishes nothing — its neighbours run into the bound (a forced protocol failure)
      if (const char* e = getenv("AMGH_MAIL_PROBE_MUTE")) mute = atoi(e) == me ? 1 : 0;
      hipLaunchKernelGGL(pipe_mail_probe_kernel, dim3(1), dim3(64), 0, d->stream, cells, nb[0], nb[1], me, kMailProbeRounds, ticks, result, mute);
      unsigned res = 0;
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess ||
          hipMemcpy(&res, result, 4, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); fail = 1.0; }
      else if (res != 1u) fail = 1.0;
      if (getenv("AMGH_VERBOSE") && res != 1u) fprintf(stderr, "[amghip] rank %d: mailbox protocol probe with its neighbours: %s\n", me, res == 2u ? "no answer within the bound" : res == 3u ? "wrong value" : "not run");
    } else fail = 1.0;
    const int rc2 = tr->allreduce(&fail, 1, true);   // (also: nobody unmaps or frees cells a neighbour's kernel may still poll)
    if (rc2 != AMGH_OK) fail = 1.0;
  }
  for (int side = 0; side < 2; ++side) if (opened[side]) (void)hipIpcCloseMemHandle(opened[side]);
  (void)tr->barrier();
  if (result) hipFree(result);
  if (cells) hipFree(cells);
  *ok = fail == 0.0;
  return AMGH_OK;
}

int dist_pipe_setup(amgh_dist* d) {
  Transport* tr = d->tr;
  const int N = tr->nranks, me = tr->rank;
  const int lc = (int)d->levels.size();
  if (N < 2 || lc == 0) return AMGH_OK;
  if (const char* e = getenv("AMGH_DIST_PIPE")) if (atoi(e) == 0) return AMGH_OK;   // (measurement hook; the same on every rank)
  if (const char* e = getenv("AMGH_IPC_STAGED")) if (atoi(e) != 0) return AMGH_OK;  // (no peer mappings wanted at all)
  std::vector<double> bad((size_t)lc, 0.0);
  for (int l = 0; l < lc; ++l) {
    DistLevel* L = d->levels[l];
    VecPlan& xp = d->xplan[l];
    const amgh_csr* M = &L->smat()->op;
    const bool gs = L->pre.kind == AMGH_SMOOTH_GS || L->pre.kind == AMGH_SMOOTH_SOR || L->post.kind == AMGH_SMOOTH_GS || L->post.kind == AMGH_SMOOTH_SOR;
    bool ok = gs && xp.nloc() > 0 && M->gs && M->gs->bw.on && M->gs->bw.flow.on && M->gs->bw.flow.xon && M->gs->bw.flow.mbox;
    for (int p = 0; p < N && ok; ++p) {
      if (p == me) continue;
      if ((p < me - 1 || p > me + 1) && (xp.recv[p].cnt > 0 || xp.send[p].cnt > 0)) ok = false;
    }
    bad[(size_t)l] = ok ? 0.0 : 1.0;
  }
  RC_TRY(tr->allreduce(bad.data(), lc, true));
  char host[256] = {0}, bus[64] = {0};
  (void)gethostname(host, sizeof host - 1);
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, d->device) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof bus, "dev%d", d->device); }
  const int64_t where = (int64_t)((fnv1a(host) ^ (fnv1a(bus) * 31ull)) >> 1);
  {
    // no level qualifies anywhere: nothing to map, nothing to probe
    bool any_level = false;
    for (int l = 0; l < lc; ++l) any_level = any_level || bad[(size_t)l] == 0.0;
    if (!any_level) return AMGH_OK;
    // ranks of one process: the sweep streams must run side by side for the protocol probe as for the sweeps (see below)
    bool protocol = true;
    bool concurrent = true;
    RC_TRY(dist_pipe_probe(d, where, &concurrent));
    // Two ranks of this process on one hardware queue.  The runtime gives a new stream the hardware queue with the FEWEST streams
    // on it (tools/hwq_probe.hip: 16 streams in a row land 0 1 2 .. 7 7 .. 0 on 8 queues; empty one queue and the next TWO streams
    // both go there) — in a process that has created and destroyed streams unevenly, consecutive creations need not land on
    // different queues.  So the first rank of every (process, device) group first creates BALLAST streams, which level the
    // queues' counts (4 fresh streams behind 16 ballast streams: 4 queues, also after the ballast is gone), then the ranks take
    // fresh sweep streams ONE AFTER THE OTHER (rank order, a barrier between them), the ballast goes, and the probe runs again —
    // with 16, 64, 256 ballast streams.  Nothing has been enqueued on the old stream that a later call depends on (the collapsed
    // levels are bound to the handle's stream after this), so it is simply replaced.
    bool leader = true;   // the lowest rank of my process on my device
    if (!concurrent) {
      std::vector<int64_t> mine{(int64_t)getpid(), where};
      std::vector<std::vector<int64_t>> all;
      RC_TRY(tr->allgatherv_host(mine, all));
      for (int p = 0; p < me && p < (int)all.size(); ++p)
        if (all[(size_t)p].size() >= 2 && all[(size_t)p][0] == mine[0] && all[(size_t)p][1] == mine[1]) leader = false;
    }
    for (int attempt = 0; attempt < 3 && !concurrent; ++attempt) {
      double bad_stream = 0.0;
      std::vector<hipStream_t> ballast;
      if (leader) {
        const int nb = 16 << (2 * attempt);
        for (int k = 0; k < nb; ++k) {
          hipStream_t bs = nullptr;
          if (hipStreamCreateWithFlags(&bs, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
          ballast.push_back(bs);
        }
      }
      RC_TRY(tr->barrier());
      hipStream_t fresh = nullptr;
      for (int p = 0; p < N; ++p) {
        if (p == me && (hipStreamSynchronize(d->stream) != hipSuccess || hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess)) {
          (void)hipGetLastError(); bad_stream = 1.0; fresh = nullptr;
        }
        RC_TRY(tr->barrier());
      }
      // (old streams and ballast go only now: a stream destroyed between two creations would draw the next one onto its queue)
      if (fresh) { (void)hipStreamDestroy(d->stream); d->stream = fresh; }
      for (hipStream_t bs : ballast) (void)hipStreamDestroy(bs);
      RC_TRY(tr->allreduce(&bad_stream, 1, true));
      if (bad_stream != 0.0) break;
      RC_TRY(dist_pipe_probe(d, where, &concurrent));
      if (getenv("AMGH_VERBOSE") && me == 0)
        fprintf(stderr, "[amghip] sweep streams of the ranks sharing a device re-created in rank order (attempt %d): %s\n", attempt + 1, concurrent ? "concurrent" : "still serialised");
    }
    const bool side_by_side = concurrent;
    if (!side_by_side) {
      d->pipe_serial = true;
      if (getenv("AMGH_VERBOSE") && me == 0)
        fprintf(stderr, "[amghip] ranks sharing a device do not run concurrently (their streams share a hardware queue): Gauss-Seidel across the ranks in turns\n");
      return AMGH_OK;
    }
    RC_TRY(dist_pipe_mail_probe(d, where, &protocol));
    d->pipe_mail_failed = side_by_side && !protocol;
    if (side_by_side && !protocol) {
      if (getenv("AMGH_VERBOSE") && me == 0)
        fprintf(stderr, "[amghip] the mailbox protocol did not come back right between neighbouring ranks (mapping / peer access / visibility): Gauss-Seidel across the ranks in turns\n");
      return AMGH_OK;
    }
  }
  for (int l = 0; l < lc; ++l) {
    DistLevel* L = d->levels[l];
    VecPlan& xp = d->xplan[l];
    std::vector<int64_t> mine;
    double fail = 0.0;
    GsSchedule* g = bad[(size_t)l] == 0.0 ? L->smat()->op.gs : nullptr;
    if (g) {
      GsSchedule::Bw::FlowDev& fl = g->bw.flow;
      hipIpcMemHandle_t hnd;
      std::memset(&hnd, 0, sizeof hnd);
      if (hipIpcGetMemHandle(&hnd, fl.mbox) != hipSuccess) { (void)hipGetLastError(); std::memset(&hnd, 0, sizeof hnd); }   // (same-process peers do not need it)
      mine.push_back((int64_t)getpid()); mine.push_back(where); mine.push_back((int64_t)(uintptr_t)fl.mbox);
      int64_t hw[8]; static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
      std::memcpy(hw, &hnd, 64);
      for (int k = 0; k < 8; ++k) mine.push_back(hw[k]);
      // cells of the rows rank - 1 reads (backward cells), then of the rows rank + 1 reads (forward cells), each in send order
      for (int side = 0; side < 2; ++side) {
        const int p = side == 0 ? me - 1 : me + 1;
        const int64_t cnt = (p >= 0 && p < N) ? xp.send[p].cnt : 0;
        mine.push_back(cnt);
        for (int64_t i = 0; i < cnt; ++i) {
          const int32_t row = xp.h_send_idx[(size_t)(xp.send[p].off + i)];
          mine.push_back(side == 0 ? fl.h_row_cell_b[(size_t)row] : fl.h_row_cell_f[(size_t)row]);
        }
      }
    }
    std::vector<std::vector<int64_t>> all;
    RC_TRY(tr->allgatherv_host(mine, all));
    if (g) {
      GsSchedule::Bw::FlowDev& fl = g->bw.flow;
      int64_t nlo = 0;
      for (int p = 0; p < me; ++p) nlo += xp.recv[p].cnt;
      const int64_t nhi = xp.nhalo() - nlo;
      // the neighbours' lists: [pid, where, ptr, handle x 8, n_prev, cells..., n_next, cells...]
      auto side_cells = [&](int p, int side, const int64_t** cells, int64_t* cnt) -> bool {
        const std::vector<int64_t>& v = all[(size_t)p];
        if (v.size() < 12) return false;
        size_t pos = 11;
        for (int sd = 0; sd < 2; ++sd) {
          if (pos >= v.size()) return false;
          const int64_t c = v[pos++];
          if (c < 0 || pos + (size_t)c > v.size()) return false;
          if (sd == side) { *cells = v.data() + pos; *cnt = c; return true; }
          pos += (size_t)c;
        }
        return false;
      };
      const int64_t *lowc = nullptr, *upc = nullptr;
      int64_t nl = 0, nu = 0;
      if (nlo > 0 && !(me > 0 && side_cells(me - 1, 1, &lowc, &nl) && nl == nlo)) fail = 1.0;    // rank - 1's forward cells of what it sends up
      if (nhi > 0 && !(me + 1 < N && side_cells(me + 1, 0, &upc, &nu) && nu == nhi)) fail = 1.0;   // rank + 1's backward cells of what it sends down
      if (fail == 0.0) {
        for (int32_t& e : fl.h_xfl_mb) {
          if (e >= 0) continue;
          const int64_t h = (int64_t)(e & 0x7fffffff);
          const int64_t c = h < nlo ? lowc[h] : (h - nlo < nhi ? upc[h - nlo] : -1);
          if (c < 0 || c >= (int64_t)0x3fffffff) { fail = 1.0; break; }
          e = bw::kRemoteCell | (int32_t)c;
        }
      }
      if (fail == 0.0 && hipMemcpy(fl.xfl_mb, fl.h_xfl_mb.data(), sizeof(int32_t) * fl.h_xfl_mb.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); fail = 1.0; }
      // the neighbours' mailbox arrays
      int nshare = 0;
      for (int p = 0; p < N; ++p) if (all[(size_t)p].size() >= 2 && all[(size_t)p][1] == where) ++nshare;
      for (int side = 0; side < 2 && fail == 0.0; ++side) {
        const int p = side == 0 ? me - 1 : me + 1;
        if (p < 0 || p >= N || (side == 0 ? nlo : nhi) == 0) continue;
        const std::vector<int64_t>& v = all[(size_t)p];
        if (v.size() < 12) { fail = 1.0; break; }
        const void* ptr = nullptr; void* opened = nullptr;
        if (v[0] == (int64_t)getpid()) ptr = (const void*)(uintptr_t)v[2];
        else {
          hipIpcMemHandle_t hnd;
          std::memcpy(&hnd, v.data() + 3, 64);
          if (hipIpcOpenMemHandle(&opened, hnd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); fail = 1.0; break; }
          ptr = opened;
        }
        if (side == 0) { L->pipe.rprev = ptr; L->pipe.ipc_prev = opened; } else { L->pipe.rnext = ptr; L->pipe.ipc_next = opened; }
      }
      // ranks sharing this device: every rank's workgroups must be resident at once (a rank waits for its neighbour's)
      if (fail == 0.0 && nshare > 1) {
        // (both record layouts — the tunable gs_bw_dict switches between them at every sweep — must fit the grid)
        int cap = bw::relay_resident_blocks<real>(g->bw.maxk, g->bw.flow.lds_max);
        if (g->bw.flow.dict_on) cap = std::min(cap, bw::relay_resident_blocks<real>(g->bw.maxk, g->bw.flow.dict_lds, true));
        L->pipe.grid = std::max(1, cap / nshare);
        if (cap <= 0) fail = 1.0;
      }
    } else fail = bad[(size_t)l];
    RC_TRY(tr->allreduce(&fail, 1, true));
    L->pipe.on = fail == 0.0;
  }
  if (getenv("AMGH_VERBOSE") && me == 0)
    for (int l = 0; l < lc; ++l)
      fprintf(stderr, "[amghip] sharded level %d: Gauss-Seidel across the ranks %s\n", l, d->levels[l]->pipe.on ? "as one pipelined sweep" : "in turns");
  return AMGH_OK;
}
}  // namespace

int amgh_dist_finalize(amgh_dist_t* d) {
  if (!d) return AMGH_EINVAL;
  if (d->finalized) return AMGH_ESTATE;
  if (!d->host_only) HIP_TRY(hipSetDevice(d->device));
  Transport* tr = d->tr;
  const int N = tr->nranks, me = tr->rank, lc = (int)d->levels.size();
  // the first collapsed level: all rows on one rank (crow_cuts of the last sharded level says which); with no sharded
  // level at all the owner is the rank that holds the tail
  std::vector<int64_t> tcuts;
  if (lc > 0) tcuts = d->levels.back()->ccuts;
  else {
    double nt = d->tail ? (double)fine_n(d->tail) : 0.0;
    std::vector<double> v(N, 0.0);
    v[me] = nt;
    RC_TRY(tr->allreduce(v.data(), N, false));
    tcuts.assign(N + 1, 0);
    for (int p = 0; p < N; ++p) tcuts[p + 1] = tcuts[p] + (int64_t)v[p];
  }
  {
    const int64_t nt = tcuts[me + 1] - tcuts[me];
    // (plans only: the owner has no GPU handle to pass, the partition alone says who it is)
    // (a tail that is still being built — its schedules take seconds — may follow: amgh_dist_set_tail after finalize)
    const bool owner_ok = nt == 0 ? true : nt == tcuts[N] && (d->host_only || !d->tail || fine_n(d->tail) == nt);
    double bad = owner_ok ? 0.0 : 1.0;
    RC_TRY(tr->allreduce(&bad, 1, true));
    if (bad != 0.0) DIST_EINVAL("the collapsed levels must live on exactly one rank, which passes the tail");
  }
  // halo plans: x_l is read by A_l, S_l and by P_{l-1}; res_l by R_l
  d->xplan.assign(lc + 1, VecPlan());
  for (int l = 0; l <= lc; ++l) {
    const std::vector<int64_t>& cuts = l < lc ? d->levels[l]->cuts : tcuts;
    std::vector<int64_t> needs;
    if (l < lc) {
      block_needs(d->levels[l]->hA, cuts[me], cuts[me + 1], needs);
      block_needs(d->levels[l]->hS, cuts[me], cuts[me + 1], needs);
    }
    if (l >= 1) block_needs(d->levels[l - 1]->hP, cuts[me], cuts[me + 1], needs);
    RC_TRY(plan_build(d, d->xplan[l], cuts, std::move(needs)));
    if (l < lc) {
      std::vector<int64_t> rneeds;
      block_needs(d->levels[l]->hR, cuts[me], cuts[me + 1], rneeds);
      RC_TRY(plan_build(d, d->levels[l]->rplan, cuts, std::move(rneeds)));
    }
  }
  for (int l = 0; l < lc; ++l) {
    DistLevel* L = d->levels[l];
    VecPlan& xp = d->xplan[l];
    RC_TRY(block_upload(d, L->A, L->hA, xp));
    RC_TRY(block_upload(d, L->S, L->hS, xp));
    RC_TRY(block_upload(d, L->P, L->hP, d->xplan[l + 1]));
    RC_TRY(block_upload(d, L->R, L->hR, L->rplan));
    if (d->host_exec) {   // host execution: the level's vectors in host memory
      const int64_t nl = xp.nloc();
      auto zalloc = [](int64_t k) { return (real*)calloc((size_t)std::max<int64_t>(1, k), sizeof(real)); };
      L->x = zalloc(nl + xp.nhalo()); L->b = zalloc(nl); L->res = zalloc(nl + L->rplan.nhalo()); L->tmp = zalloc(nl);
      if (!L->x || !L->b || !L->res || !L->tmp) return AMGH_ENOMEM;
    }
    if (d->host_only) continue;
    const int64_t nloc = xp.nloc();
    RC_TRY(dev_alloc(&L->x, nloc + xp.nhalo()));
    RC_TRY(dev_alloc(&L->b, nloc));
    RC_TRY(dev_alloc(&L->res, nloc + L->rplan.nhalo()));
    RC_TRY(dev_alloc(&L->tmp, nloc));
    HIP_TRY(hipMemset(L->x, 0, sizeof(real) * std::max<int64_t>(1, nloc + xp.nhalo())));
    HIP_TRY(hipMemset(L->res, 0, sizeof(real) * std::max<int64_t>(1, nloc + L->rplan.nhalo())));
    // smoother metadata now, not inside the first timed cycle
    const bool jac = L->pre.kind == AMGH_SMOOTH_JACOBI || L->post.kind == AMGH_SMOOTH_JACOBI;
    const bool gs = L->pre.kind == AMGH_SMOOTH_GS || L->pre.kind == AMGH_SMOOTH_SOR || L->post.kind == AMGH_SMOOTH_GS ||
                    L->post.kind == AMGH_SMOOTH_SOR;
    amgh_csr* M = &L->smat()->op;
    if (nloc > 0 && jac) RC_TRY(csr_ensure_diag(M, d->stream));
    M->gs_nrhs_hint = 1;   // (the sharded cycle carries one right-hand side)
    if (nloc > 0 && gs) {
      // the rows the neighbours read get mailboxes, the halo columns places in the fetch lists: what a sweep pipelined across
      // the ranks needs (dist_pipe_setup decides, collectively, whether the level gets one)
      std::vector<unsigned char> pub_f((size_t)nloc, 0), pub_b((size_t)nloc, 0);
      int64_t nlo = 0;
      for (int p = 0; p < N; ++p) {
        if (p == me) continue;
        if (p < me) nlo += xp.recv[p].cnt;
        for (int64_t i = 0; i < xp.send[p].cnt; ++i) (p > me ? pub_f : pub_b)[(size_t)xp.h_send_idx[(size_t)(xp.send[p].off + i)]] = 1;
      }
      bw::FlowHalo fh; fh.nlo = nlo; fh.pub_f = pub_f.data(); fh.pub_b = pub_b.data();
      tl_flow_halo = N > 1 ? &fh : nullptr;
      tl_gs_level_rows = N > 1 ? L->n_glob : 0;
      const int rcg = csr_ensure_gs(M);
      tl_flow_halo = nullptr; tl_gs_level_rows = 0;
      RC_TRY(rcg);
    }
  }
  if (!d->host_only) RC_TRY(dist_pipe_setup(d));
  if (d->host_only) {
    if (d->host_exec) {
      VecPlan& tp = d->xplan[lc];
      d->xt = (real*)calloc((size_t)std::max<int64_t>(1, tp.nloc() + tp.nhalo()), sizeof(real));
      d->bt = (real*)calloc((size_t)std::max<int64_t>(1, tp.nloc()), sizeof(real));
      if (!d->xt || !d->bt) return AMGH_ENOMEM;
    }
    RC_TRY(tr->barrier());
    d->finalized = true;
    return AMGH_OK;
  }
  {
    VecPlan& tp = d->xplan[lc];
    RC_TRY(dev_alloc(&d->xt, tp.nloc() + tp.nhalo()));
    RC_TRY(dev_alloc(&d->bt, tp.nloc()));
    HIP_TRY(hipMemset(d->xt, 0, sizeof(real) * std::max<int64_t>(1, tp.nloc() + tp.nhalo())));
  }
  if (d->tail) {  // the collapsed levels enqueue on this handle's stream
    d->tail->ext_stream = true;
    d->tail->own_stream = d->tail->own_stream ? d->tail->own_stream : d->tail->stream;
    d->tail->stream = d->stream;
  }
  RC_TRY(dev_alloc(&d->partial, kRedBlocks));
  RC_TRY(dev_alloc(&d->scal, 8));
  HIP_TRY(hipDeviceSynchronize());
  RC_TRY(tr->barrier());
  d->finalized = true;
  return AMGH_OK;
}

int amgh_dist_num_sharded_levels(const amgh_dist_t* d) { return d ? (int)d->levels.size() : 0; }

int amgh_dist_local_range(const amgh_dist_t* d, int level, int64_t* r0, int64_t* r1) {
  if (!d || !d->finalized || level < 0 || level > (int)d->levels.size() || !r0 || !r1) return AMGH_EINVAL;
  *r0 = d->xplan[level].r0;
  *r1 = d->xplan[level].r1;
  return AMGH_OK;
}

// Halo plan of a distributed vector (tests, diagnostics): which = 0: x of `level` (0..lc; lc = the first collapsed level),
// which = 1: the residual of `level` (what R reads).  Counts, then optionally the arrays.
// out_counts: {nloc, nhalo, nsend, interior row begin, interior row end of the consuming operator (A / R)}
int amgh_dist_plan_info2(const amgh_dist_t* d, int level, int which, int64_t* out_counts, int64_t* halo_globals,
                         int32_t* send_idx, int64_t* send_cnt_per_peer, int64_t* recv_cnt_per_peer) {
  if (!d || !d->finalized || level < 0 || !out_counts || (which != 0 && which != 1)) return AMGH_EINVAL;
  const int lc = (int)d->levels.size();
  if (level > lc || (which == 1 && level >= lc)) return AMGH_EINVAL;
  const VecPlan& pl = which == 0 ? d->xplan[level] : d->levels[level]->rplan;
  out_counts[0] = pl.nloc(); out_counts[1] = pl.nhalo(); out_counts[2] = pl.nsend;
  const DistOp* op = level < lc ? (which == 0 ? &d->levels[level]->A : &d->levels[level]->R) : nullptr;
  out_counts[3] = op ? op->i0 : 0;
  out_counts[4] = op ? op->i1 : 0;
  if (halo_globals) std::copy(pl.halo.begin(), pl.halo.end(), halo_globals);
  if (send_idx) std::copy(pl.h_send_idx.begin(), pl.h_send_idx.end(), send_idx);
  for (int p = 0; p < d->tr->nranks; ++p) {
    if (send_cnt_per_peer) send_cnt_per_peer[p] = pl.send[p].cnt;
    if (recv_cnt_per_peer) recv_cnt_per_peer[p] = pl.recv[p].cnt;
  }
  return AMGH_OK;
}
int amgh_dist_plan_info(const amgh_dist_t* d, int level, int64_t* out_counts, int64_t* halo_globals, int32_t* send_idx,
                        int64_t* send_cnt_per_peer, int64_t* recv_cnt_per_peer) {
  return amgh_dist_plan_info2(d, level, 0, out_counts, halo_globals, send_idx, send_cnt_per_peer, recv_cnt_per_peer);
}

namespace {
// (host execution: the "device pointers" of the entry points are host pointers, copies are memcpy, nothing is enqueued)
int dist_copy(amgh_dist* d, real* dst, const real* src, int64_t n) {
  if (n <= 0) return AMGH_OK;
  if (d->host_exec) { std::memmove(dst, src, sizeof(real) * (size_t)n); return AMGH_OK; }
  HIP_TRY(hipMemcpyAsync(dst, src, sizeof(real) * n, hipMemcpyDeviceToDevice, d->stream));
  return AMGH_OK;
}
int dist_zero(amgh_dist* d, real* p, int64_t n) {
  if (n <= 0) return AMGH_OK;
  if (d->host_exec) { std::memset(p, 0, sizeof(real) * (size_t)n); return AMGH_OK; }
  HIP_TRY(hipMemsetAsync(p, 0, sizeof(real) * n, d->stream));
  return AMGH_OK;
}
int dist_setdev(amgh_dist* d) {
  if (d->host_exec) return AMGH_OK;
  HIP_TRY(hipSetDevice(d->device));
  return AMGH_OK;
}
}  // namespace

// ldiv!(x, p, b) on the sharded hierarchy: r_loc_d / z_loc_d hold this rank's rows of level 0 (device pointers).
// Enqueues on the handle's stream; amgh_dist_sync waits for it.
int amgh_dist_precond_apply_d(amgh_dist_t* d, const real* r_loc_d, real* z_loc_d, int cycle_) {
  RC_TRY(dist_check(d));
  if (cycle_ < 0 || cycle_ > 2) return AMGH_EINVAL;
  RC_TRY(dist_setdev(d));
  const VecPlan& p0 = d->xplan[0];
  const int64_t n = p0.nloc();
  if (n > 0 && (!r_loc_d || !z_loc_d)) return AMGH_EINVAL;
  RC_TRY(dist_copy(d, dist_b0(d), r_loc_d, n));
  RC_TRY(dist_zero(d, dist_x0(d), n + p0.nhalo()));
  RC_TRY(dist_apply_cycle(d, cycle_, true));
  return dist_copy(d, z_loc_d, dist_x0(d), n);
}

// _solve!(x, ml, b, cycle; ...) (multilevel.jl:158-198) on the sharded hierarchy: x_loc_d in/out (initial guess).
int amgh_dist_solve_d(amgh_dist_t* d, const real* b_loc_d, real* x_loc_d, int cycle_, int maxiter, double abstol,
                      double reltol, int calculate_residual, real* resid_hist, int* iters) {
  RC_TRY(dist_check(d));
  if (cycle_ < 0 || cycle_ > 2 || maxiter < 0) return AMGH_EINVAL;
  RC_TRY(dist_setdev(d));
  const VecPlan& p0 = d->xplan[0];
  const int64_t n = p0.nloc();
  if (n > 0 && (!b_loc_d || !x_loc_d)) return AMGH_EINVAL;
  RC_TRY(dist_copy(d, dist_b0(d), b_loc_d, n));
  RC_TRY(dist_copy(d, dist_x0(d), x_loc_d, n));
  real nb2 = 0.0;
  RC_TRY(dist_dot(d, dist_b0(d), dist_b0(d), n, &nb2));
  const real normb = std::sqrt(nb2);
  real normres = normb;
  if (normb != 0.0) abstol = std::max(reltol * normb, abstol);
  if (resid_hist) resid_hist[0] = normb;
  int itr = 1;
  while (itr <= maxiter && (!calculate_residual || normres > abstol)) {
    RC_TRY(dist_apply_cycle(d, cycle_, false));
    if (calculate_residual) {
      RC_TRY(dist_resnorm(d, &normres));
      if (resid_hist) resid_hist[itr] = normres;
    }
    ++itr;
  }
  if (iters) *iters = itr - 1;
  RC_TRY(dist_copy(d, x_loc_d, dist_x0(d), n));
  if (d->host_exec) return AMGH_OK;
  RC_TRY(d->tr->wait_stream(d->stream));
  return bw_err_check();   // (sharded levels and the tail sweep by dataflow kernels too: a poll give-up is this call's error)
}

// y_loc = A_level x_loc with the halo exchange in front (roofline hook of the sharded SpMV).  Enqueue only.
int amgh_dist_spmv_d(amgh_dist_t* d, int level, const real* x_loc_d, real* y_loc_d) {
  RC_TRY(dist_check(d));
  if (level < 0 || level >= (int)d->levels.size()) return AMGH_EINVAL;
  RC_TRY(dist_setdev(d));
  DistLevel* L = d->levels[level];
  VecPlan& pl = d->xplan[level];
  const int64_t n = pl.nloc();
  if (n > 0 && !y_loc_d) return AMGH_EINVAL;
  // x_loc_d == NULL: multiply the level's own x (what the last cycle left there) — no copy in front of the exchange
  if (n > 0 && x_loc_d && x_loc_d != L->x) RC_TRY(dist_copy(d, L->x, x_loc_d, n));
  return dist_apply(d, L->A, M_SPMV, pl, L->x, nullptr, n > 0 ? y_loc_d : L->tmp, 0.0);
}

int amgh_dist_sync(amgh_dist_t* d) {
  if (!d) return AMGH_EINVAL;
  if (d->host_only) return AMGH_OK;
  HIP_TRY(hipSetDevice(d->device));
  RC_TRY(d->tr->wait_stream(d->stream));
  return bw_err_check();   // (what amgh_dist_precond_apply_d enqueued has run: a sweep that gave up a poll is reported here)
}
int amgh_dist_barrier(amgh_dist_t* d) {
  if (!d) return AMGH_EINVAL;
  if (!d->host_only) {
    HIP_TRY(hipSetDevice(d->device));
    RC_TRY(d->tr->wait_stream(d->stream));
  }
  return d->tr->barrier();
}
int amgh_dist_allreduce(amgh_dist_t* d, double* v, int n, int max_op) {
  if (!d || !v || n < 1 || n > 64) return AMGH_EINVAL;
  if (!d->host_only) HIP_TRY(hipSetDevice(d->device));
  return d->tr->allreduce(v, n, max_op != 0);
}
// out: {halo exchanges, bytes this rank sent} since the last reset
int amgh_dist_stats(amgh_dist_t* d, int64_t* out2, int reset) {
  if (!d || !out2) return AMGH_EINVAL;
  out2[0] = d->ex_count; out2[1] = d->ex_bytes;
  if (reset) d->ex_count = d->ex_bytes = 0;
  return AMGH_OK;
}
int64_t amgh_dist_device_bytes(const amgh_dist_t* d) {
  if (!d) return 0;
  int64_t b = 0;
  for (const DistLevel* L : d->levels) {
    b += L->A.op.bytes + L->S.op.bytes + L->P.op.bytes + L->R.op.bytes;
    b += (int64_t)sizeof(real) * (3 * (L->cuts[d->tr->rank + 1] - L->cuts[d->tr->rank]));
  }
  return b;
}
void* amgh_dist_stream(amgh_dist_t* d) { return d ? (void*)d->stream : nullptr; }

}  // extern "C"
