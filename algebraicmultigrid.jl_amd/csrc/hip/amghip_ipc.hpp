// amghip_ipc.hpp — IPC transport of the row-sharded cycle: one PROCESS per rank and nothing between the processes but
// shared memory ("direct xGMI peer reads" of 512 KiB halo faces instead of a collective library).
//
//   rendezvous   a POSIX shared-memory segment named by the caller (rank 0 creates it, the others attach): a barrier,
//                per-rank slots for host all-reduces, the IPC handles of the plan under construction, the abort word.
//                Variable-length lists (the halo needs at setup) travel through one shared-memory file per rank.
//   halo data    every plan's packed send buffer is exported with hipIpcGetMemHandle and mapped by the ranks that read
//                it; an exchange is: producer packs into copy (k & 1) of its buffer and its STREAM writes k into the
//                plan's `ready` flag (hipStreamWriteValue64 on a flag page of the segment, host-registered by every
//                process); the consumer's stream waits for that flag (hipStreamWaitValue64), pulls its entries with one
//                copy kernel per producer straight into the halo region of the vector (the halo is sorted by owner, so
//                they land in place), and writes k into the pair's `done` flag, which the producer's stream waits on
//                before it reuses the copy (exchange k + 2).  No host thread is involved in an exchange.  The transfer is
//                enqueued on the caller's stream; only when interior rows of the consuming operator are to run
//                meanwhile does it go to the communication stream between two events (a second stream costs more than
//                an exchange of this size: tools/ipc_pingpong.hip).
//   failure      host waits are bounded (AMGH_IPC_TIMEOUT_S, default 300) and watch the peers' pids; a rank that gives
//                up sets the abort word and RELEASES every flag (writes a huge sequence number), so that no stream of
//                any rank keeps waiting for a producer that is gone; from then on every call returns AMGH_ESTATE.
// Two processes may share one GPU (the single-GPU test box does); across GPUs the mapped buffers are peer memory.
// device < 0: plans only — the same rendezvous with no GPU call at all (the CPU multi-process test).
#pragma once

#include <cerrno>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int kIpcMaxRanks = 64;
constexpr int kIpcMaxPlans = 128;
constexpr uint64_t kIpcMagic = 0x414d474849504331ull;  // "AMGHIPC1"
constexpr uint64_t kIpcReleased = 1ull << 62;           // sequence number no exchange reaches: releases every wait
constexpr size_t kIpcFlagStride = 64;                   // one flag per cache line

struct IpcHeader {
  std::atomic<uint64_t> magic;
  int32_t nranks;
  std::atomic<int32_t> abort_word;
  std::atomic<uint32_t> bar_count, bar_gen;
  std::atomic<int32_t> attached;
  int32_t pid[kIpcMaxRanks];
  double vals[kIpcMaxRanks][64];            // host all-reduce slots
  int64_t list_len[kIpcMaxRanks];           // allgatherv: entries in rank p's list file
  // the plan being attached: handle and geometry of rank p's send buffer
  hipIpcMemHandle_t mem[kIpcMaxRanks];
  int64_t nsend[kIpcMaxRanks];
  int64_t span_off[kIpcMaxRanks][kIpcMaxRanks];   // [producer][consumer]
  int64_t span_cnt[kIpcMaxRanks][kIpcMaxRanks];
};

inline size_t ipc_page_round(size_t b) { return (b + 4095) & ~size_t(4095); }

struct IpcTransport : Transport {
  std::string name;
  bool host_only = false;
  int device = 0;
  IpcHeader* hdr = nullptr;
  char* base = nullptr;
  size_t map_bytes = 0, flag_off = 0, flag_bytes = 0;
  char* d_flags = nullptr;           // device view of the flag pages
  bool registered = false, broken = false, creator = false;
  double timeout_s = 300.0;
  hipStream_t cs = nullptr;          // communication stream
  struct PlanState {
    uint64_t seq = 0;                // exchanges of this plan so far (the same on every rank)
    std::vector<real*> peer;         // peer-mapped send buffers of the producers this rank reads
    std::vector<int64_t> peer_nsend, peer_off;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    bool attached = false, on_cs = false;
  };
  std::vector<PlanState> plans;

  ~IpcTransport() override {
    for (PlanState& ps : plans) close_plan(ps);
    if (cs) hipStreamDestroy(cs);
    if (registered) hipHostUnregister(base + flag_off);
    if (base) munmap(base, map_bytes);
    if (creator) shm_unlink(name.c_str());
  }
  bool async() const override { return !staged; }
  // AMGH_IPC_STAGED=1 (the same on every rank): no peer mapping at all — an exchange is packed on the device, copied to the host,
  // carried by the shared-memory rendezvous (exchange_host) and copied back.  Slow and synchronous; what is left when neither
  // RCCL nor hipIpc peer mappings work between the devices of a node (bench_dist.py's last preflight candidate).
  bool staged = false;
  std::vector<real> st_send, st_recv;
  int send_copies() const override { return 2; }

  // ---- flags ----------------------------------------------------------------------------------------------------
  size_t ready_index(int plan, int p) const { return (size_t)plan * nranks + p; }
  size_t done_index(int plan, int consumer, int producer) const {
    return (size_t)kIpcMaxPlans * nranks + ((size_t)plan * nranks + consumer) * nranks + producer;
  }
  std::atomic<uint64_t>* host_flag(size_t idx) const {
    return reinterpret_cast<std::atomic<uint64_t>*>(base + flag_off + idx * kIpcFlagStride);
  }
  void* dev_flag(size_t idx) const { return d_flags + idx * kIpcFlagStride; }
  void release_all_flags() {
    const size_t nflags = flag_bytes / kIpcFlagStride;
    for (size_t i = 0; i < nflags; ++i) host_flag(i)->store(kIpcReleased, std::memory_order_release);
  }
  int give_up() {
    if (hdr) hdr->abort_word.store(1);
    if (base) release_all_flags();
    if (creator) { shm_unlink(name.c_str()); creator = false; }  // the mappings stay, the name goes
    broken = true;
    return AMGH_ESTATE;
  }
  // gone, or exited and not yet reaped by its parent (a zombie still answers kill(pid, 0))
  static bool pid_dead(int32_t pid) {
    if (kill(pid, 0) != 0 && errno == ESRCH) return true;
    char path[64], buf[512];
    snprintf(path, sizeof path, "/proc/%d/stat", (int)pid);
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    const ssize_t n = read(fd, buf, sizeof buf - 1);
    close(fd);
    if (n <= 0) return false;
    buf[n] = 0;
    const char* rp = strrchr(buf, ')');  // "pid (comm) S ..."
    return rp && rp[1] == ' ' && (rp[2] == 'Z' || rp[2] == 'X');
  }
  bool peers_alive() const {
    for (int p = 0; p < nranks; ++p) {
      const int32_t pid = hdr->pid[p];
      if (p != rank && pid > 0 && pid_dead(pid)) return false;
    }
    return true;
  }
  // spin until pred(); false on abort / dead peer / timeout
  template <class F>
  bool wait_for(F pred) {
    const auto t0 = std::chrono::steady_clock::now();
    auto last_check = t0;
    for (int64_t spins = 0;; ++spins) {
      if (pred()) return true;
      if (spins < 2000) continue;
      if (spins < 20000) std::this_thread::yield();
      else usleep(50);
      const auto now = std::chrono::steady_clock::now();
      if (now - last_check > std::chrono::milliseconds(20)) {
        last_check = now;
        if (hdr->abort_word.load() || !peers_alive()) return false;
        if (std::chrono::duration<double>(now - t0).count() > timeout_s) return false;
      }
    }
  }

  // ---- rendezvous -----------------------------------------------------------------------------------------------
  int init(const char* shm_name, int rank_, int nranks_, int device_) {
    rank = rank_; nranks = nranks_; device = device_; host_only = device_ < 0;
    name = shm_name;
    if (const char* ev = getenv("AMGH_IPC_TIMEOUT_S")) timeout_s = std::max(1.0, atof(ev));
    if (const char* ev = getenv("AMGH_IPC_STAGED")) staged = atoi(ev) != 0 && !host_only;
    flag_off = ipc_page_round(sizeof(IpcHeader));
    flag_bytes = ipc_page_round(((size_t)kIpcMaxPlans * nranks + (size_t)kIpcMaxPlans * nranks * nranks) * kIpcFlagStride);
    map_bytes = flag_off + flag_bytes;
    if (rank == 0) {
      shm_unlink(name.c_str());
      const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) return AMGH_ESTATE;
      creator = true;
      if (ftruncate(fd, (off_t)map_bytes) != 0) { close(fd); return AMGH_ENOMEM; }
      base = (char*)mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (base == (char*)MAP_FAILED) { base = nullptr; return AMGH_ENOMEM; }
      hdr = reinterpret_cast<IpcHeader*>(base);
      hdr->nranks = nranks;
      hdr->abort_word.store(0); hdr->bar_count.store(0); hdr->bar_gen.store(0); hdr->attached.store(0);
      for (int p = 0; p < kIpcMaxRanks; ++p) hdr->pid[p] = 0;
      hdr->pid[0] = (int32_t)getpid();
      hdr->magic.store(kIpcMagic, std::memory_order_release);
    } else {
      // until rank 0 has created, sized and initialised the segment; a segment left behind by an earlier run under the
      // same name (its creator is gone, or it already has all its ranks) is not the one: look again
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        const int fd = shm_open(name.c_str(), O_RDWR, 0600);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= map_bytes) {
          char* m = (char*)mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
          close(fd);
          if (m != (char*)MAP_FAILED) {
            IpcHeader* h = reinterpret_cast<IpcHeader*>(m);
            const bool fresh = h->magic.load(std::memory_order_acquire) == kIpcMagic && h->pid[0] > 0 &&
                               !pid_dead(h->pid[0]) && h->attached.load() < h->nranks &&
                               h->pid[rank] == 0 && !h->abort_word.load();
            if (fresh) { base = m; hdr = h; break; }
            munmap(m, map_bytes);
          }
        } else if (fd >= 0) {
          close(fd);
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return AMGH_ESTATE;
        usleep(1000);
      }
      if (hdr->nranks != nranks) return AMGH_EINVAL;
      hdr->pid[rank] = (int32_t)getpid();
    }
    hdr->attached.fetch_add(1);
    if (!wait_for_plain([&] { return hdr->attached.load() >= nranks || hdr->abort_word.load(); })) return give_up();
    if (hdr->abort_word.load()) return give_up();
    if (!host_only) {
      HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      int can = 0;
      HIP_TRY(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device));
      if (!can) return AMGH_EUNSUPPORTED;
      HIP_TRY(hipHostRegister(base + flag_off, flag_bytes, hipHostRegisterMapped | hipHostRegisterPortable));
      registered = true;
      void* dp = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dp, base + flag_off, 0));
      d_flags = (char*)dp;
    }
    plans.resize(kIpcMaxPlans);
    return barrier();
  }
  // before every rank's pid is known only the timeout bounds the wait
  template <class F>
  bool wait_for_plain(F pred) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!pred()) {
      usleep(200);
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    }
    return true;
  }

  int barrier() override {
    if (broken) return AMGH_ESTATE;
    const uint32_t g = hdr->bar_gen.load();
    if (hdr->bar_count.fetch_add(1) + 1 == (uint32_t)nranks) {
      hdr->bar_count.store(0);
      hdr->bar_gen.fetch_add(1);
      return AMGH_OK;
    }
    if (!wait_for([&] { return hdr->bar_gen.load() != g; })) return give_up();
    return AMGH_OK;
  }
  int allreduce(double* v, int n, bool max_op) override {
    if (n > 64) return AMGH_EINVAL;
    if (broken) return AMGH_ESTATE;
    for (int i = 0; i < n; ++i) hdr->vals[rank][i] = v[i];
    RC_TRY(barrier());
    for (int i = 0; i < n; ++i) {
      double acc = hdr->vals[0][i];  // rank order: every rank computes the same bits
      for (int p = 1; p < nranks; ++p) acc = max_op ? std::max(acc, hdr->vals[p][i]) : acc + hdr->vals[p][i];
      v[i] = acc;
    }
    return barrier();  // nobody overwrites a slot somebody still reads
  }
  std::string list_name(int p) const { return name + ".l" + std::to_string(p); }
  int allgatherv_host(const std::vector<int64_t>& mine, std::vector<std::vector<int64_t>>& all) override {
    if (broken) return AMGH_ESTATE;
    const std::string my = list_name(rank);
    {
      shm_unlink(my.c_str());
      const int fd = shm_open(my.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) return give_up();
      const char* src = (const char*)mine.data();
      size_t left = mine.size() * sizeof(int64_t);
      while (left > 0) {
        const ssize_t w = write(fd, src, left);
        if (w <= 0) { close(fd); shm_unlink(my.c_str()); return give_up(); }
        src += w; left -= (size_t)w;
      }
      close(fd);
      hdr->list_len[rank] = (int64_t)mine.size();
    }
    int rc = barrier();
    all.assign(nranks, {});
    for (int p = 0; p < nranks && rc == AMGH_OK; ++p) {
      if (p == rank) { all[p] = mine; continue; }
      const int64_t len = hdr->list_len[p];
      all[p].resize((size_t)len);
      if (len == 0) continue;
      const int fd = shm_open(list_name(p).c_str(), O_RDONLY, 0600);
      if (fd < 0) { rc = give_up(); break; }
      char* dst = (char*)all[p].data();
      size_t left = (size_t)len * sizeof(int64_t);
      while (left > 0) {
        const ssize_t r = read(fd, dst, left);
        if (r <= 0) { rc = give_up(); break; }
        dst += r; left -= (size_t)r;
      }
      close(fd);
    }
    const int rc2 = rc == AMGH_OK ? barrier() : rc;
    shm_unlink(my.c_str());
    return rc2;
  }

  // ---- plans ----------------------------------------------------------------------------------------------------
  void close_plan(PlanState& ps) {
    for (real* q : ps.peer)
      if (q) hipIpcCloseMemHandle(q);
    ps.peer.clear();
    if (ps.ev_ready) hipEventDestroy(ps.ev_ready);
    if (ps.ev_done) hipEventDestroy(ps.ev_done);
    ps.ev_ready = ps.ev_done = nullptr;
    ps.attached = false;
  }
  int plan_attach(VecPlan& pl) override {
    if (broken) return AMGH_ESTATE;
    if (pl.id < 0 || pl.id >= kIpcMaxPlans) return AMGH_EUNSUPPORTED;
    PlanState& ps = plans[pl.id];
    ps.seq = 0;
    if (staged) { ps.attached = true; return barrier(); }
    ps.peer.assign(nranks, nullptr);
    ps.peer_nsend.assign(nranks, 0);
    ps.peer_off.assign(nranks, 0);
    hdr->nsend[rank] = pl.nsend;
    for (int q = 0; q < nranks; ++q) { hdr->span_off[rank][q] = pl.send[q].off; hdr->span_cnt[rank][q] = pl.send[q].cnt; }
    int rc = AMGH_OK;
    if (pl.nsend > 0) {
      const hipError_t e = hipIpcGetMemHandle(&hdr->mem[rank], pl.d_sendbuf);
      if (e != hipSuccess) rc = -(1000 + (int)e);
    }
    if (rc == AMGH_OK) rc = ev_create(&ps.ev_ready);
    if (rc == AMGH_OK) rc = ev_create(&ps.ev_done);
    if (rc != AMGH_OK) { give_up(); return rc; }
    // (the flags this rank writes for the plan start from zero with its sequence numbers: a plan id attached a second time
    // on the same segment would otherwise pass its waits on the flags of its first life)
    host_flag(ready_index(pl.id, rank))->store(0, std::memory_order_release);
    for (int p = 0; p < nranks; ++p) host_flag(done_index(pl.id, rank, p))->store(0, std::memory_order_release);
    RC_TRY(barrier());
    for (int p = 0; p < nranks; ++p) {
      if (p == rank || pl.recv[p].cnt <= 0) continue;
      if (hdr->span_cnt[p][rank] != pl.recv[p].cnt) { give_up(); return AMGH_ESTATE; }  // the two sides disagree
      void* q = nullptr;
      const hipError_t e = hipIpcOpenMemHandle(&q, hdr->mem[p], hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) { give_up(); return -(1000 + (int)e); }
      ps.peer[p] = (real*)q;
      ps.peer_nsend[p] = hdr->nsend[p];
      ps.peer_off[p] = hdr->span_off[p][rank];
    }
    ps.attached = true;
    return barrier();  // the header slots are free for the next plan
  }
  void plan_detach(VecPlan& pl) override {
    if (pl.id >= 0 && pl.id < (int)plans.size() && plans[pl.id].attached) close_plan(plans[pl.id]);
  }
  static int ev_create(hipEvent_t* ev) {
    HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return AMGH_OK;
  }

  // ---- data path ------------------------------------------------------------------------------------------------
  int pack_target(VecPlan& pl, hipStream_t st, real** sendbuf) override {
    if (broken) return AMGH_ESTATE;
    PlanState& ps = plans[pl.id];
    if (staged) { *sendbuf = pl.d_sendbuf; return AMGH_OK; }
    const uint64_t k = ++ps.seq;
    *sendbuf = pl.d_sendbuf + (k & 1) * pl.nsend;
    if (k > 2)  // the readers of exchange k - 2 are done with this copy of the buffer
      for (int q = 0; q < nranks; ++q)
        if (q != rank && pl.send[q].cnt > 0)
          HIP_TRY(hipStreamWaitValue64(st, dev_flag(done_index(pl.id, q, rank)), k - 2, hipStreamWaitValueGte, ~0ull));
    return AMGH_OK;
  }
  int exchange_begin(VecPlan& pl, const real* sendbuf, real* recvbuf, hipStream_t st, bool overlap) override {
    if (broken) return AMGH_ESTATE;
    if (staged) {
      HIP_TRY(hipStreamSynchronize(st));
      st_send.resize((size_t)std::max<int64_t>(1, pl.nsend)); st_recv.assign((size_t)std::max<int64_t>(1, pl.nhalo()), (real)0);
      if (pl.nsend > 0) HIP_TRY(hipMemcpy(st_send.data(), sendbuf, sizeof(real) * (size_t)pl.nsend, hipMemcpyDeviceToHost));
      RC_TRY(exchange_host(pl, st_send.data(), st_recv.data()));
      if (pl.nhalo() > 0) HIP_TRY(hipMemcpy(recvbuf, st_recv.data(), sizeof(real) * (size_t)pl.nhalo(), hipMemcpyHostToDevice));
      plans[pl.id].on_cs = false;
      return AMGH_OK;
    }
    PlanState& ps = plans[pl.id];
    const uint64_t k = ps.seq;
    if (pl.nsend > 0) HIP_TRY(hipStreamWriteValue64(st, dev_flag(ready_index(pl.id, rank)), k, 0));
    ps.on_cs = overlap;
    hipStream_t ts = st;
    if (overlap) {   // the halo region may still be read by what `st` ran before this exchange
      HIP_TRY(hipEventRecord(ps.ev_ready, st));
      HIP_TRY(hipStreamWaitEvent(cs, ps.ev_ready, 0));
      ts = cs;
    }
    for (int p = 0; p < nranks; ++p) {
      if (p == rank || pl.recv[p].cnt <= 0) continue;
      HIP_TRY(hipStreamWaitValue64(ts, dev_flag(ready_index(pl.id, p)), k, hipStreamWaitValueGte, ~0ull));
      const real* src = ps.peer[p] + (k & 1) * ps.peer_nsend[p] + ps.peer_off[p];
      // a copy KERNEL pulls the entries out of the peer-mapped buffer: hipMemcpyAsync from IPC-mapped memory costs an
      // order of magnitude more per call (tools/ipc_pingpong.hip: 200 vs 28 us per round trip at 512 KiB)
      hipLaunchKernelGGL(copy_kernel, dim3(grid_for(pl.recv[p].cnt)), dim3(256), 0, ts, recvbuf + pl.recv[p].off, src,
                         (int64_t)pl.recv[p].cnt);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamWriteValue64(ts, dev_flag(done_index(pl.id, rank, p)), k, 0));
    }
    if (overlap) HIP_TRY(hipEventRecord(ps.ev_done, cs));
    return AMGH_OK;
  }
  int exchange_finish(VecPlan& pl, hipStream_t st) override {
    if (plans[pl.id].on_cs) HIP_TRY(hipStreamWaitEvent(st, plans[pl.id].ev_done, 0));
    return AMGH_OK;
  }
  // a stream of this rank may wait for a flag only another process writes: bounded, and released when a peer is gone
  int wait_stream(hipStream_t st) override {
    hipError_t q = hipSuccess;
    const bool ok = wait_for([&] { q = hipStreamQuery(st); return q != hipErrorNotReady; });
    if (!ok) {
      give_up();
      hipStreamSynchronize(st);  // every flag is released: the stream drains
      return AMGH_ESTATE;
    }
    if (q != hipSuccess) return -(1000 + (int)q);
    return broken ? AMGH_ESTATE : AMGH_OK;
  }
};

}  // namespace

