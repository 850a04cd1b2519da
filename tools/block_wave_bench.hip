// block_wave_bench — the block-wavefront Gauss-Seidel sweep of csrc/hip/gs_blocks.hpp: plan (partition by
// monotone potentials), layout, forward / backward sweeps checked bit for bit against the scalar loops, timings — as one
// launch per depth of the block graph (gs_bw_packed_kernel, with in-kernel clocks per launch) and as ONE launch per sweep
// with the blocks chained by flags (gs_bw_chain_kernel: stamps per block and per depth, 20 alternating sweeps against the
// launched ones, poll give-ups).  BW_PLAN_TIMING=1: laps of the plan.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/block_wave_bench tools/block_wave_bench.hip
// usage: block_wave_bench poisson N [target_rows]     (7-point, N^3)   |   block_wave_bench poisson2 N [target_rows]   (5-point, N^2)
//        block_wave_bench file PATH [target_rows]    PATH: int64 n, int64 nnz, int32 rowptr[n+1], int32 col[nnz], double val[nnz]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define BW_RELAY_ALL_W 1
#include "../algebraicmultigrid.jl_amd/csrc/hip/gs_relay.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace amgh;

struct Csr { int64_t n = 0; std::vector<int32_t> rp, ci; std::vector<double> va; };
static Csr poisson3(int N) {
  Csr A; A.n = (int64_t)N * N * N; A.rp.assign(A.n + 1, 0);
  A.ci.reserve(7 * A.n); A.va.reserve(7 * A.n);
  for (int k = 0; k < N; ++k) for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) {
    const int64_t r = i + (int64_t)N * (j + (int64_t)N * k);
    if (k > 0) { A.ci.push_back((int32_t)(r - (int64_t)N * N)); A.va.push_back(-1.0); }
    if (j > 0) { A.ci.push_back((int32_t)(r - N)); A.va.push_back(-1.0); }
    if (i > 0) { A.ci.push_back((int32_t)(r - 1)); A.va.push_back(-1.0); }
    A.ci.push_back((int32_t)r); A.va.push_back(6.0);
    if (i < N - 1) { A.ci.push_back((int32_t)(r + 1)); A.va.push_back(-1.0); }
    if (j < N - 1) { A.ci.push_back((int32_t)(r + N)); A.va.push_back(-1.0); }
    if (k < N - 1) { A.ci.push_back((int32_t)(r + (int64_t)N * N)); A.va.push_back(-1.0); }
    A.rp[r + 1] = (int32_t)A.ci.size();
  }
  return A;
}
static Csr poisson2(int N) {
  Csr A; A.n = (int64_t)N * N; A.rp.assign(A.n + 1, 0);
  for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) {
    const int64_t r = i + (int64_t)N * j;
    if (j > 0) { A.ci.push_back((int32_t)(r - N)); A.va.push_back(-1.0); }
    if (i > 0) { A.ci.push_back((int32_t)(r - 1)); A.va.push_back(-1.0); }
    A.ci.push_back((int32_t)r); A.va.push_back(4.0);
    if (i < N - 1) { A.ci.push_back((int32_t)(r + 1)); A.va.push_back(-1.0); }
    if (j < N - 1) { A.ci.push_back((int32_t)(r + N)); A.va.push_back(-1.0); }
    A.rp[r + 1] = (int32_t)A.ci.size();
  }
  return A;
}
static Csr load(const char* path) {
  Csr A; FILE* f = fopen(path, "rb"); if (!f) { printf("cannot open %s\n", path); exit(1); }
  int64_t hdr[2]; if (fread(hdr, 8, 2, f) != 2) exit(1);
  A.n = hdr[0]; A.rp.resize(A.n + 1); A.ci.resize(hdr[1]); A.va.resize(hdr[1]);
  if (fread(A.rp.data(), 4, A.n + 1, f) != (size_t)A.n + 1 || fread(A.ci.data(), 4, hdr[1], f) != (size_t)hdr[1] ||
      fread(A.va.data(), 8, hdr[1], f) != (size_t)hdr[1]) exit(1);
  fclose(f); return A;
}

int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: block_wave_bench poisson N | file PATH  [target_rows]\n"); return 1; }
  Csr A = std::string(argv[1]) == "poisson" ? poisson3(atoi(argv[2])) : std::string(argv[1]) == "poisson2" ? poisson2(atoi(argv[2])) : load(argv[2]);
  const int64_t n = A.n;
  bw::Params prm; if (argc > 3) prm.target_rows = atoi(argv[3]);
  prm.threads = 8;
  bw::Plan P;
  const auto t0 = std::chrono::steady_clock::now();
  const bool ok = bw::plan<double>(n, A.rp.data(), A.ci.data(), A.va.data(), prm, &P);
  const double tplan = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!ok) { printf("plan: not eligible\n"); return 0; }
  const int nl = (int)P.launch_ptr.size() - 1;
  int maxb = 0, under = 0;
  for (int l = 0; l < nl; ++l) { const int c = P.launch_ptr[l + 1] - P.launch_ptr[l]; maxb = std::max(maxb, c); under += c < 256; }
  std::vector<int> rows; for (auto& d : P.blocks) rows.push_back(d.nrows); std::sort(rows.begin(), rows.end());
  printf("n = %lld nnz = %lld: plan %.2f s; offset cuts %d %d, potential ranges %d %d %d, cells %d x %d x %d -> %zu blocks (rows: min %d median %d max %d), maxk %d\n",
         (long long)n, (long long)P.nnz, tplan, P.cuts[0], P.cuts[1], P.range[0], P.range[1], P.range[2], P.cells[0], P.cells[1], P.cells[2], P.blocks.size(),
         rows.front(), rows[rows.size() / 2], rows.back(), P.blocks[0].maxk);
  printf("launches %d (dependency levels %d), up to %d blocks, %d launches under 256 blocks; sum of deepest blocks %lld levels; external x per row %.3f; LDS %.1f KB; record bytes %.3f GB; model %.3f ms per sweep\n",
         nl, P.nlevels, maxb, under, (long long)P.sum_depth, (double)P.ext_total / n, P.lds_max / 1024.0, P.rec.size() / 1e9, P.est_seconds * 1e3);
  // data
  std::vector<double> b(n), x0(n), xb(n), bb(n);
  for (int64_t i = 0; i < n; ++i) { b[i] = std::sin(0.37 * (double)(i % 1000)) + 0.5; x0[i] = std::cos(0.11 * (double)(i % 777)); }
  for (int64_t p = 0; p < n; ++p) { xb[p] = x0[P.perm[p]]; bb[p] = b[P.perm[p]]; }
  std::vector<double> xr = x0;
  auto scalar = [&](bool bwd) {
    for (int64_t s = 0; s < n; ++s) {
      const int64_t i = bwd ? n - 1 - s : s;
      double acc = 0.0, dg = 0.0;
      for (int32_t j = A.rp[i]; j < A.rp[i + 1]; ++j) { if (A.ci[j] == i) dg = A.va[j]; else acc += A.va[j] * xr[A.ci[j]]; }
      if (dg != 0.0) xr[i] = (b[i] - acc) / dg;
    }
  };
  bw::Desc* d_blocks; unsigned char* d_rec; int32_t* d_ext; double *d_b, *d_x;
  CHECK(hipMalloc(&d_blocks, sizeof(bw::Desc) * P.blocks.size())); CHECK(hipMalloc(&d_rec, P.rec.size())); CHECK(hipMalloc(&d_ext, 4 * std::max<size_t>(1, P.ext_col.size())));
  CHECK(hipMalloc(&d_b, 8 * n)); CHECK(hipMalloc(&d_x, 8 * n));
  CHECK(hipMemcpy(d_blocks, P.blocks.data(), sizeof(bw::Desc) * P.blocks.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_rec, P.rec.data(), P.rec.size(), hipMemcpyHostToDevice));
  CHECK(hipMemset(d_ext, 0, 4 * std::max<size_t>(1, P.ext_col.size()))); CHECK(hipMemcpy(d_ext, P.ext_col.data(), 4 * P.ext_col.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_b, bb.data(), 8 * n, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
  bw::Args<double> a{d_blocks, d_rec, d_ext, d_b, d_x, n, n, 1.0, 0, nullptr};
  hipStream_t st; CHECK(hipStreamCreate(&st));
  CHECK(bw::sweep<double>(a, P.blocks[0].maxk, P.launch_ptr, P.lds_max, false, false, 1, st));
  CHECK(bw::sweep<double>(a, P.blocks[0].maxk, P.launch_ptr, P.lds_max, false, true, 1, st));
  CHECK(hipStreamSynchronize(st));
  std::vector<double> xg(n);
  CHECK(hipMemcpy(xg.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
  scalar(false); scalar(true);
  int64_t diff = 0; double maxd = 0.0;
  for (int64_t p = 0; p < n; ++p) { const double e = std::fabs(xg[p] - xr[P.perm[p]]); if (xg[p] != xr[P.perm[p]]) ++diff; maxd = std::max(maxd, e); }
  printf("forward + backward vs the scalar loops: %lld values differ (max |diff| %.3e)\n", (long long)diff, maxd);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double bytes = (double)P.rec.size() + n * 24.0 + P.ext_total * 12.0;
  for (int bwd = 0; bwd < 2; ++bwd) {
    CHECK(bw::sweep<double>(a, P.blocks[0].maxk, P.launch_ptr, P.lds_max, false, bwd, 1, st)); CHECK(hipStreamSynchronize(st));
    const int reps = 5;
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(bw::sweep<double>(a, P.blocks[0].maxk, P.launch_ptr, P.lds_max, false, bwd, 1, st));
    CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s sweep: %.3f ms = %d launches x %.2f us (%.0f GB/s)\n", bwd ? "backward" : "forward ", ms / reps, nl, ms / reps / nl * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
  }
  // where a launch's time goes: wall-clock stamps (100 MHz) of every block of one forward sweep
  {
    long long* d_tim; CHECK(hipMalloc(&d_tim, 32 * P.blocks.size()));
    a.tim = d_tim;
    CHECK(bw::sweep<double>(a, P.blocks[0].maxk, P.launch_ptr, P.lds_max, false, false, 1, st)); CHECK(hipStreamSynchronize(st));
    std::vector<long long> tim(4 * P.blocks.size());
    CHECK(hipMemcpy(tim.data(), d_tim, 32 * P.blocks.size(), hipMemcpyDeviceToHost));
    double s_load = 0, s_sweep = 0, s_store = 0, s_span = 0, s_gap = 0; long long prev_end = 0; int cnt = 0;
    printf("launch: blocks | first start -> last end (span), mean load / sweep / write-back per block, gap to the previous launch [us]\n");
    for (int l = 0; l < nl; ++l) {
      long long t0 = -1, t3 = 0; double ld = 0, sw = 0, wb = 0; const int nb = P.launch_ptr[l + 1] - P.launch_ptr[l];
      for (int b = P.launch_ptr[l]; b < P.launch_ptr[l + 1]; ++b) {
        const long long* t = &tim[4 * (size_t)b];
        if (t0 < 0 || t[0] < t0) t0 = t[0];
        t3 = std::max(t3, t[3]); ld += (t[1] - t[0]) * 0.01; sw += (t[2] - t[1]) * 0.01; wb += (t[3] - t[2]) * 0.01;
      }
      const double span = (t3 - t0) * 0.01, gap = l ? (t0 - prev_end) * 0.01 : 0.0;
      if (l % 8 == 0 || l == nl - 1) printf("  %3d: %4d | span %6.2f  load %6.2f  sweep %6.2f  write-back %5.2f  gap %5.2f\n", l, nb, span, ld / nb, sw / nb, wb / nb, gap);
      s_load += ld / nb; s_sweep += sw / nb; s_store += wb / nb; s_span += span; s_gap += gap; prev_end = t3; ++cnt;
    }
    printf("sum over %d launches: spans %.1f us, gaps %.1f us; mean per launch: load %.2f, sweep %.2f, write-back %.2f us\n", cnt, s_span, s_gap,
           s_load / cnt, s_sweep / cnt, s_store / cnt);
  }
  // ---- the same sweeps as ONE launch each: blocks chained by flags (gs_bw_chain_kernel) ----
  {
    const int32_t B = (int32_t)P.blocks.size();
    int32_t *d_dp, *d_d, *d_sp, *d_s, *d_err; unsigned int* d_flags; unsigned long long* d_head; long long* d_tim;
    CHECK(hipMalloc(&d_dp, 4 * (B + 1))); CHECK(hipMalloc(&d_sp, 4 * (B + 1)));
    CHECK(hipMalloc(&d_d, 4 * std::max<size_t>(1, P.dep.size()))); CHECK(hipMalloc(&d_s, 4 * std::max<size_t>(1, P.sdep.size())));
    CHECK(hipMalloc(&d_flags, 4 * (size_t)B)); CHECK(hipMalloc(&d_head, 8)); CHECK(hipMalloc(&d_err, 4)); CHECK(hipMalloc(&d_tim, 40 * (size_t)B));
    CHECK(hipMemcpy(d_dp, P.dep_ptr.data(), 4 * (B + 1), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_sp, P.sdep_ptr.data(), 4 * (B + 1), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_d, P.dep.data(), 4 * P.dep.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_s, P.sdep.data(), 4 * P.sdep.size(), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_flags, 0, 4 * (size_t)B)); CHECK(hipMemset(d_head, 0, 8)); CHECK(hipMemset(d_err, 0, 4));
    a.tim = nullptr;
    bw::ChainArgs<double> cf{a, d_dp, d_d, d_flags, d_head, B, d_err}, cb{a, d_sp, d_s, d_flags, d_head, B, d_err};
    int maxdep = 0; for (int32_t b = 0; b < B; ++b) maxdep = std::max(maxdep, P.dep_ptr[b + 1] - P.dep_ptr[b]);
    printf("== chained: one launch per sweep, %d blocks, %.2f dependencies per block (max %d)\n", B, (double)P.dep.size() / B, maxdep);
    CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
    CHECK(bw::sweep_chain<double>(cf, P.blocks[0].maxk, P.lds_max, false, false, st));
    CHECK(bw::sweep_chain<double>(cb, P.blocks[0].maxk, P.lds_max, false, true, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(xg.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
    int err = 0; CHECK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    xr = x0; scalar(false); scalar(true);
    diff = 0; maxd = 0.0;
    for (int64_t p = 0; p < n; ++p) { const double e = std::fabs(xg[p] - xr[P.perm[p]]); if (xg[p] != xr[P.perm[p]]) ++diff; maxd = std::max(maxd, e); }
    printf("chained forward + backward vs the scalar loops: %lld values differ (max |diff| %.3e), poll give-ups %d\n", (long long)diff, maxd, err);
    for (int bwd = 0; bwd < 2; ++bwd) {
      const bw::ChainArgs<double>& c = bwd ? cb : cf;
      CHECK(bw::sweep_chain<double>(c, P.blocks[0].maxk, P.lds_max, false, bwd, st)); CHECK(hipStreamSynchronize(st));
      const int reps = 5;
      CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) CHECK(bw::sweep_chain<double>(c, P.blocks[0].maxk, P.lds_max, false, bwd, st));
      CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("chained %s sweep: %.3f ms (%.0f GB/s)\n", bwd ? "backward" : "forward ", ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
    }
    // repeated sweeps keep agreeing with themselves (epochs, stale lines): 20 forward sweeps chained vs launched
    {
      std::vector<double> xc(n), xl2(n);
      CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
      for (int r = 0; r < 20; ++r) CHECK(bw::sweep_chain<double>(r & 1 ? cb : cf, P.blocks[0].maxk, P.lds_max, false, r & 1, st));
      CHECK(hipStreamSynchronize(st)); CHECK(hipMemcpy(xc.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
      for (int r = 0; r < 20; ++r) CHECK(bw::sweep<double>(a, P.blocks[0].maxk, P.launch_ptr, P.lds_max, false, r & 1, 1, st));
      CHECK(hipStreamSynchronize(st)); CHECK(hipMemcpy(xl2.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
      int64_t dd = 0; for (int64_t p = 0; p < n; ++p) dd += xc[p] != xl2[p];
      CHECK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
      printf("20 alternating sweeps, chained vs launched: %lld values differ, poll give-ups %d\n", (long long)dd, err);
    }
    cf.a.tim = d_tim;
    CHECK(bw::sweep_chain<double>(cf, P.blocks[0].maxk, P.lds_max, false, false, st)); CHECK(hipStreamSynchronize(st));
    std::vector<long long> tim(5 * (size_t)B);
    CHECK(hipMemcpy(tim.data(), d_tim, 40 * (size_t)B, hipMemcpyDeviceToHost));
    long long tmin = tim[0], tmax = 0; double ld = 0, wt = 0, wk = 0, wb = 0;
    for (int32_t b = 0; b < B; ++b) {
      const long long* t = &tim[5 * (size_t)b];
      tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[4]);
      ld += (t[1] - t[0]) * 0.01; wt += (t[2] - t[1]) * 0.01; wk += (t[3] - t[2]) * 0.01; wb += (t[4] - t[3]) * 0.01;
    }
    printf("chained forward sweep, stamps: first start -> last end %.1f us; mean per block: load %.2f, wait + fetch %.2f, walk %.2f, write-through + flag %.2f us\n",
           (tmax - tmin) * 0.01, ld / B, wt / B, wk / B, wb / B);
    // the critical path as the stamps show it: per launch depth, the latest end
    for (int l = 0; l < nl; l += std::max(1, nl / 12)) {
      long long te = 0, ts = -1; for (int b = P.launch_ptr[l]; b < P.launch_ptr[l + 1]; ++b) { te = std::max(te, tim[5 * (size_t)b + 4]); if (ts < 0 || tim[5 * (size_t)b] < ts) ts = tim[5 * (size_t)b]; }
      printf("  depth %3d: %4d blocks, first start %8.1f us, last end %8.1f us\n", l, P.launch_ptr[l + 1] - P.launch_ptr[l], (ts - tmin) * 0.01, (te - tmin) * 0.01);
    }
  }
  // ---- the same sweeps as a DATAFLOW (gs_flow.hpp): rows published as they are computed, records streamed into registers ----
  if (!getenv("BW_NO_FLOW")) {
    const auto tf0 = std::chrono::steady_clock::now();
    const bool sym = bw::structurally_symmetric(n, A.rp.data(), A.ci.data(), 8);
    bw::Flow F;
    const bool fok = bw::flow_build<double>(P, 8, &F);
    const double tflow = std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
    printf("== dataflow: structurally symmetric %d, flow build %.2f s (ok %d), %lld mailboxes (%.1f MB), LDS %.1f KB per block, depth %d\n", (int)sym, tflow, (int)fok,
           (long long)F.nmail, F.nmail * 16e-6, F.lds_max / 1024.0, BW_FLOW_DEPTH);
    if (sym && fok) {
      const int32_t B = (int32_t)P.blocks.size();
      bw::FlowDesc* d_fd; unsigned char* d_srec; uint32_t* d_aux; uint16_t* d_fs; int32_t *d_fm, *d_err; void* d_mbox; unsigned long long* d_head; long long* d_tim;
      CHECK(hipMalloc(&d_fd, sizeof(bw::FlowDesc) * (size_t)B)); CHECK(hipMalloc(&d_srec, F.srec.size())); CHECK(hipMalloc(&d_aux, 4 * std::max<size_t>(1, F.aux.size())));
      CHECK(hipMalloc(&d_fs, 2 * F.fl_slot.size())); CHECK(hipMalloc(&d_fm, 4 * F.fl_mb.size()));
      CHECK(hipMalloc(&d_mbox, 16 * (size_t)(F.nmail + 1024))); CHECK(hipMalloc(&d_head, 8)); CHECK(hipMalloc(&d_err, 4)); const size_t tim_bytes = (32 + 8 * 384 + 32) * (size_t)B + 16 * (size_t)(F.nmail + 1024);
      CHECK(hipMalloc(&d_tim, tim_bytes)); CHECK(hipMemset(d_tim, 0, tim_bytes));
      CHECK(hipMemcpy(d_fd, F.fd.data(), sizeof(bw::FlowDesc) * (size_t)B, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_srec, F.srec.data(), F.srec.size(), hipMemcpyHostToDevice));
      CHECK(hipMemcpy(d_aux, F.aux.data(), 4 * F.aux.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_fs, F.fl_slot.data(), 2 * F.fl_slot.size(), hipMemcpyHostToDevice));
      CHECK(hipMemcpy(d_fm, F.fl_mb.data(), 4 * F.fl_mb.size(), hipMemcpyHostToDevice));
      CHECK(hipMemset(d_mbox, 0, 16 * (size_t)(F.nmail + 1024))); CHECK(hipMemset(d_head, 0, 8)); CHECK(hipMemset(d_err, 0, 4));
      bw::FlowArgs<double> fa{d_blocks, d_fd, d_srec, d_aux, d_ext, d_fm, d_fs, d_mbox, d_b, d_x, 1.0, d_head, B, (int32_t)F.nmail, d_err, nullptr, 0u, -1};
      const int mk = P.blocks[0].maxk;
      CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
      CHECK(bw::sweep_flow<double>(fa, mk, F.lds_max, false, false, st));
      CHECK(bw::sweep_flow<double>(fa, mk, F.lds_max, false, true, st));
      CHECK(hipStreamSynchronize(st));
      CHECK(hipMemcpy(xg.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
      int err = 0; CHECK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
      xr = x0; scalar(false); scalar(true);
      diff = 0; maxd = 0.0;
      for (int64_t p = 0; p < n; ++p) { const double e = std::fabs(xg[p] - xr[P.perm[p]]); if (xg[p] != xr[P.perm[p]]) ++diff; maxd = std::max(maxd, e); }
      printf("dataflow forward + backward vs the scalar loops: %lld values differ (max |diff| %.3e), give-ups %d\n", (long long)diff, maxd, err);
      const double fbytes = (double)F.srec.size() + n * 28.0 + P.ext_total * 22.0;
      for (int bwd = 0; bwd < 2; ++bwd) {
        CHECK(bw::sweep_flow<double>(fa, mk, F.lds_max, false, bwd, st)); CHECK(hipStreamSynchronize(st));
        const int reps = 5;
        CHECK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CHECK(bw::sweep_flow<double>(fa, mk, F.lds_max, false, bwd, st));
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("dataflow %s sweep: %.3f ms (%.0f GB/s)\n", bwd ? "backward" : "forward ", ms / reps, fbytes / (ms / reps * 1e-3) / 1e9);
      }
      {   // repeated alternating sweeps against the launched ones
        std::vector<double> xc(n), xl2(n);
        CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
        for (int r = 0; r < 20; ++r) CHECK(bw::sweep_flow<double>(fa, mk, F.lds_max, false, r & 1, st));
        CHECK(hipStreamSynchronize(st)); CHECK(hipMemcpy(xc.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
        a.tim = nullptr;
        for (int r = 0; r < 20; ++r) CHECK(bw::sweep<double>(a, mk, P.launch_ptr, P.lds_max, false, r & 1, 1, st));
        CHECK(hipStreamSynchronize(st)); CHECK(hipMemcpy(xl2.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
        int64_t dd = 0; for (int64_t p = 0; p < n; ++p) dd += xc[p] != xl2[p];
        CHECK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        printf("20 alternating sweeps, dataflow vs launched: %lld values differ, give-ups %d\n", (long long)dd, err);
      }
      fa.tim = d_tim;
      for (int bwd = 0; bwd < 2; ++bwd) {
        CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
        if (bwd) { fa.tim = nullptr; CHECK(bw::sweep_flow<double>(fa, mk, F.lds_max, false, false, st)); fa.tim = d_tim; }
        CHECK(bw::sweep_flow<double>(fa, mk, F.lds_max, false, bwd, st)); CHECK(hipStreamSynchronize(st));
        std::vector<long long> tim(4 * (size_t)B);
        CHECK(hipMemcpy(tim.data(), d_tim, 32 * (size_t)B, hipMemcpyDeviceToHost));
        long long tmin = tim[0], tmax = 0; double ld = 0, wk = 0;
        for (int32_t bq = 0; bq < B; ++bq) {
          const long long* t = &tim[4 * (size_t)bq];
          tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[2]);
          ld += (t[1] - t[0]) * 0.01; wk += (t[2] - t[1]) * 0.01;
        }
        printf("dataflow %s sweep, stamps: first start -> last end %.1f us; mean per block: ticket + load %.2f, walk (with its waits) %.2f us\n", bwd ? "backward" : "forward", (tmax - tmin) * 0.01, ld / B, wk / B);
        for (int l0 = 0; l0 < nl; l0 += std::max(1, nl / 12)) {
          const int l = bwd ? nl - 1 - l0 : l0;
          long long te = 0, ts = -1, tw = -1; double w = 0; const int nbk = P.launch_ptr[l + 1] - P.launch_ptr[l];
          for (int bq = P.launch_ptr[l]; bq < P.launch_ptr[l + 1]; ++bq) {
            te = std::max(te, tim[4 * (size_t)bq + 2]); if (ts < 0 || tim[4 * (size_t)bq] < ts) ts = tim[4 * (size_t)bq];
            if (tw < 0 || tim[4 * (size_t)bq + 1] < tw) tw = tim[4 * (size_t)bq + 1];
            w += (tim[4 * (size_t)bq + 2] - tim[4 * (size_t)bq + 1]) * 0.01;
          }
          printf("  depth %3d: %4d blocks, first start %8.1f us, first walk %8.1f us, last end %8.1f us, mean walk %6.2f us\n", l, nbk, (ts - tmin) * 0.01, (tw - tmin) * 0.01, (te - tmin) * 0.01, w / nbk);
        }
      }
      // ---- the same dataflow with the walk of a block RELAYED between W waves (gs_relay.hpp) ----
      if (!getenv("BW_NO_RELAY")) {
        std::vector<double> xq(n);
        if (const char* ep = getenv("BW_RELAY_LDS_PAD")) { bw::relay_lds_pad() = (size_t)atoi(ep) * 1024; printf("relay: %d KB of LDS padding per workgroup\n", atoi(ep)); }
        if (const char* eg = getenv("BW_RELAY_GRID")) { fa.grid = atoi(eg); printf("relay: persistent launches of %d workgroups (resident capacity at W = 3: %d)\n", fa.grid, bw::relay_resident_blocks<double>(mk, F.lds_max)); }
        for (int W : {2, 3, 4}) {
          if (const char* ew = getenv("BW_RELAY_ONLY")) if (atoi(ew) != W) continue;
          fa.tim = nullptr;
          fa.spin_limit = 1u << 18;   // (a protocol error of a kernel under development ends in seconds)
          CHECK(hipMemset(d_err, 0, 4));
          CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
          CHECK(bw::sweep_relay<double>(fa, mk, F.lds_max, false, false, st, W));
          CHECK(bw::sweep_relay<double>(fa, mk, F.lds_max, false, true, st, W));
          CHECK(hipStreamSynchronize(st));
          CHECK(hipMemcpy(xq.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
          CHECK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
          int64_t dd = 0; double md = 0.0;
          for (int64_t p = 0; p < n; ++p) { if (xq[p] != xg[p]) ++dd; md = std::max(md, std::fabs(xq[p] - xg[p])); }
          printf("== relay, %d walker waves per block: forward + backward vs the dataflow sweeps (= the scalar loops): %lld values differ (max |diff| %.3e), give-ups %d\n", W, (long long)dd, md, err);
          float msd[2];
          for (int bwd = 0; bwd < 2; ++bwd) {
            CHECK(bw::sweep_relay<double>(fa, mk, F.lds_max, false, bwd, st, W)); CHECK(hipStreamSynchronize(st));
            const int reps = 5;
            CHECK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) CHECK(bw::sweep_relay<double>(fa, mk, F.lds_max, false, bwd, st, W));
            CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&msd[bwd], e0, e1)); msd[bwd] /= reps;
          }
          printf("relay W = %d forward  sweep: %.3f ms (%.0f GB/s)\nrelay W = %d backward sweep: %.3f ms (%.0f GB/s)\n", W, msd[0], fbytes / (msd[0] * 1e-3) / 1e9, W, msd[1], fbytes / (msd[1] * 1e-3) / 1e9);
          {   // 20 alternating sweeps against the single-walker dataflow
            std::vector<double> xc(n), xl2(n);
            CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
            for (int r = 0; r < 20; ++r) CHECK(bw::sweep_relay<double>(fa, mk, F.lds_max, false, r & 1, st, W));
            CHECK(hipStreamSynchronize(st)); CHECK(hipMemcpy(xc.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
            for (int r = 0; r < 20; ++r) CHECK(bw::sweep_flow<double>(fa, mk, F.lds_max, false, r & 1, st));
            CHECK(hipStreamSynchronize(st)); CHECK(hipMemcpy(xl2.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
            int64_t d2 = 0; for (int64_t p = 0; p < n; ++p) d2 += xc[p] != xl2[p];
            CHECK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
            printf("20 alternating sweeps, relay W = %d vs dataflow: %lld values differ, give-ups %d\n", W, (long long)d2, err);
          }
          fa.tim = d_tim;
          for (int bwd = 0; bwd < 1; ++bwd) {
            CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
            CHECK(bw::sweep_relay<double>(fa, mk, F.lds_max, false, bwd, st, W)); CHECK(hipStreamSynchronize(st));
            std::vector<long long> tim(4 * (size_t)B);
            CHECK(hipMemcpy(tim.data(), d_tim, 32 * (size_t)B, hipMemcpyDeviceToHost));
            long long tmin = tim[0], tmax = 0; double ld = 0, wk = 0;
            for (int32_t bq = 0; bq < B; ++bq) { const long long* t = &tim[4 * (size_t)bq]; tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[2]); ld += (t[1] - t[0]) * 0.01; wk += (t[2] - t[1]) * 0.01; }
            printf("relay W = %d forward sweep, stamps: first start -> last end %.1f us; mean per block: ticket + load %.2f, walk (with its waits) %.2f us\n", W, (tmax - tmin) * 0.01, ld / B, wk / B);
            for (int l = 0; l < nl; l += std::max(1, nl / 12)) {
              long long te = 0, ts = -1, tw = -1; double w = 0; const int nbk = P.launch_ptr[l + 1] - P.launch_ptr[l];
              for (int bq = P.launch_ptr[l]; bq < P.launch_ptr[l + 1]; ++bq) {
                te = std::max(te, tim[4 * (size_t)bq + 2]); if (ts < 0 || tim[4 * (size_t)bq] < ts) ts = tim[4 * (size_t)bq];
                if (tw < 0 || tim[4 * (size_t)bq + 1] < tw) tw = tim[4 * (size_t)bq + 1];
                w += (tim[4 * (size_t)bq + 2] - tim[4 * (size_t)bq + 1]) * 0.01;
              }
              printf("  depth %3d: %4d blocks, first start %8.1f us, first walk %8.1f us, last end %8.1f us, mean walk %6.2f us\n", l, nbk, (ts - tmin) * 0.01, (tw - tmin) * 0.01, (te - tmin) * 0.01, w / nbk);
            }
#ifdef BW_RELAY_STAMPS
          {   // per-step stamps: record landed | hand-over arrived | x + word written — a chain of blocks through the first depths, and the medians of the ramp
            std::vector<long long> stp((size_t)384 * B);
            CHECK(hipMemcpy(stp.data(), d_tim + 4 * (size_t)B, 8 * 384 * (size_t)B, hipMemcpyDeviceToHost));
            long long tmin2 = tim[0]; for (int32_t bq = 0; bq < B; ++bq) tmin2 = std::min(tmin2, tim[4 * (size_t)bq]);
            for (int l : {0, 1, 2, 3, nl / 2}) {
              if (l >= nl) continue;
              const int bq = P.launch_ptr[l] + (l == nl / 2 ? (P.launch_ptr[l + 1] - P.launch_ptr[l]) / 2 : 0);
              const int nsb = std::min(128, (int)P.blocks[bq].nlev);
              printf("  block %d (depth %d, %d steps): per step [us since launch] record landed | go | done  (go - previous done | done - go)\n", bq, l, nsb);
              for (int k2 = 0; k2 < nsb; ++k2) {
                const long long* q = &stp[(size_t)384 * bq + 3 * k2];
                const long long pd = k2 ? stp[(size_t)384 * bq + 3 * (k2 - 1) + 2] : q[1];
                printf("    step %2d: %8.2f | %8.2f | %8.2f   (%5.2f | %5.2f)\n", k2, (q[0] - tmin2) * 0.01, (q[1] - tmin2) * 0.01, (q[2] - tmin2) * 0.01, (q[1] - pd) * 0.01, (q[2] - q[1]) * 0.01);
              }
            }
            // medians over all steps of all blocks: step period (done - previous done), hand-over (go - previous done), tail (done - go), record slack (go - record landed)
            std::vector<double> per, hand, tail, slack;
            for (int32_t bq = 0; bq < B; ++bq) {
              const int nsb = std::min(128, (int)P.blocks[bq].nlev);
              for (int k2 = 1; k2 < nsb; ++k2) {
                const long long* q = &stp[(size_t)384 * bq + 3 * k2];
                const long long pd = stp[(size_t)384 * bq + 3 * (k2 - 1) + 2];
                per.push_back((q[2] - pd) * 0.01); hand.push_back((q[1] - pd) * 0.01); tail.push_back((q[2] - q[1]) * 0.01); slack.push_back((q[1] - q[0]) * 0.01);
              }
            }
            auto med = [](std::vector<double>& v, double f) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
            {   // hand-offs: row published (the producing step's done stamp) -> seen by the reading block's fetcher; the fetchers' polls
              const size_t NM = (size_t)F.nmail + 1024;
              std::vector<long long> tp(NM), tsn(NM), fq(4 * (size_t)B);
              CHECK(hipMemcpy(tp.data(), d_tim + (4 + 384) * (size_t)B, 8 * NM, hipMemcpyDeviceToHost));
              CHECK(hipMemcpy(tsn.data(), d_tim + (4 + 384) * (size_t)B + NM, 8 * NM, hipMemcpyDeviceToHost));
              CHECK(hipMemcpy(fq.data(), d_tim + (4 + 384) * (size_t)B + 2 * NM, 32 * (size_t)B, hipMemcpyDeviceToHost));
              std::vector<double> ho;
              for (size_t m = 0; m < (size_t)F.nmail; ++m) if (tp[m] && tsn[m]) ho.push_back((tsn[m] - tp[m]) * 0.01);
              double np = 0, ps = 0, pm = 0;
              for (int32_t bq = 0; bq < B; ++bq) { np += fq[4 * (size_t)bq]; ps += fq[4 * (size_t)bq + 1] * 0.01; pm = std::max(pm, fq[4 * (size_t)bq + 2] * 0.01); }
              auto pct = [](std::vector<double>& v, double f) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
              {   // the same by placement: the producing block and the reading block on one XCD (one L2) or on two
                std::vector<double> same, cross;
                std::vector<int32_t> row0s(B); for (int32_t bq = 0; bq < B; ++bq) row0s[bq] = P.blocks[bq].row0;
                // blocks are in launch order, their row ranges ascending with it
                for (int32_t bq = 0; bq < B; ++bq) {
                  const bw::Desc& dd = P.blocks[bq];
                  for (int32_t i2 = 0; i2 < dd.npre; ++i2) {
                    const int32_t cellm = F.fl_mb[(size_t)dd.ext0 + i2];
                    const int32_t q2 = P.ext_col[(size_t)dd.ext0 + (F.fl_slot[(size_t)dd.ext0 + i2] - dd.nrows)];
                    const int32_t pb = (int32_t)(std::upper_bound(row0s.begin(), row0s.end(), q2) - row0s.begin()) - 1;
                    if (pb < 0 || !tp[(size_t)cellm] || !tsn[(size_t)cellm]) continue;
                    const double dt = (tsn[(size_t)cellm] - tp[(size_t)cellm]) * 0.01;
                    (tim[4 * (size_t)pb + 3] == tim[4 * (size_t)bq + 3] ? same : cross).push_back(dt);
                  }
                }
                printf("  hand-offs by placement: producer and reader on ONE XCD: %zu, p10 %.2f median %.2f us | on TWO XCDs: %zu, p10 %.2f median %.2f us\n",
                       same.size(), pct(same, 0.1), pct(same, 0.5), cross.size(), pct(cross, 0.1), pct(cross, 0.5));
              }
              printf("  hand-offs (row published -> seen by the reader's fetcher), %zu forward mailboxes: p10 %.2f median %.2f p90 %.2f p99 %.2f us; fetcher polls: %.1f per block, mean round trip %.2f us, longest %.2f us\n",
                     ho.size(), pct(ho, 0.1), pct(ho, 0.5), pct(ho, 0.9), pct(ho, 0.99), np / B, np ? ps / np : 0.0, pm);
              // by depth of the reading... (the producing block's depth: early ramp vs the wide middle)
            }
            printf("  all steps: period p10 %.2f median %.2f p90 %.2f | hand-over (go - previous done) p10 %.2f median %.2f p90 %.2f | tail (done - go) p10 %.2f median %.2f p90 %.2f | go - record landed p10 %.2f median %.2f us\n",
                   med(per, 0.1), med(per, 0.5), med(per, 0.9), med(hand, 0.1), med(hand, 0.5), med(hand, 0.9), med(tail, 0.1), med(tail, 0.5), med(tail, 0.9), med(slack, 0.1), med(slack, 0.5));
          }
#endif
          }
          fa.tim = nullptr; fa.spin_limit = 0u;
        }
      }
      // ---- blocks of right-hand sides: up to 8 columns per workgroup (one walker / fetcher pair of waves per column, one
      // record stream); every column carries the same b and x here, so each must come out as the single-column sweep ----
      if (const char* ec = getenv("BW_COLS")) {
        const int cols = std::max(1, atoi(ec));
        double *d_xm, *d_bm; void* d_mm;
        const size_t mstride = 16 * (size_t)(F.nmail + 1024);
        CHECK(hipMalloc(&d_xm, 8 * (size_t)n * cols)); CHECK(hipMalloc(&d_bm, 8 * (size_t)n * cols)); CHECK(hipMalloc(&d_mm, mstride * cols));
        CHECK(hipMemset(d_mm, 0, mstride * cols));
        for (int c = 0; c < cols; ++c) CHECK(hipMemcpy(d_bm + (size_t)n * c, d_b, 8 * n, hipMemcpyDeviceToDevice));
        bw::FlowArgs<double> fm = fa;
        fm.tim = nullptr; fm.b = d_bm; fm.x = d_xm; fm.mbox = d_mm; fm.ldb = n; fm.ldx = n; fm.mail_stride = (int64_t)mstride;
        std::vector<double> xm((size_t)n * cols);
        printf("== blocks of %d right-hand sides on the dataflow layout (algorithmic bytes per sweep: records %.2f GB + %d x %.2f GB of b, x, mail)\n", cols,
               F.srec.size() / 1e9, cols, (n * 28.0 + P.ext_total * 22.0) / 1e9);
        for (int cap : {1, 2, 3, 4}) {
          if (cap > cols) break;
          for (int c = 0; c < cols; ++c) CHECK(hipMemcpy(d_xm + (size_t)n * c, xb.data(), 8 * n, hipMemcpyHostToDevice));
          CHECK(bw::sweep_flow<double>(fm, mk, F.lds_max, false, false, st, cols, cap));
          CHECK(bw::sweep_flow<double>(fm, mk, F.lds_max, false, true, st, cols, cap));
          CHECK(hipStreamSynchronize(st));
          CHECK(hipMemcpy(xm.data(), d_xm, 8 * (size_t)n * cols, hipMemcpyDeviceToHost));
          int64_t dd = 0;
          for (int c = 0; c < cols; ++c) for (int64_t p = 0; p < n; ++p) dd += xm[(size_t)n * c + p] != xg[p];
          CHECK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
          float msd[2];
          for (int bwd = 0; bwd < 2; ++bwd) {
            const int reps = 5;
            CHECK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) CHECK(bw::sweep_flow<double>(fm, mk, F.lds_max, false, bwd, st, cols, cap));
            CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&msd[bwd], e0, e1)); msd[bwd] /= reps;
          }
          printf("columns per workgroup <= %d: forward %.3f ms, backward %.3f ms (%.3f / %.3f per column); %lld values differ from the single-column sweeps, give-ups %d\n",
                 cap, msd[0], msd[1], msd[0] / cols, msd[1] / cols, (long long)dd, err);
        }
        // stamps of the widest launch (first column's pair)
        fm.tim = d_tim;
        CHECK(bw::sweep_flow<double>(fm, mk, F.lds_max, false, false, st, cols, 0)); CHECK(hipStreamSynchronize(st));
        std::vector<long long> tim(4 * (size_t)B);
        CHECK(hipMemcpy(tim.data(), d_tim, 32 * (size_t)B, hipMemcpyDeviceToHost));
        long long tmin = tim[0], tmax = 0; double ld = 0, wk = 0;
        for (int32_t bq = 0; bq < B; ++bq) { const long long* t = &tim[4 * (size_t)bq]; tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[2]); ld += (t[1] - t[0]) * 0.01; wk += (t[2] - t[1]) * 0.01; }
        printf("forward sweep of the first launch's columns, stamps: first start -> last end %.1f us; mean per block: ticket + load %.2f, walk %.2f us\n", (tmax - tmin) * 0.01, ld / B, wk / B);
        for (int l = 0; l < nl; l += std::max(1, nl / 12)) {
          long long te = 0, ts = -1; double w = 0; const int nbk = P.launch_ptr[l + 1] - P.launch_ptr[l];
          for (int bq = P.launch_ptr[l]; bq < P.launch_ptr[l + 1]; ++bq) {
            te = std::max(te, tim[4 * (size_t)bq + 2]); if (ts < 0 || tim[4 * (size_t)bq] < ts) ts = tim[4 * (size_t)bq];
            w += (tim[4 * (size_t)bq + 2] - tim[4 * (size_t)bq + 1]) * 0.01;
          }
          printf("  depth %3d: %4d blocks, first start %8.1f us, last end %8.1f us, mean walk %6.2f us\n", l, nbk, (ts - tmin) * 0.01, (te - tmin) * 0.01, w / nbk);
        }
      }
    }
  }
  return 0;
}
