// relay_bench — the dataflow sweep with relayed walkers (csrc/hip/gs_relay.hpp) on an operator file, rows of any length the
// tool is compiled for: flow-only plan (a block's LDS holds x only), forward / backward sweeps checked bit for bit against
// the scalar loops, timings per number of walker waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DBW_PLAN_MAXK=36 -DBW_EXTRA_MAXK=36 [-DBW_EXTRA_DICT] -o tools/relay_bench tools/relay_bench.hip
//   environment: BW_RELAY_LATE=1 (the split row sum), BW_RELAY_DICT=0/1, BW_RELAY_ONLY=W, BW_RELAY_GRID=g, BW_PLAN_CUTS=a,b (offset classes by hand)
// usage: relay_bench PATH [target_rows] [max_rows]     PATH: int64 n, int64 nnz, int32 rowptr[n+1], int32 col[nnz], double val[nnz]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define BW_RELAY_ALL_W 1
#include "../algebraicmultigrid.jl_amd/csrc/hip/gs_relay.hpp"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace amgh;
int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb"); if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
  int64_t hdr[2]; if (fread(hdr, 8, 2, f) != 2) return 1;
  const int64_t n = hdr[0];
  std::vector<int32_t> rp(n + 1), ci(hdr[1]); std::vector<double> va(hdr[1]);
  if (fread(rp.data(), 4, n + 1, f) != (size_t)n + 1 || fread(ci.data(), 4, hdr[1], f) != (size_t)hdr[1] || fread(va.data(), 8, hdr[1], f) != (size_t)hdr[1]) return 1;
  fclose(f);
  bw::Params prm; if (argc > 2) prm.target_rows = atoi(argv[2]);
  prm.threads = 8; prm.flow_only = true; prm.max_rows = argc > 3 ? atoi(argv[3]) : 2048;
  bw::Plan P;
  if (!bw::plan<double>(n, rp.data(), ci.data(), va.data(), prm, &P)) { printf("plan: not eligible\n"); return 0; }
  const int nl = (int)P.launch_ptr.size() - 1;
  int64_t nsteps = 0; for (auto& d : P.blocks) nsteps += d.nlev;
  printf("n = %lld nnz = %lld: %zu blocks, %d depths (dependency levels %d), sum of deepest blocks %lld steps, rows per step %.1f, maxk %d, external x per row %.3f, records %.3f GB\n",
         (long long)n, (long long)P.nnz, P.blocks.size(), nl, P.nlevels, (long long)P.sum_depth, (double)n / nsteps, P.blocks[0].maxk, (double)P.ext_total / n, P.rec.size() / 1e9);
  bw::Flow F;
  if (!bw::structurally_symmetric(n, rp.data(), ci.data(), 8) || !bw::flow_build<double>(P, 8, &F)) { printf("no dataflow layout\n"); return 0; }
  printf("mailboxes %lld (%.3f per row), LDS %.1f KB per block\n", (long long)F.nmail, (double)F.nmail / n, F.lds_max / 1024.0);
  std::vector<double> b(n), x0(n), xb(n), bb(n), xg(n);
  for (int64_t i = 0; i < n; ++i) { b[i] = std::sin(0.37 * (double)(i % 1000)) + 0.5; x0[i] = std::cos(0.11 * (double)(i % 777)); }
  for (int64_t p = 0; p < n; ++p) { xb[p] = x0[P.perm[p]]; bb[p] = b[P.perm[p]]; }
  std::vector<double> xr = x0;
  for (int bwd = 0; bwd < 2; ++bwd)
    for (int64_t s = 0; s < n; ++s) {
      const int64_t i = bwd ? n - 1 - s : s;
      double acc = 0.0, dg = 0.0;
      for (int32_t j = rp[i]; j < rp[i + 1]; ++j) { if (ci[j] == i) dg = va[j]; else acc += va[j] * xr[ci[j]]; }
      if (dg != 0.0) xr[i] = (b[i] - acc) / dg;
    }
  const int32_t B = (int32_t)P.blocks.size();
  bw::Desc* d_blocks; int32_t* d_ext; double *d_b, *d_x;
  bw::FlowDesc* d_fd; unsigned char* d_srec; uint32_t* d_aux; uint16_t* d_fs; int32_t *d_fm, *d_err; void* d_mbox; unsigned long long* d_head;
  CHECK(hipMalloc(&d_blocks, sizeof(bw::Desc) * (size_t)B)); CHECK(hipMalloc(&d_ext, 4 * std::max<size_t>(1, P.ext_col.size())));
  CHECK(hipMalloc(&d_b, 8 * n)); CHECK(hipMalloc(&d_x, 8 * n));
  CHECK(hipMalloc(&d_fd, sizeof(bw::FlowDesc) * (size_t)B)); CHECK(hipMalloc(&d_srec, F.srec.size())); CHECK(hipMalloc(&d_aux, 4 * std::max<size_t>(1, F.aux.size())));
  CHECK(hipMalloc(&d_fs, 2 * F.fl_slot.size())); CHECK(hipMalloc(&d_fm, 4 * F.fl_mb.size()));
  CHECK(hipMalloc(&d_mbox, 16 * (size_t)(F.nmail + 1024))); CHECK(hipMalloc(&d_head, 8)); CHECK(hipMalloc(&d_err, 4));
  CHECK(hipMemcpy(d_blocks, P.blocks.data(), sizeof(bw::Desc) * (size_t)B, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_ext, P.ext_col.data(), 4 * P.ext_col.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_b, bb.data(), 8 * n, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_fd, F.fd.data(), sizeof(bw::FlowDesc) * (size_t)B, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_srec, F.srec.data(), F.srec.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_aux, F.aux.data(), 4 * F.aux.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_fs, F.fl_slot.data(), 2 * F.fl_slot.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_fm, F.fl_mb.data(), 4 * F.fl_mb.size(), hipMemcpyHostToDevice));
  CHECK(hipMemset(d_mbox, 0, 16 * (size_t)(F.nmail + 1024))); CHECK(hipMemset(d_head, 0, 8)); CHECK(hipMemset(d_err, 0, 4));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  bw::FlowArgs<double> fa{d_blocks, d_fd, d_srec, d_aux, d_ext, d_fm, d_fs, d_mbox, d_b, d_x, 1.0, d_head, B, (int32_t)F.nmail, d_err, nullptr, 1u << 18, -1};
  const int mk = P.blocks[0].maxk;
  // the dictionary layout (FlowDict): distinct value rows per block in LDS, the record carries columns only
  unsigned char *d_crec = nullptr, *d_dict = nullptr; int32_t* d_dent = nullptr;
  if (F.dc.on) {
    int64_t drows = 0; int dmax = 0;
    for (int32_t e : F.dc.ent) { drows += ((uint32_t)e >> 24) + 1; dmax = std::max(dmax, (int)((uint32_t)e >> 24) + 1); }
    printf("dictionary layout: %.3f GB column records (%.1f B per row), %.1f distinct value rows per block (max %d), %.3f MB of dictionaries, LDS %.1f KB per block\n",
           F.dc.crec.size() / 1e9, (double)F.dc.crec.size() / n, (double)drows / B, dmax, F.dc.dict.size() / 1e6, F.dc.lds_max / 1024.0);
    CHECK(hipMalloc(&d_crec, F.dc.crec.size())); CHECK(hipMalloc(&d_dict, F.dc.dict.size() + 16)); CHECK(hipMalloc(&d_dent, 4 * (size_t)B));
    CHECK(hipMemcpy(d_crec, F.dc.crec.data(), F.dc.crec.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_dict, F.dc.dict.data(), F.dc.dict.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_dent, F.dc.ent.data(), 4 * (size_t)B, hipMemcpyHostToDevice));
  } else printf("no dictionary layout for this operator\n");
  if (const char* eg = getenv("BW_RELAY_GRID")) fa.grid = atoi(eg);
  if (const char* el = getenv("BW_RELAY_LATE")) fa.late = (atoi(el) != 0 && P.late_ok) ? 1 : 0;   // the dependency-aware row sum (not the scalar loops' bits: max |diff| below)
  printf("records split around the padding (late_ok): %d; row sum: %s\n", (int)P.late_ok, fa.late ? "far half above the hand-over (LATE)" : "stored order");
  for (int dict = 0; dict < (F.dc.on ? 2 : 1); ++dict) {
  fa.crec = dict ? d_crec : nullptr; fa.dict = dict ? d_dict : nullptr; fa.dict_ent = dict ? d_dent : nullptr;
  const size_t lds_use = dict ? F.dc.lds_max : F.lds_max;
  if (const char* ed = getenv("BW_RELAY_DICT")) if (atoi(ed) != dict) continue;
  printf("---- %s records ----\n", dict ? "dictionary" : "plain");
  const double fbytes = (double)(dict ? F.dc.crec.size() + F.dc.dict.size() : F.srec.size()) + n * 28.0 + P.ext_total * 22.0;
  for (int W : {2, 3, 4}) {
    if (const char* ew = getenv("BW_RELAY_ONLY")) if (atoi(ew) != W) continue;
    CHECK(hipMemset(d_err, 0, 4));
    CHECK(hipMemcpy(d_x, xb.data(), 8 * n, hipMemcpyHostToDevice));
    hipError_t e = bw::sweep_relay<double>(fa, mk, lds_use, false, false, st, W);
    if (e != hipSuccess) { printf("relay W = %d: launch failed: %s\n", W, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    CHECK(bw::sweep_relay<double>(fa, mk, lds_use, false, true, st, W));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(xg.data(), d_x, 8 * n, hipMemcpyDeviceToHost));
    int err = 0; CHECK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    int64_t dd = 0; double md = 0.0;
    for (int64_t p = 0; p < n; ++p) { const double v = xr[P.perm[p]]; if (xg[p] != v) ++dd; md = std::max(md, std::fabs(xg[p] - v)); }
    printf("relay W = %d: forward + backward vs the scalar loops: %lld values differ (max |diff| %.3e), give-ups %d\n", W, (long long)dd, md, err);
    for (int bwd = 0; bwd < 2; ++bwd) {
      CHECK(bw::sweep_relay<double>(fa, mk, lds_use, false, bwd, st, W)); CHECK(hipStreamSynchronize(st));
      const int reps = 5;
      CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) CHECK(bw::sweep_relay<double>(fa, mk, lds_use, false, bwd, st, W));
      CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("relay W = %d %s sweep: %.3f ms (%.0f GB/s)\n", W, bwd ? "backward" : "forward ", ms / reps, fbytes / (ms / reps * 1e-3) / 1e9);
    }
  }
  }
  return 0;
}
