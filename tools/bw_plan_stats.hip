// bw_plan_stats — the block plan of csrc/hip/gs_blocks.hpp on an operator file, host only (no device needed): blocks,
// depths of the quotient graph, steps per block, rows per step, external columns per row, row lengths.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBW_PLAN_MAXK=96 -o tools/bw_plan_stats tools/bw_plan_stats.hip
// usage: bw_plan_stats PATH [target_rows]    PATH: int64 n, int64 nnz, int32 rowptr[n+1], int32 col[nnz], double val[nnz]
#include <cstdio>
#include <cstdlib>
#include "../algebraicmultigrid.jl_amd/csrc/hip/gs_flow.hpp"
using namespace amgh;
int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
  int64_t hdr[2]; if (fread(hdr, 8, 2, f) != 2) return 1;
  const int64_t n = hdr[0];
  std::vector<int32_t> rp(n + 1), ci(hdr[1]); std::vector<double> va(hdr[1]);
  if (fread(rp.data(), 4, n + 1, f) != (size_t)n + 1 || fread(ci.data(), 4, hdr[1], f) != (size_t)hdr[1] || fread(va.data(), 8, hdr[1], f) != (size_t)hdr[1]) return 1;
  fclose(f);
  bw::Params prm; if (argc > 2) prm.target_rows = atoi(argv[2]);
  prm.threads = 8;
  if (argc > 3) prm.flow_only = atoi(argv[3]) != 0;
  if (argc > 4) prm.max_rows = atoi(argv[4]);
  bw::Plan P;
  if (!bw::plan<double>(n, rp.data(), ci.data(), va.data(), prm, &P)) { printf("plan: not eligible\n"); return 0; }
  const int nl = (int)P.launch_ptr.size() - 1;
  int maxb = 0; for (int l = 0; l < nl; ++l) maxb = std::max(maxb, P.launch_ptr[l + 1] - P.launch_ptr[l]);
  std::vector<int> rows, steps; int64_t nsteps = 0, npre = 0;
  for (auto& d : P.blocks) { rows.push_back(d.nrows); steps.push_back(d.nlev); nsteps += d.nlev; npre += d.npre; }
  std::sort(rows.begin(), rows.end()); std::sort(steps.begin(), steps.end());
  printf("n = %lld nnz = %lld: cuts %d %d, ranges %d %d %d, cells %d x %d x %d -> %zu blocks (rows min %d median %d max %d), maxk %d\n",
         (long long)n, (long long)P.nnz, P.cuts[0], P.cuts[1], P.range[0], P.range[1], P.range[2], P.cells[0], P.cells[1], P.cells[2], P.blocks.size(),
         rows.front(), rows[rows.size() / 2], rows.back(), P.blocks[0].maxk);
  printf("depths %d (dependency levels %d), up to %d blocks per depth; sum of deepest blocks %lld steps; steps per block min %d median %d max %d; rows per step %.1f\n",
         nl, P.nlevels, maxb, (long long)P.sum_depth, steps.front(), steps[steps.size() / 2], steps.back(), (double)n / nsteps);
  printf("external x per row %.3f (near side %.3f); LDS %.1f KB; record bytes %.3f GB (%.1f B per row; CSR %.1f B per row); model %.3f ms launched, %.3f ms chained\n",
         (double)P.ext_total / n, (double)npre / n, P.lds_max / 1024.0, P.rec.size() / 1e9, (double)P.rec.size() / n, (P.nnz * 12.0 + n * 4.0) / n, P.est_seconds * 1e3, P.est_chain_seconds * 1e3);
  bw::Flow F;
  const bool sym = bw::structurally_symmetric(n, rp.data(), ci.data(), 8);
  const bool ok = sym && bw::flow_build<double>(P, 8, &F);
  printf("structurally symmetric %d, flow ok %d, mailboxes %lld (%.3f per row), flow LDS %.1f KB\n", (int)sym, (int)ok, (long long)F.nmail, (double)F.nmail / n, F.lds_max / 1024.0);
  return 0;
}
