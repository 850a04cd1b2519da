// Which hardware queue does the runtime give a new stream?  Pairs of streams are tested for concurrency with the flag exchange of
// dist_pipe_probe (amghip_dist.hpp): two streams on one hardware queue run their kernels one after the other.
//   hipcc --offload-arch=gfx950 -O2 tools/hwq_probe.hip -o tools/hwq_probe && GPU_MAX_HW_QUEUES=8 tools/hwq_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
__global__ void probe(unsigned* mine, const unsigned* peer, unsigned tag, long long ticks, unsigned* result) {
  __hip_atomic_store(mine, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long t0 = wall_clock64();
  bool ok = false;
  for (;;) {
    if (__hip_atomic_load(peer, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == tag) { ok = true; break; }
    if (wall_clock64() - t0 > ticks) break;
    __builtin_amdgcn_s_sleep(32);
  }
  *result = ok ? 1u : 2u;
}
static unsigned* g_buf = nullptr;
static unsigned g_tag = 1;
static bool concurrent(hipStream_t a, hipStream_t b) {
  ++g_tag;
  hipMemset(g_buf, 0, 16);
  hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, a, g_buf + 0, g_buf + 1, g_tag, 300000LL, g_buf + 2);   // 3 ms
  hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, b, g_buf + 1, g_buf + 0, g_tag, 300000LL, g_buf + 3);
  hipStreamSynchronize(a); hipStreamSynchronize(b);
  unsigned r[4]; hipMemcpy(r, g_buf, 16, hipMemcpyDeviceToHost);
  return r[2] == 1u && r[3] == 1u;
}
static void classes(const std::vector<hipStream_t>& s, const char* what) {
  const int n = (int)s.size();
  std::vector<int> cls(n, -1);
  int nc = 0;
  for (int i = 0; i < n; ++i) {
    if (cls[i] >= 0) continue;
    cls[i] = nc;
    for (int j = i + 1; j < n; ++j) if (cls[j] < 0 && !concurrent(s[i], s[j])) cls[j] = nc;
    ++nc;
  }
  printf("%s: %d streams on %d queues:", what, n, nc);
  for (int i = 0; i < n; ++i) printf(" %d", cls[i]);
  printf("\n");
}
int main() {
  hipMalloc(&g_buf, 64);
  std::vector<hipStream_t> s(16);
  for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  classes(s, "16 streams created in a row");
  // vacate one queue: destroy every stream that shares a queue with s[1]
  std::vector<hipStream_t> keep;
  for (int i = 0; i < 16; ++i) { if (i == 1 || !concurrent(s[1], s[i]) ) { if (i != 1) hipStreamDestroy(s[i]); } else keep.push_back(s[i]); }
  hipStreamDestroy(s[1]);
  classes(keep, "after destroying the streams of one queue");
  std::vector<hipStream_t> fresh(4);
  for (auto& x : fresh) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  classes(fresh, "4 fresh streams (one queue was empty)");
  std::vector<hipStream_t> all = keep; all.insert(all.end(), fresh.begin(), fresh.end());
  classes(all, "kept + fresh");
  // the remedy tried in amgh_dist_finalize: ballast streams first (they level the queues' reference counts), then the real ones
  for (auto x : fresh) hipStreamDestroy(x);
  std::vector<hipStream_t> ballast(16);
  for (auto& x : ballast) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  for (auto& x : fresh) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  classes(fresh, "4 fresh streams behind 16 ballast streams");
  for (auto x : ballast) hipStreamDestroy(x);
  classes(fresh, "... after the ballast is gone");
  return 0;
}
