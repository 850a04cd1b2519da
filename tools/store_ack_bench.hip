// How long does a wave wait for its own stores?  On gfx950 loads and stores share vmcnt and retire in order: a step of the
// dataflow sweep (gs_flow.hpp) that waits for operands loaded D - 1 steps ago also waits for the stores of those steps.
// One wave, ITER iterations of { store 64 x 16 bytes (write-through sc1 | plain), optional load, s_waitcnt vmcnt(K) }:
// time per iteration against K gives the acknowledgement latency of each kind of store.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/store_ack_bench tools/store_ack_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int K, bool SC1, bool LOAD>
__global__ void k(unsigned char* buf, const unsigned char* src, long long* out, int iters) {
  const unsigned long long u = (unsigned long long)buf, us = (unsigned long long)src;
  i32x4 rs, rl;
  rs.x = __builtin_amdgcn_readfirstlane((int)u); rs.y = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffff)); rs.z = 0x7ffffff0; rs.w = 0x00020000;
  rl.x = __builtin_amdgcn_readfirstlane((int)us); rl.y = __builtin_amdgcn_readfirstlane((int)((us >> 32) & 0xffff)); rl.z = 0x7ffffff0; rl.w = 0x00020000;
  const unsigned lane = threadIdx.x;
  u32x4 v = {lane, 1u, lane, 1u};
  u32x4 ld = {0, 0, 0, 0};
  const long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    const unsigned off = (unsigned)(((i & 1023) * 64 + lane) * 16u);
    if (LOAD) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld) : "v"(off), "s"(rl) : "memory");
    if (SC1) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1" :: "v"(v), "v"(off), "s"(rs) : "memory");
    else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" :: "v"(v), "v"(off), "s"(rs) : "memory");
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(K) : "memory");
    if (LOAD) asm volatile("" : "+v"(ld));
    v.x += ld.x;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) out[0] = wall_clock64() - t0;
}

template <int K, bool SC1, bool LOAD> void run(unsigned char* buf, unsigned char* src, long long* out, const char* what) {
  const int iters = 4000;
  hipLaunchKernelGGL((k<K, SC1, LOAD>), dim3(1), dim3(64), 0, nullptr, buf, src, out, iters);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<K, SC1, LOAD>), dim3(1), dim3(64), 0, nullptr, buf, src, out, iters);
  CHECK(hipDeviceSynchronize());
  long long t; CHECK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
  printf("%-34s vmcnt(%2d): %7.1f ns per iteration\n", what, K, (double)t * 10.0 / iters);
}

int main() {
  unsigned char *buf, *src; long long* out;
  CHECK(hipMalloc(&buf, 1 << 21)); CHECK(hipMalloc(&src, 1 << 21)); CHECK(hipMalloc(&out, 8));
  CHECK(hipMemset(src, 0, 1 << 21));
  printf("one wave, 64 x 16-byte stores per iteration (wall clock 100 MHz); in flight = K + 1 memory operations\n");
  run<0, true, false>(buf, src, out, "sc1 store (write-through)"); run<1, true, false>(buf, src, out, "sc1 store (write-through)");
  run<2, true, false>(buf, src, out, "sc1 store (write-through)"); run<4, true, false>(buf, src, out, "sc1 store (write-through)");
  run<8, true, false>(buf, src, out, "sc1 store (write-through)"); run<16, true, false>(buf, src, out, "sc1 store (write-through)");
  run<0, false, false>(buf, src, out, "plain store"); run<1, false, false>(buf, src, out, "plain store");
  run<2, false, false>(buf, src, out, "plain store"); run<4, false, false>(buf, src, out, "plain store");
  run<8, false, false>(buf, src, out, "plain store"); run<16, false, false>(buf, src, out, "plain store");
  run<0, true, true>(buf, src, out, "load (L2 hit) + sc1 store"); run<2, true, true>(buf, src, out, "load (L2 hit) + sc1 store");
  run<6, true, true>(buf, src, out, "load (L2 hit) + sc1 store");
  run<0, false, true>(buf, src, out, "load (L2 hit) + plain store"); run<2, false, true>(buf, src, out, "load (L2 hit) + plain store");
  run<6, false, true>(buf, src, out, "load (L2 hit) + plain store");
  return 0;
}
