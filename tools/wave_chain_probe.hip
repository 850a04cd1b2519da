// wave_chain_probe — what ONE wave64 pays per dependent operation on gfx950 (the block-wavefront sweeps are one wave
// walking a dependency chain): dependent fp64 FMA, fp64 division, LDS pointer chase, LDS write -> read, v_readlane.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/wave_chain_probe tools/wave_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int N = 4096;
__global__ void probe(double* out, long long* t, int* chase_init) {
  __shared__ int chase[1024];
  __shared__ double buf[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) { chase[i] = chase_init[i]; buf[i] = 1.0 + i; }
  __syncthreads();
  double a = 1.0 + lane * 1e-9, b = 1.0000001, c = 1e-9;
  long long t0 = wall_clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) a = __builtin_fma(a, b, c);
  long long t1 = wall_clock64();
  double d = 3.0 + lane;
#pragma unroll 4
  for (int i = 0; i < N; ++i) d = (1.5 + d) / (d + 0.25);
  long long t2 = wall_clock64();
  int p = lane;
#pragma unroll 16
  for (int i = 0; i < N; ++i) p = chase[p];
  long long t3 = wall_clock64();
  double w = a;
#pragma unroll 8
  for (int i = 0; i < N; ++i) { buf[lane] = w; w = buf[(lane + 1) & 63] + 1.0; }
  long long t4 = wall_clock64();
  double acc = 0.0;
#pragma unroll 4
  for (int i = 0; i < N; ++i) {   // a 6-entry row: 6 gathers, 6 dependent multiply-adds, one store
    double x0 = buf[(p + i) & 1023], x1 = buf[(p + 2 * i) & 1023], x2 = buf[(p + 3 * i) & 1023], x3 = buf[(p + 5 * i) & 1023], x4 = buf[(p + 7 * i) & 1023], x5 = buf[(p + 11 * i) & 1023];
    acc = 0.0; acc += 0.5 * x0; acc += 0.25 * x1; acc += 0.125 * x2; acc += 0.5 * x3; acc += 0.25 * x4; acc += 0.125 * x5;
    buf[(lane + i) & 1023] = (1.0 - acc) * 0.1666;
  }
  long long t5 = wall_clock64();
  out[lane] = a + d + p + w + acc;
  if (lane == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; }
}
int main() {
  double* out; long long* t; int* ci; int h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (i * 37 + 11) & 1023;
  CHECK(hipMalloc(&out, 8 * 64)); CHECK(hipMalloc(&t, 8 * 8)); CHECK(hipMalloc(&ci, 4096));
  CHECK(hipMemcpy(ci, h, 4096, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) { probe<<<1, 64>>>(out, t, ci); CHECK(hipDeviceSynchronize()); }
  long long ht[8]; CHECK(hipMemcpy(ht, t, 64, hipMemcpyDeviceToHost));
  const char* what[5] = {"dependent fp64 fma", "dependent fp64 add + add + division", "LDS pointer chase (ds_read_b32 -> address)", "LDS write -> read (other lane) -> add",
                         "6 LDS gathers + 6 dependent mul/add + 1 LDS write"};
  for (int k = 0; k < 5; ++k) printf("%-52s %7.1f ns per iteration\n", what[k], ht[k] * 10.0 / N);
  return 0;
}
