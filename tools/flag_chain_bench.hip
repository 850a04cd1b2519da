// Point-to-point synchronisation between workgroups of ONE persistent kernel: every workgroup advances step by
// step, waiting only for its two neighbours' previous step (flags, no device-wide barrier), reading what they
// wrote.  This is the dependency pattern of a level-scheduled Gauss-Seidel sweep with rows assigned to
// workgroups by position.  Compare with a kernel launch per step (tools/grid_barrier_bench.hip: 3.2 us).
//   hipcc --offload-arch=gfx950 -O3 -o tools/flag_chain_bench tools/flag_chain_bench.hip
// All spins are bounded (error flag + exit), the launch is cooperative (all workgroups resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kSpinLimit = 1 << 22;
constexpr int W = 64;  // doubles each workgroup publishes per step

__device__ __forceinline__ bool wait_flag(const unsigned* f, unsigned want, int* err) {
  int spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpinLimit) { *err = 1; return false; }
  }
  return true;
}

// d: 2 x G x W doubles (ping-pong by step parity), flag[i] = last step workgroup i has published
__global__ void chain_kernel(double* d, unsigned* flag, int steps, int* err) {
  const int G = gridDim.x, i = blockIdx.x, t = threadIdx.x;
  const int l = (i + G - 1) % G, r = (i + 1) % G;
  __shared__ int s_ok;
  for (int k = 1; k <= steps; ++k) {
    if (t == 0) {
      bool ok = wait_flag(flag + l, (unsigned)(k - 1), err) && wait_flag(flag + r, (unsigned)(k - 1), err);
      __threadfence();  // acquire
      s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    const double* src = d + (size_t)((k - 1) & 1) * G * W;
    double* dst = d + (size_t)(k & 1) * G * W;
    if (t < W) dst[(size_t)i * W + t] = 0.5 * (src[(size_t)l * W + t] + src[(size_t)r * W + t]) + 1.0;
    __syncthreads();
    if (t == 0) {
      __threadfence();  // release
      __hip_atomic_store(flag + i, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main() {
  const int steps = 2000;
  for (int G : {64, 256, 512, 1024}) {
    const int threads = 512;
    double* d; unsigned* flag; int* err;
    CHECK(hipMalloc(&d, sizeof(double) * 2 * G * W)); CHECK(hipMalloc(&flag, 4 * G)); CHECK(hipMalloc(&err, 4));
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    std::vector<double> init(2 * G * W);
    for (int i = 0; i < G; ++i) for (int t = 0; t < W; ++t) init[(size_t)i * W + t] = (double)(i % 7);
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemcpy(d, init.data(), sizeof(double) * 2 * G * W, hipMemcpyHostToDevice));
      CHECK(hipMemset(flag, 0, 4 * G)); CHECK(hipMemset(err, 0, 4));
      int ss = steps;
      void* args[] = {&d, &flag, &ss, &err};
      CHECK(hipEventRecord(e0, st));
      CHECK(hipLaunchCooperativeKernel((void*)chain_kernel, dim3(G), dim3(threads), args, 0, st));
      CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    // host reference
    std::vector<double> a(init.begin(), init.begin() + G * W), b(G * W);
    for (int k = 1; k <= steps; ++k) {
      for (int i = 0; i < G; ++i) for (int t = 0; t < W; ++t)
        b[(size_t)i * W + t] = 0.5 * (a[(size_t)((i + G - 1) % G) * W + t] + a[(size_t)((i + 1) % G) * W + t]) + 1.0;
      a.swap(b);
    }
    std::vector<double> got(2 * G * W);
    CHECK(hipMemcpy(got.data(), d, sizeof(double) * 2 * G * W, hipMemcpyDeviceToHost));
    int herr = 0; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    int bad = 0;
    const double* res = got.data() + (size_t)(steps & 1) * G * W;
    for (int i = 0; i < G * W; ++i) bad += (res[i] != a[i]);
    printf("workgroups %4d x %d: neighbour-flag step %.2f us   (spin timeout %d, mismatches %d)\n", G, threads, 1e3 * ms / steps, herr, bad);
    (void)hipFree(d); (void)hipFree(flag); (void)hipFree(err);
  }
  return 0;
}
