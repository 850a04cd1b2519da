// Cost of a device-wide barrier inside one persistent kernel vs. a kernel boundary, on the pattern the
// Gauss-Seidel sweeps have: step k reads values step k-1 wrote from OTHER workgroups (other XCDs).
//   hipcc --offload-arch=gfx950 -O3 -o tools/grid_barrier_bench tools/grid_barrier_bench.hip
// Every spin is bounded: a barrier that never completes sets an error flag and the kernel exits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kSpinLimit = 1 << 22;

__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, int* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();  // release: this workgroup's stores visible device-wide
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit) { *err = 1; ok = false; break; }
    }
    __threadfence();  // acquire
  }
  __syncthreads();
  return ok;
}

// step k: every thread reads x[(i + shift) % n] written in step k-1 by another workgroup, adds 1, writes y[i]
__global__ void persistent_kernel(double* a, double* b, int n, int steps, int shift, unsigned* cnt, int* err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double* src = a;
  double* dst = b;
  for (int k = 0; k < steps; ++k) {
    if (i < n) {
      int j = i + shift; if (j >= n) j -= n;
      dst[i] = __builtin_nontemporal_load(src + j) + 1.0;
    }
    if (!grid_barrier(cnt, (unsigned)(k + 1) * gridDim.x, err)) return;
    double* t = src; src = dst; dst = t;
  }
}

__global__ void step_kernel(const double* src, double* dst, int n, int shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    int j = i + shift; if (j >= n) j -= n;
    dst[i] = src[j] + 1.0;
  }
}

int main() {
  const int steps = 2000;
  for (int wgs : {64, 256, 512, 1024}) {
    const int threads = 512, n = wgs * threads, shift = n / 2 + 7;
    double *a, *b; unsigned* cnt; int* err;
    CHECK(hipMalloc(&a, 8 * n)); CHECK(hipMalloc(&b, 8 * n)); CHECK(hipMalloc(&cnt, 4)); CHECK(hipMalloc(&err, 4));
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms_k = 0, ms_p = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemsetAsync(a, 0, 8 * n, st)); CHECK(hipMemsetAsync(b, 0, 8 * n, st));
      CHECK(hipEventRecord(e0, st));
      for (int k = 0; k < steps; ++k) {
        hipLaunchKernelGGL(step_kernel, dim3(wgs), dim3(threads), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, n, shift);
      }
      CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st)); CHECK(hipEventElapsedTime(&ms_k, e0, e1));
    }
    std::vector<double> ref(n); CHECK(hipMemcpy(ref.data(), a, 8 * n, hipMemcpyDeviceToHost));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemsetAsync(a, 0, 8 * n, st)); CHECK(hipMemsetAsync(b, 0, 8 * n, st));
      CHECK(hipMemsetAsync(cnt, 0, 4, st)); CHECK(hipMemsetAsync(err, 0, 4, st));
      int nn = n, ss = steps, sh = shift;
      void* args[] = {&a, &b, &nn, &ss, &sh, &cnt, &err};
      CHECK(hipEventRecord(e0, st));
      CHECK(hipLaunchCooperativeKernel((void*)persistent_kernel, dim3(wgs), dim3(threads), args, 0, st));
      CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st)); CHECK(hipEventElapsedTime(&ms_p, e0, e1));
    }
    std::vector<double> got(n); CHECK(hipMemcpy(got.data(), a, 8 * n, hipMemcpyDeviceToHost));
    int herr = 0; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += (got[i] != ref[i]);
    printf("workgroups %4d x %d: kernel per step %.2f us   persistent + grid barrier %.2f us per step   (barrier timeout %d, mismatches %d, value %.0f)\n",
           wgs, threads, 1e3 * ms_k / steps, 1e3 * ms_p / steps, herr, bad, ref[0]);
    hipFree(a); hipFree(b); hipFree(cnt); hipFree(err);
  }
  return 0;
}
