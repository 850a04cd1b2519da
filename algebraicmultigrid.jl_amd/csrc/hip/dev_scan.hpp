// dev_scan.hpp — exclusive prefix sums of int32 counts on the device (setup kernels, schedule construction).
#pragma once

namespace {

// ---- exclusive scan of int32 counts (out[n] = total); totals are tracked in 64 bits ---------------------------------
constexpr int kScanT = 1024;

__global__ __launch_bounds__(kScanT) void scan_block_kernel(const int32_t* in, int32_t* out, int64_t n, long long* bsum) {
  __shared__ long long s_w[kScanT / kWave];
  const int64_t i = (int64_t)blockIdx.x * kScanT + threadIdx.x;
  const long long v = i < n ? in[i] : 0;
  long long x = v;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const long long y = __shfl_up(x, o, kWave);
    if ((int)(threadIdx.x & (kWave - 1)) >= o) x += y;
  }
  const int w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  if (lane == kWave - 1) s_w[w] = x;
  __syncthreads();
  long long base = 0;
  for (int q = 0; q < w; ++q) base += s_w[q];
  if (i < n) out[i] = (int32_t)(base + x - v);  // exclusive, block-local
  if (threadIdx.x == kScanT - 1) bsum[blockIdx.x] = base + x;
}
__global__ __launch_bounds__(kScanT) void scan_sums_kernel(long long* bsum, int nb, long long* total) {
  __shared__ long long s_w[kScanT / kWave];
  __shared__ long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nb; c0 += kScanT) {
    const int i = c0 + threadIdx.x;
    const long long v = i < nb ? bsum[i] : 0;
    long long x = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const long long y = __shfl_up(x, o, kWave);
      if ((int)(threadIdx.x & (kWave - 1)) >= o) x += y;
    }
    const int w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    if (lane == kWave - 1) s_w[w] = x;
    __syncthreads();
    long long base = s_carry;
    for (int q = 0; q < w; ++q) base += s_w[q];
    if (i < nb) bsum[i] = base + x - v;
    __syncthreads();
    if (threadIdx.x == kScanT - 1) s_carry = base + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_carry;
}
__global__ void scan_add_kernel(int32_t* out, int64_t n, const long long* bsum, const long long* total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)(out[i] + bsum[i / kScanT]);
  if (i == n) out[n] = (int32_t)(*total);
}

// out[0..n] = exclusive prefix sums of in[0..n); *total_host = the total.  AMGH_EUNSUPPORTED when it leaves int32.
int dev_exclusive_scan(const int32_t* in, int32_t* out, int64_t n, int64_t* total_host, hipStream_t st) {
  const int nb = (int)std::max<int64_t>(1, (n + kScanT - 1) / kScanT);
  long long* bsum = nullptr;
  RC_TRY(dev_alloc(&bsum, nb + 1));
  if (n > 0) hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(kScanT), 0, st, in, out, n, bsum);
  else HIP_TRY(hipMemsetAsync(bsum, 0, sizeof(long long) * (nb + 1), st));
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanT), 0, st, bsum, n > 0 ? nb : 0, bsum + nb);
  hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, out, n, bsum, bsum + nb);
  long long total = 0;
  hipError_t e = hipMemcpyAsync(&total, bsum + nb, sizeof(long long), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(bsum);
  if (e != hipSuccess) return -(1000 + (int)e);
  if (total >= (long long)INT32_MAX) return AMGH_EUNSUPPORTED;
  *total_host = total;
  return AMGH_OK;
}

}  // namespace
