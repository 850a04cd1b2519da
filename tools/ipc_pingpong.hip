// ipc_pingpong.hip — round-trip latency of cross-PROCESS stream hand-offs on this box (2 processes, GPU devA / devB):
//   (a) hipStreamWriteValue64 / hipStreamWaitValue64 on a host-registered shared-memory flag   (what the IPC transport used first)
//   (b) a one-thread kernel that stores the flag (system scope) / a one-wave kernel that polls it
//   (c) host hand-off: hipStreamSynchronize + a flag in shared memory polled by the host + the next launch
// Each round trip: A signals k, B waits for it and signals back, A waits.  Usage: ipc_pingpong [devA devB] [rounds]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__); fflush(stdout); _exit(3); } } while (0)

struct Shared { std::atomic<uint64_t> fa, fb; std::atomic<int> ready; hipIpcMemHandle_t ha, hb; char pad[4096]; };

__global__ void store_flag(uint64_t* f, uint64_t v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void wait_flag(uint64_t* f, uint64_t v, int* err) {
  long spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1L << 26)) { *err = 1; return; }
  }
}
__global__ void tiny(double* p) { if (p) p[0] += 1.0; }
__global__ void copyk(const double* __restrict__ s, double* __restrict__ d, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = s[i]; }

static double now();
static double pingpong_a(bool A, hipStream_t st, uint64_t* dfa, uint64_t* dfb, uint64_t& k, int rounds, double* dummy, Shared* sh, int& phase) {
  CK(hipStreamSynchronize(st));
  sh->ready.fetch_add(1);
  phase += 2;
  while (sh->ready.load() < phase) usleep(50);
  const double t0 = now();
  for (int r = 0; r < rounds; ++r) {
    ++k;
    if (A) { CK(hipStreamWriteValue64(st, dfa, k, 0)); CK(hipStreamWaitValue64(st, dfb, k, hipStreamWaitValueGte, ~0ull)); }
    else   { CK(hipStreamWaitValue64(st, dfa, k, hipStreamWaitValueGte, ~0ull)); CK(hipStreamWriteValue64(st, dfb, k, 0)); }
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, dummy);
  }
  CK(hipStreamSynchronize(st));
  return 1e6 * (now() - t0) / rounds;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int devA = argc > 2 ? atoi(argv[1]) : 0, devB = argc > 2 ? atoi(argv[2]) : 0;
  const int rounds = argc > 3 ? atoi(argv[3]) : (argc == 2 ? atoi(argv[1]) : 2000);
  Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  new (sh) Shared();
  sh->fa = 0; sh->fb = 0; sh->ready = 0;
  const pid_t pid = fork();
  const bool A = pid != 0;
  CK(hipSetDevice(A ? devA : devB));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipHostRegister((void*)sh, sizeof(Shared), hipHostRegisterMapped | hipHostRegisterPortable));
  void* dp = nullptr; CK(hipHostGetDevicePointer(&dp, (void*)sh, 0));
  uint64_t* dfa = (uint64_t*)((char*)dp + offsetof(Shared, fa));
  uint64_t* dfb = (uint64_t*)((char*)dp + offsetof(Shared, fb));
  int* derr; CK(hipMalloc(&derr, 4)); CK(hipMemset(derr, 0, 4));
  double* dummy; CK(hipMalloc(&dummy, 8)); CK(hipMemset(dummy, 0, 8));
  // a 512 KiB buffer per process (one halo face of the 256^3 problem), peer-mapped by the other one
  const int nbuf = 65536;
  double *mine, *peer, *land;
  uint64_t k = 0;
  int phase = 0;
  sh->ready.fetch_add(1);
  while (sh->ready.load() < 2) usleep(100);
  phase = 2;
  double t = pingpong_a(A, st, dfa, dfb, k, rounds, dummy, sh, phase);
  if (A) printf("(a0) write / wait value ping-pong, nothing else in the process        %8.1f us per round trip\n", t);
  CK(hipMalloc(&mine, 8 * nbuf)); CK(hipMemset(mine, 0, 8 * nbuf)); CK(hipMalloc(&land, 8 * nbuf));
  hipStream_t cs; CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  t = pingpong_a(A, st, dfa, dfb, k, rounds, dummy, sh, phase);
  if (A) printf("(a1) ... after two hipMalloc, a second stream, two events             %8.1f us per round trip\n", t);
  CK(hipIpcGetMemHandle(A ? &sh->ha : &sh->hb, mine));
  t = pingpong_a(A, st, dfa, dfb, k, rounds, dummy, sh, phase);
  if (A) printf("(a2) ... after hipIpcGetMemHandle of the own buffer                    %8.1f us per round trip\n", t);
  CK(hipIpcOpenMemHandle((void**)&peer, A ? sh->hb : sh->ha, hipIpcMemLazyEnablePeerAccess));
  t = pingpong_a(A, st, dfa, dfb, k, rounds, dummy, sh, phase);
  if (A) printf("(a3) ... after hipIpcOpenMemHandle of the peer's buffer                %8.1f us per round trip\n", t);
  fflush(stdout);
  for (int variant = 0; variant < 7; ++variant) {
    // rendezvous on the host before every variant
    CK(hipStreamSynchronize(st));
    sh->ready.fetch_add(1);
    phase += 2;
    while (sh->ready.load() < phase) usleep(50);
    const double t0 = now();
    if (variant == 0) {
      for (int r = 0; r < rounds; ++r) {
        ++k;
        if (A) { CK(hipStreamWriteValue64(st, dfa, k, 0)); CK(hipStreamWaitValue64(st, dfb, k, hipStreamWaitValueGte, ~0ull)); }
        else   { CK(hipStreamWaitValue64(st, dfa, k, hipStreamWaitValueGte, ~0ull)); CK(hipStreamWriteValue64(st, dfb, k, 0)); }
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, dummy);
      }
      CK(hipStreamSynchronize(st));
    } else if (variant == 1) {
      for (int r = 0; r < rounds; ++r) {
        ++k;
        if (A) { hipLaunchKernelGGL(store_flag, dim3(1), dim3(1), 0, st, dfa, k); hipLaunchKernelGGL(wait_flag, dim3(1), dim3(1), 0, st, dfb, k, derr); }
        else   { hipLaunchKernelGGL(wait_flag, dim3(1), dim3(1), 0, st, dfa, k, derr); hipLaunchKernelGGL(store_flag, dim3(1), dim3(1), 0, st, dfb, k); }
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, dummy);
      }
      CK(hipStreamSynchronize(st));
    } else if (variant == 2) {
      for (int r = 0; r < rounds; ++r) {
        ++k;
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, dummy);
        CK(hipStreamSynchronize(st));
        if (A) { sh->fa.store(k); while (sh->fb.load() < k) {} }
        else   { while (sh->fa.load() < k) {} sh->fb.store(k); }
      }
    } else if (variant == 3 || variant == 4) {
      // the transport's exchange, both directions at once: signal, wait for the peer's signal, pull 512 KiB from its buffer
      // (3: hipMemcpyAsync, 4: a copy kernel), all on ONE stream
      for (int r = 0; r < rounds; ++r) {
        ++k;
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, mine);
        CK(hipStreamWriteValue64(st, A ? dfa : dfb, k, 0));
        CK(hipStreamWaitValue64(st, A ? dfb : dfa, k, hipStreamWaitValueGte, ~0ull));
        if (variant == 3) CK(hipMemcpyAsync(land, peer, 8 * nbuf, hipMemcpyDeviceToDevice, st));
        else hipLaunchKernelGGL(copyk, dim3(nbuf / 256), dim3(256), 0, st, (const double*)peer, land, nbuf);
      }
      CK(hipStreamSynchronize(st));
    } else {
      // the same with the copy on a second stream between two events (5: hipMemcpyAsync, 6: copy kernel): the transport's shape
      for (int r = 0; r < rounds; ++r) {
        ++k;
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, mine);
        CK(hipStreamWriteValue64(st, A ? dfa : dfb, k, 0));
        CK(hipEventRecord(e1, st)); CK(hipStreamWaitEvent(cs, e1, 0));
        CK(hipStreamWaitValue64(cs, A ? dfb : dfa, k, hipStreamWaitValueGte, ~0ull));
        if (variant == 5) CK(hipMemcpyAsync(land, peer, 8 * nbuf, hipMemcpyDeviceToDevice, cs));
        else hipLaunchKernelGGL(copyk, dim3(nbuf / 256), dim3(256), 0, cs, (const double*)peer, land, nbuf);
        CK(hipEventRecord(e2, cs)); CK(hipStreamWaitEvent(st, e2, 0));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, land);
      }
      CK(hipStreamSynchronize(st));
    }
    const double dt = now() - t0;
    int herr = 0; CK(hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost));
    if (A) printf("%-70s %8.1f us per round trip%s\n",
                  variant == 0 ? "(a) hipStreamWriteValue64 / hipStreamWaitValue64" :
                  variant == 1 ? "(b) store kernel / polling kernel (system-scope atomics on registered shm)" :
                  variant == 2 ? "(c) host hand-off: stream sync + host-polled flag" :
                  variant == 3 ? "(d) exchange on one stream: write, wait, hipMemcpyAsync 512 KiB from the peer" :
                  variant == 4 ? "(e) exchange on one stream: write, wait, COPY KERNEL 512 KiB from the peer" :
                  variant == 5 ? "(f) exchange via a 2nd stream + 2 events, hipMemcpyAsync" :
                                 "(g) exchange via a 2nd stream + 2 events, copy kernel", 1e6 * dt / rounds, herr ? "  [SPIN TIMEOUT]" : "");
    fflush(stdout);
  }
  if (A) { int status = 0; waitpid(pid, &status, 0); return 0; }
  _exit(0);
}
