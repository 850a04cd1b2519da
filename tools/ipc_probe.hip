// ipc_probe.hip — what inter-process primitives work on this box (2 processes, ONE GPU or two):
//   (1) hipIpcGetMemHandle / hipIpcOpenMemHandle of a hipMalloc'ed buffer, peer read by a copy and by a kernel
//   (2) hipIpcGetEventHandle / hipIpcOpenEventHandle: stream of process B waits for an event recorded by process A
//   (3) hipStreamWriteValue64 / hipStreamWaitValue64 on a host-registered POSIX shared-memory page
// Usage: ipc_probe [devA devB]   (forks; parent = producer A, child = consumer B)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("[%s] %s failed: %s (line %d)\n", who, #e, hipGetErrorString(r_), __LINE__); fflush(stdout); return false; } } while (0)

struct Shared {
  std::atomic<int> stage;
  std::atomic<int> b_stage;
  hipIpcMemHandle_t mem;
  hipIpcEventHandle_t ev;
  int ev_ok;
  std::atomic<uint64_t> flag;   // stream write/wait value target
  std::atomic<int> fail;
};

__global__ void fill(double* p, int n, double v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void slow_fill(double* p, int n, double v, long spin) {
  long t0 = clock64();
  while (clock64() - t0 < spin) {}
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i;
}
__global__ void copyk(const double* s, double* d, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = s[i]; }

static bool wait_stage(std::atomic<int>& a, int v, std::atomic<int>& fail) {
  auto t0 = std::chrono::steady_clock::now();
  while (a.load() < v) {
    if (fail.load()) return false;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) return false;
    usleep(50);
  }
  return true;
}

static bool producer(Shared* sh, int devA, int n) {
  const char* who = "A";
    double* buf = nullptr; hipEvent_t ev = nullptr; hipStream_t st = nullptr;
    CK(hipSetDevice(devA));
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipMalloc(&buf, 8 * n));
    fill<<<n / 256, 256, 0, st>>>(buf, n, 1000.0);
    CK(hipStreamSynchronize(st));
    CK(hipIpcGetMemHandle(&sh->mem, buf));
    {
      hipError_t e = hipEventCreateWithFlags(&ev, hipEventInterprocess | hipEventDisableTiming);
      if (e == hipSuccess) e = hipIpcGetEventHandle(&sh->ev, ev);
      sh->ev_ok = e == hipSuccess;
      printf("[A] ipc event export: %s\n", hipGetErrorString(e));
    }
    sh->stage = 1;                                   // handles published
    if (!wait_stage(sh->b_stage, 1, sh->fail)) { printf("[A] B never opened\n"); return false; }
    // (2) event: a slow kernel rewrites the buffer, then the event; B's stream must see the new values
    if (sh->ev_ok) {
      slow_fill<<<n / 256, 256, 0, st>>>(buf, n, 2000.0, 200000000L);   // ~0.1 s
      CK(hipEventRecord(ev, st));
      sh->stage = 2;                                 // "record has been called"
      if (!wait_stage(sh->b_stage, 2, sh->fail)) { printf("[A] B stage 2 timeout\n"); return false; }
    } else { sh->stage = 2; wait_stage(sh->b_stage, 2, sh->fail); }
    // (3) stream write value on registered shared host memory
    {
      void* dflag = nullptr;
      hipError_t e = hipHostRegister((void*)&sh->flag, 64, hipHostRegisterMapped);
      if (e == hipSuccess) e = hipHostGetDevicePointer(&dflag, (void*)&sh->flag, 0);
      printf("[A] host register shm: %s\n", hipGetErrorString(e));
      if (e == hipSuccess) {
        slow_fill<<<n / 256, 256, 0, st>>>(buf, n, 3000.0, 200000000L);
        e = hipStreamWriteValue64(st, dflag, 7, 0);
        printf("[A] hipStreamWriteValue64: %s\n", hipGetErrorString(e));
        if (e != hipSuccess) sh->flag = 7;
      } else sh->flag = 7;
      sh->stage = 3;
      wait_stage(sh->b_stage, 3, sh->fail);
      hipStreamSynchronize(st);
    }
    return true;
}

static bool consumer(Shared* sh, int devB, int n) {
  const char* who = "B";
    double *peer = nullptr, *mine = nullptr; hipStream_t st = nullptr; hipEvent_t ev = nullptr;
    double* h = (double*)malloc(8 * n);
    CK(hipSetDevice(devB));
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipMalloc(&mine, 8 * n));
    if (!wait_stage(sh->stage, 1, sh->fail)) { printf("[B] no handles\n"); return false; }
    CK(hipIpcOpenMemHandle((void**)&peer, sh->mem, hipIpcMemLazyEnablePeerAccess));
    CK(hipMemcpyAsync(mine, peer, 8 * n, hipMemcpyDeviceToDevice, st));
    CK(hipMemcpyAsync(h, mine, 8 * n, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("[B] (1) peer copy: h[5]=%.1f (want 1005.0) %s\n", h[5], h[5] == 1005.0 ? "OK" : "BAD");
    copyk<<<n / 256, 256, 0, st>>>(peer, mine, n);
    CK(hipMemcpyAsync(h, mine, 8 * n, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("[B] (1) peer kernel read: h[n-1]=%.1f (want %.1f) %s\n", h[n - 1], 1000.0 + n - 1, h[n - 1] == 1000.0 + n - 1 ? "OK" : "BAD");
    if (sh->ev_ok) {
      hipError_t e = hipIpcOpenEventHandle(&ev, sh->ev);
      printf("[B] ipc event open: %s\n", hipGetErrorString(e));
      sh->b_stage = 1;
      wait_stage(sh->stage, 2, sh->fail);
      if (e == hipSuccess) {
        auto t0 = std::chrono::steady_clock::now();
        e = hipStreamWaitEvent(st, ev, 0);
        printf("[B] hipStreamWaitEvent(ipc): %s\n", hipGetErrorString(e));
        CK(hipMemcpyAsync(mine, peer, 8 * n, hipMemcpyDeviceToDevice, st));
        CK(hipMemcpyAsync(h, mine, 8 * n, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("[B] (2) after ipc event wait: h[5]=%.1f (want 2005.0) %s, waited %.1f ms\n", h[5], h[5] == 2005.0 ? "OK" : "BAD (event did not order)", ms);
      }
    } else { sh->b_stage = 1; wait_stage(sh->stage, 2, sh->fail); }
    sh->b_stage = 2;
    wait_stage(sh->stage, 3, sh->fail);
    {
      void* dflag = nullptr;
      hipError_t e = hipHostRegister((void*)&sh->flag, 64, hipHostRegisterMapped);
      if (e == hipSuccess) e = hipHostGetDevicePointer(&dflag, (void*)&sh->flag, 0);
      printf("[B] host register shm: %s\n", hipGetErrorString(e));
      int can = 0;
      hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, devB);
      printf("[B] CanUseStreamWaitValue=%d\n", can);
      if (e == hipSuccess && can) {
        auto t0 = std::chrono::steady_clock::now();
        e = hipStreamWaitValue64(st, dflag, 7, hipStreamWaitValueGte, ~0ull);
        printf("[B] hipStreamWaitValue64: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) {
          CK(hipMemcpyAsync(mine, peer, 8 * n, hipMemcpyDeviceToDevice, st));
          CK(hipMemcpyAsync(h, mine, 8 * n, hipMemcpyDeviceToHost, st));
          CK(hipStreamSynchronize(st));
          double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          printf("[B] (3) after stream wait value: h[5]=%.1f (want 3005.0) %s, waited %.1f ms\n", h[5], h[5] == 3005.0 ? "OK" : "BAD", ms);
        }
      }
    }
    sh->b_stage = 3;
    hipIpcCloseMemHandle(peer);
    return true;
}

int main(int argc, char** argv) {
  int devA = argc > 2 ? atoi(argv[1]) : 0, devB = argc > 2 ? atoi(argv[2]) : 0;
  const int n = 1 << 16;
  Shared* sh = (Shared*)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  new (sh) Shared();
  sh->stage = 0; sh->b_stage = 0; sh->flag = 0; sh->fail = 0; sh->ev_ok = 0;
  pid_t pid = fork();
  if (pid) {
    bool ok = producer(sh, devA, n);
    if (!ok) sh->fail = 1;
    int status = 0;
    waitpid(pid, &status, 0);
    printf("[A] done ok=%d child=%d\n", (int)ok, WEXITSTATUS(status));
    return ok && WEXITSTATUS(status) == 0 ? 0 : 1;
  }
  bool ok = consumer(sh, devB, n);
  if (!ok) sh->fail = 1;
  sh->b_stage = 99;
  fflush(stdout);
  _exit(ok ? 0 : 1);
}
