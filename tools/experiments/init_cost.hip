// init_cost.hip — where the HIP runtime's start-up goes on this box, call by call (ms since process start), and what
// process exit costs afterwards.  Usage: init_cost [host_MB] [device_MB] [thp]  — touch host_MB of anonymous memory
// (thp: madvise(MADV_HUGEPAGE) on it), hipMalloc + memset device_MB, then _Exit; the caller times the whole process.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
static void stamp(const char * what) {
  std::fprintf(stderr, "[%8.2f ms] %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
}
__global__ void k_touch(uint32_t * p) { if (threadIdx.x == 0) { p[0] = 1; } }
int main(int argc, char ** argv) {
  const size_t host_mb = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 0;
  const size_t dev_mb = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 0;
  const bool thp = argc > 3 && argv[3][0] == '1';
  const bool nohip = argc > 4 && argv[4][0] == '1';
  stamp("main");
  char * host = nullptr;
  if (host_mb != 0) {
    const size_t bytes = host_mb << 20;
    host = static_cast<char *>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (thp) { madvise(host, bytes, MADV_HUGEPAGE); }
    for (size_t i = 0; i < bytes; i += 4096) { host[i] = 1; }
    stamp("host memory touched");
  }
  if (!nohip) {
    hipInit(0); stamp("hipInit");
    int n = 0; hipGetDeviceCount(&n); stamp("hipGetDeviceCount");
    hipSetDevice(0); stamp("hipSetDevice");
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); stamp("hipGetDeviceProperties");
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); stamp("hipStreamCreate");
    void * d = nullptr; hipMalloc(&d, 4096); stamp("hipMalloc 4 KB");
    hipMemsetAsync(d, 0, 4096, s); hipStreamSynchronize(s); stamp("memset + sync");
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, static_cast<uint32_t *>(d)); hipStreamSynchronize(s); stamp("first kernel + sync");
    void * big = nullptr; hipMalloc(&big, 64u << 20); stamp("hipMalloc 64 MB");
    if (host != nullptr) { hipMemcpyAsync(big, host, 32u << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); stamp("H2D 32 MB pageable"); }
    if (dev_mb != 0) {
      void * dd = nullptr; hipMalloc(&dd, dev_mb << 20); hipMemsetAsync(dd, 1, dev_mb << 20, s); hipStreamSynchronize(s); stamp("device memory allocated + set");
    }
  }
  stamp("leaving");
  std::_Exit(0);
}
