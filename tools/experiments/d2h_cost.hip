// d2h_cost.hip — what a 40 MB download costs by the kind of host memory it goes to (lease r6d): pageable (fresh / touched),
// hipHostRegister'ed, hipHostMalloc'ed (default / non-coherent), and a kernel storing straight into mapped host memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_copy(const uint4 * __restrict__ src, uint4 * __restrict__ dst, size_t quads) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) { dst[i] = src[i]; }
}
int main(int argc, char ** argv) {
  const size_t bytes = (argc > 1 ? std::atol(argv[1]) : 40) << 20;
  hipStream_t s; CK(hipStreamCreate(&s));
  void * d = nullptr; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes)); CK(hipDeviceSynchronize());
  auto copy2 = [&](const char * what, void * h) {
    for (int r = 0; r < 3; ++r) {
      const double t0 = now();
      if (hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { std::printf("copy failed\n"); return; }
      std::printf("  %-34s copy %d: %7.3f ms  (%.1f GB/s)\n", what, r, now() - t0, bytes / (now() - t0) / 1e6);
    }
  };
  if (argc > 2) {   // what comes FIRST in a process: "r" a registered buffer, "s" a 64 KB pinned warm-up copy and then the registered buffer
    if (argv[2][0] == 's') { void * w = nullptr; CK(hipHostMalloc(&w, 65536, 0)); double t0 = now(); CK(hipMemcpyAsync(w, d, 65536, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); std::printf("64 KB warm-up copy: %.3f ms\n", now() - t0); }
    void * h = aligned_alloc(2 << 20, bytes); double t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); std::printf("FIRST: registered: register %.3f ms\n", now() - t0);
    for (int r = 0; r < 2; ++r) { t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); const double t1 = now(); CK(hipStreamSynchronize(s)); std::printf("  first registered copy %d: call %.3f ms, wait %.3f ms\n", r, t1 - t0, now() - t1); }
  }
  { double t0 = now(); void * h = aligned_alloc(2 << 20, bytes); std::printf("pageable, fresh: alloc %.3f ms\n", now() - t0); copy2("pageable fresh", h); free(h); }
  { void * h = aligned_alloc(2 << 20, bytes); double t0 = now(); std::memset(h, 0, bytes); std::printf("pageable, touched: memset %.3f ms\n", now() - t0); copy2("pageable touched", h); free(h); }
  { void * h = aligned_alloc(2 << 20, bytes); double t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); std::printf("registered (untouched): register %.3f ms\n", now() - t0);
    copy2("registered", h); t0 = now(); CK(hipHostUnregister(h)); std::printf("  unregister %.3f ms\n", now() - t0); free(h); }
  { void * h = aligned_alloc(2 << 20, bytes); double t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterMapped | hipHostRegisterPortable)); std::printf("registered mapped|portable: register %.3f ms\n", now() - t0);
    copy2("registered mapped", h);
    void * dp = nullptr; CK(hipHostGetDevicePointer(&dp, h, 0));
    for (int r = 0; r < 3; ++r) { t0 = now(); hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, (const uint4 *)d, (uint4 *)dp, bytes / 16); CK(hipStreamSynchronize(s)); std::printf("  kernel store into registered: %7.3f ms (%.1f GB/s)\n", now() - t0, bytes / (now() - t0) / 1e6); }
    CK(hipHostUnregister(h)); free(h); }
  for (unsigned flags : {(unsigned)hipHostMallocDefault, (unsigned)hipHostMallocNonCoherent, (unsigned)(hipHostMallocMapped | hipHostMallocPortable)}) {
    void * h = nullptr; double t0 = now(); CK(hipHostMalloc(&h, bytes, flags)); std::printf("hipHostMalloc flags %u: alloc %.3f ms\n", flags, now() - t0);
    copy2("hipHostMalloc", h);
    void * dp = nullptr; CK(hipHostGetDevicePointer(&dp, h, 0));
    for (int r = 0; r < 3; ++r) { t0 = now(); hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, (const uint4 *)d, (uint4 *)dp, bytes / 16); CK(hipStreamSynchronize(s)); std::printf("  kernel store into hipHostMalloc: %7.3f ms (%.1f GB/s)\n", now() - t0, bytes / (now() - t0) / 1e6); }
    t0 = now(); CK(hipHostFree(h)); std::printf("  free %.3f ms\n", now() - t0);
  }
  // chunked: a registered buffer filled by four copies on two streams
  { void * h = aligned_alloc(2 << 20, bytes); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); hipStream_t s2; CK(hipStreamCreate(&s2));
    for (int r = 0; r < 3; ++r) { double t0 = now();
      for (int c = 0; c < 4; ++c) { CK(hipMemcpyAsync((char *)h + bytes / 4 * c, (char *)d + bytes / 4 * c, bytes / 4, hipMemcpyDeviceToHost, c & 1 ? s2 : s)); }
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); std::printf("  registered, 4 chunks on 2 streams: %7.3f ms\n", now() - t0); }
    CK(hipHostUnregister(h)); free(h); }
  return 0;
}
