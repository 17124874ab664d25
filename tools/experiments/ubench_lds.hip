// tools/experiments/ubench_lds.hip — what an LDS instruction costs on gfx950 when sixteen waves of a CU issue them at
// random addresses (the shape of k_group1's tables): cycles per wave-instruction per CU, for reads of 4 / 8 / 16 bytes,
// returning and non-returning atomics, with 1..8 independent operations in flight per thread, all lanes active or a few.
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_lds tools/experiments/ubench_lds.hip && /tmp/ubench_lds
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }     \
  } while (0)

constexpr uint32_t kThreads = 1024, kWords = 32768;    // 128 KB of dynamic LDS: one workgroup per CU
constexpr uint32_t kRounds = 256;

__device__ __forceinline__ uint32_t next(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

// OP: 0 read b32, 1 read b64, 2 read b128, 3 add (no return), 4 add returning, 5 compare-and-swap returning,
//     6 add returning on 64-bit entries, 7 write b32, 8 nothing (the loop's own VALU work)
// ILP: independent operations a thread has in flight; ACTIVE: lanes of each wave that take part (the others idle)
template <int OP, int ILP>
__global__ __launch_bounds__(kThreads) void k_lds(uint32_t * sink, uint32_t active, unsigned long long * cycles) {
  extern __shared__ uint32_t lds[];
  for (uint32_t i = threadIdx.x; i < kWords; i += kThreads) { lds[i] = i * 2654435761u; }
  __syncthreads();
  uint32_t x[ILP], acc = 0;
#pragma unroll
  for (int k = 0; k < ILP; ++k) { x[k] = next(next((blockIdx.x * kThreads + threadIdx.x) * 977u + k * 0x9E3779B9u + 12345u)); }
  const bool on = (threadIdx.x & 63u) < active;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (on) {
    for (uint32_t r = 0; r < kRounds; ++r) {
      uint32_t v[ILP];
#pragma unroll
      for (int k = 0; k < ILP; ++k) {
        x[k] += 0x9E3779B1u;                                       // (one add: the addresses of a wave's lanes stay unrelated)
        if (OP == 0) { v[k] = lds[x[k] >> 17]; }
        if (OP == 1) { const uint2 q = *reinterpret_cast<const uint2 *>(lds + 2u * (x[k] >> 18)); v[k] = q.x ^ q.y; }
        if (OP == 2) { const uint4 q = *reinterpret_cast<const uint4 *>(lds + 4u * (x[k] >> 19)); v[k] = q.x ^ q.y ^ q.z ^ q.w; }
        if (OP == 3) { atomicAdd(&lds[x[k] >> 17], 1u); v[k] = 0; }
        if (OP == 4) { v[k] = atomicAdd(&lds[x[k] >> 17], 1u); }
        if (OP == 5) { v[k] = atomicCAS(&lds[x[k] >> 17], x[k], r); }
        if (OP == 6) { v[k] = (uint32_t)atomicAdd(reinterpret_cast<unsigned long long *>(lds) + (x[k] >> 18), 1ull); }
        if (OP == 7) { lds[x[k] >> 17] = r; v[k] = 0; }
        if (OP == 8) { v[k] = x[k] >> 17; }
      }
#pragma unroll
      for (int k = 0; k < ILP; ++k) { acc += v[k]; x[k] ^= v[k] & 0x10000u; }     // (the next address waits for this result: a chain per slot)
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; }
  if (acc == 0x1234567u) { sink[0] = acc + lds[5]; }
}

template <int OP, int ILP>
static void run(const char * name, uint32_t active, uint32_t * sink, unsigned long long * d_cycles, int cus) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds<OP, ILP>), hipFuncAttributeMaxDynamicSharedMemorySize, kWords * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_lds<OP, ILP>), dim3(cus), dim3(kThreads), kWords * 4, 0, sink, active, d_cycles);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_lds<OP, ILP>), dim3(cus), dim3(kThreads), kWords * 4, 0, sink, active, d_cycles);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // wave-instructions per CU: 16 waves x rounds x ILP
  const double instr = 16.0 * kRounds * ILP;
  const double ns_per_instr = (double)ms * 1e6 / instr;
  printf("{\"op\": \"%s\", \"in_flight\": %d, \"active_lanes\": %u, \"ms\": %.4f, \"ns_per_wave_instruction_per_cu\": %.2f, \"cycles_at_2.4GHz\": %.1f}\n",
         name, ILP, active, ms, ns_per_instr, ns_per_instr * 2.4);
}

template <int OP>
static void sweep(const char * name, uint32_t * sink, unsigned long long * c, int cus) {
  run<OP, 1>(name, 64, sink, c, cus);
  run<OP, 4>(name, 64, sink, c, cus);
  run<OP, 8>(name, 64, sink, c, cus);
  run<OP, 4>(name, 8, sink, c, cus);
  run<OP, 4>(name, 1, sink, c, cus);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  fprintf(stderr, "%s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
  uint32_t * sink; unsigned long long * cyc;
  CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&cyc, 8 * 1024));
  sweep<8>("no_lds_baseline", sink, cyc, cus);
  sweep<0>("read_b32", sink, cyc, cus);
  sweep<1>("read_b64", sink, cyc, cus);
  sweep<2>("read_b128", sink, cyc, cus);
  sweep<7>("write_b32", sink, cyc, cus);
  sweep<3>("add_u32", sink, cyc, cus);
  sweep<4>("add_rtn_u32", sink, cyc, cus);
  sweep<5>("cmpst_rtn_b32", sink, cyc, cus);
  sweep<6>("add_rtn_u64", sink, cyc, cus);
  return 0;
}
