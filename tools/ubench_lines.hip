// tools/ubench_lines.hip — the ceilings the d=1 step is graded against.
//
// The step's kernels are not streaming-bandwidth kernels: they read and write single cache lines at
// random addresses, issue device-scope atomics, or are bound by VALU issue.  SURVEY.md 8(d)'s byte
// model (8 bytes per microvariant) prices the reference's probing loop, which the pair kernels never
// run; so the roofline of each kernel class is MEASURED here, on the box the bench runs on:
//
//   stream      float4 copy                                     -> GB/s      (the HBM ceiling reached by code)
//   gather64    one random 64-byte line per access (4 x 16 B)   -> lines/s   (k_d1_group_pairs' member fetch)
//   gather8     one random  8-byte word per access              -> lines/s   (offset / cursor lookups)
//   scatter8/16 one random 8- / 16-byte store per access        -> lines/s   (record scatter)
//   atom_add64  returning device-scope atomicAdd(u64), random   -> ops/s     (slot tables, cursors)
//   atom_add32  non-returning device-scope atomicAdd(u32)       -> ops/s     (per-amplicon link counts)
//   valu        independent v_min_u32 / v_ffbl chains           -> wave-instructions/s (pair kernels)
//
// over working sets of 40 MB, 160 MB, 1.3 GB and 8 GB (L2 is 4 MB per XCD, the Infinity Cache 256 MB).
// Every kernel is also a calibration point for rocprofv3's FETCH_SIZE / WRITE_SIZE: its exact byte
// count is printed, so `rocprofv3 --pmc FETCH_SIZE` over this program gives the correction factor per
// access pattern (tools/calibrate_pmc.py).
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_lines tools/ubench_lines.hip
//   tools/ubench_lines [out.json] [quick]       (quick: stream copy, the 1 GB working set and the VALU test only: ~2 s)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }     \
  } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

constexpr int kPerThread = 16;     // accesses per thread, all independent (in flight together)

__global__ __launch_bounds__(256) void k_stream_copy(const uint4 * __restrict__ in, uint4 * __restrict__ out, uint64_t n16) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { out[i] = in[i]; }
}

// ... three more shapes of the same copy (the guide's float4 copy reaches 6.3 TB/s, the grid-stride loop above
// 4.8): one quad per thread and a grid as large as the array; four quads per thread, a workgroup's 16 KB contiguous, all
// loads before the stores; the same with non-temporal loads and stores.  The best of the four is reported as stream_copy.
__global__ __launch_bounds__(256) void k_stream_copy_flat(const uint4 * __restrict__ in, uint4 * __restrict__ out, uint64_t n16) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n16) { out[i] = in[i]; }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_stream_copy_chunk(const uint4 * __restrict__ in, uint4 * __restrict__ out, uint64_t n16) {
  for (uint64_t base = (uint64_t)blockIdx.x * 1024u; base < n16; base += (uint64_t)gridDim.x * 1024u) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t i = base + (uint64_t)k * 256u + threadIdx.x;
      if (i < n16) {
        if (NT) {
          v[k].x = __builtin_nontemporal_load(&in[i].x); v[k].y = __builtin_nontemporal_load(&in[i].y);
          v[k].z = __builtin_nontemporal_load(&in[i].z); v[k].w = __builtin_nontemporal_load(&in[i].w);
        } else { v[k] = in[i]; }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t i = base + (uint64_t)k * 256u + threadIdx.x;
      if (i < n16) {
        if (NT) {
          __builtin_nontemporal_store(v[k].x, &out[i].x); __builtin_nontemporal_store(v[k].y, &out[i].y);
          __builtin_nontemporal_store(v[k].z, &out[i].z); __builtin_nontemporal_store(v[k].w, &out[i].w);
        } else { out[i] = v[k]; }
      }
    }
  }
}

// one random 64-byte line per access: four consecutive lanes read its four 16-byte quarters (the shape of a
// member fetch: 40 bytes of sequence + record out of one line)
__global__ __launch_bounds__(256) void k_gather64(const uint4 * __restrict__ tab, uint64_t lines_mask, uint32_t * sink, uint32_t salt) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t group = t >> 2;
  const uint32_t quarter = (uint32_t)t & 3u;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const uint64_t line = mix(group * kPerThread + k + ((uint64_t)salt << 40)) & lines_mask;
    const uint4 v = tab[line * 4u + quarter];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345u) { sink[0] = acc; }
}

// every lane its own random line, 16 bytes of it (what a lane-per-member gather looks like)
__global__ __launch_bounds__(256) void k_gather16(const uint4 * __restrict__ tab, uint64_t lines_mask, uint32_t * sink, uint32_t salt) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const uint64_t line = mix(t * kPerThread + k + ((uint64_t)salt << 40)) & lines_mask;
    const uint4 v = tab[line * 4u];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345u) { sink[0] = acc; }
}

__global__ __launch_bounds__(256) void k_gather8(const uint64_t * __restrict__ tab, uint64_t words_mask, uint32_t * sink, uint32_t salt) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) { acc ^= tab[mix(t * kPerThread + k + ((uint64_t)salt << 40)) & words_mask]; }
  if (acc == 0x12345ull) { sink[0] = (uint32_t)acc; }
}

__global__ __launch_bounds__(256) void k_scatter8(uint64_t * tab, uint64_t words_mask, uint32_t salt) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) { tab[mix(t * kPerThread + k + ((uint64_t)salt << 40)) & words_mask] = t; }
}

__global__ __launch_bounds__(256) void k_scatter16(uint4 * tab, uint64_t q_mask, uint32_t salt) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) { tab[mix(t * kPerThread + k + ((uint64_t)salt << 40)) & q_mask] = make_uint4((uint32_t)t, k, 0u, 0u); }
}

__global__ __launch_bounds__(256) void k_atom_add64(unsigned long long * tab, uint64_t words_mask, uint32_t * sink, uint32_t salt) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) { acc += atomicAdd(&tab[mix(t * kPerThread + k + ((uint64_t)salt << 40)) & words_mask], 1ull); }
  if (acc == 0xFFFFFFFFFFFFull) { sink[0] = 1u; }
}

__global__ __launch_bounds__(256) void k_atom_add32(uint32_t * tab, uint64_t words_mask, uint32_t salt) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) { atomicAdd(&tab[mix(t * kPerThread + k + ((uint64_t)salt << 40)) & words_mask], 1u); }
}

// VALU issue: eight independent chains of three instructions per step, all written out (nothing for the compiler to fold):
//   KIND 0  v_xor + v_ffbl + v_min   (the pair test's own instructions)
//   KIND 1  v_xor + v_and + v_add    (plain full-rate integer instructions: the issue rate itself)
//   KIND 2  v_xor + v_ffbl + v_ffbh  (is the bit scan a full-rate instruction?)
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(uint32_t * sink, uint32_t rounds, uint32_t seed) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 1u) + (uint32_t)i; }
  for (uint32_t r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t x, f;
      asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(a[i]), "v"(r));
      if (KIND == 0) {
        asm volatile("v_ffbl_b32 %0, %1" : "=v"(f) : "v"(x));
        asm volatile("v_min_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
      } else if (KIND == 1) {
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(f) : "v"(x), "v"(0x7FFFFFFFu));
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
      } else {
        asm volatile("v_ffbl_b32 %0, %1" : "=v"(f) : "v"(x));
        asm volatile("v_ffbh_u32 %0, %1" : "=v"(a[i]) : "v"(f));
      }
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc ^= a[i]; }
  if (acc == 0x12345u) { sink[0] = acc; }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
  template <class F> double best_ms(F && launch, int reps = 5) {
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
      CHECK(hipEventRecord(a, nullptr));
      launch(r);
      CHECK(hipEventRecord(b, nullptr));
      CHECK(hipEventSynchronize(b));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, a, b));
      if (r > 0 && ms < best) { best = ms; }     // (the first repetition warms up)
    }
    return best;
  }
};

int main(int argc, char ** argv) {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  Timer tm;
  uint32_t * sink;
  CHECK(hipMalloc(&sink, 64));
  std::string json = "{\"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(cus) + ", \"results\": [";
  bool first = true;
  auto emit = [&](const char * name, double set_mb, double ms, double ops, const char * unit, double bytes_read, double bytes_written) {
    char buf[512];
    snprintf(buf, sizeof buf,
             "%s{\"test\": \"%s\", \"working_set_mb\": %.0f, \"ms\": %.4f, \"rate\": %.4e, \"unit\": \"%s\", \"bytes_read\": %.0f, \"bytes_written\": %.0f}",
             first ? "" : ", ", name, set_mb, ms, ops / (ms * 1e-3), unit, bytes_read, bytes_written);
    json += buf;
    first = false;
    fprintf(stderr, "%-12s %8.0f MB  %9.4f ms  %.3e %s\n", name, set_mb, ms, ops / (ms * 1e-3), unit);
  };

  // ---- streaming copy (2 GB in, 2 GB out)
  {
    const uint64_t bytes = 2ull << 30;
    uint4 *in, *out;
    CHECK(hipMalloc(&in, bytes)); CHECK(hipMalloc(&out, bytes));
    CHECK(hipMemset(in, 1, bytes));
    const uint64_t n16 = bytes / 16;
    double ms = tm.best_ms([&](int) { hipLaunchKernelGGL(k_stream_copy, dim3(cus * 16), dim3(256), 0, nullptr, in, out, n16); });
    emit("stream_copy_grid_stride", 2.0 * bytes / 1e6, ms, 2.0 * bytes / 1e9, "GB/s", (double)bytes, (double)bytes);
    const double ms_flat = tm.best_ms([&](int) { hipLaunchKernelGGL(k_stream_copy_flat, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, nullptr, in, out, n16); });
    emit("stream_copy_flat", 2.0 * bytes / 1e6, ms_flat, 2.0 * bytes / 1e9, "GB/s", (double)bytes, (double)bytes);
    const double ms_chunk = tm.best_ms([&](int) { hipLaunchKernelGGL(k_stream_copy_chunk<false>, dim3(cus * 32), dim3(256), 0, nullptr, in, out, n16); });
    emit("stream_copy_chunks", 2.0 * bytes / 1e6, ms_chunk, 2.0 * bytes / 1e9, "GB/s", (double)bytes, (double)bytes);
    const double ms_nt = tm.best_ms([&](int) { hipLaunchKernelGGL(k_stream_copy_chunk<true>, dim3(cus * 32), dim3(256), 0, nullptr, in, out, n16); });
    emit("stream_copy_chunks_nontemporal", 2.0 * bytes / 1e6, ms_nt, 2.0 * bytes / 1e9, "GB/s", (double)bytes, (double)bytes);
    ms = std::min(std::min(ms, ms_flat), std::min(ms_chunk, ms_nt));
    emit("stream_copy", 2.0 * bytes / 1e6, ms, 2.0 * bytes / 1e9, "GB/s", (double)bytes, (double)bytes);
    CHECK(hipFree(in)); CHECK(hipFree(out));
  }

  // ---- random accesses over several working sets
  const bool quick = argc > 2 && std::string(argv[2]) == "quick";
  const uint64_t sets[] = {32ull << 20, 128ull << 20, 1ull << 30, 8ull << 30};     // powers of two (masks)
  for (uint64_t bytes : sets) {
    if (quick && bytes != (1ull << 30)) { continue; }
    void * tab;
    if (hipMalloc(&tab, bytes) != hipSuccess) { continue; }
    CHECK(hipMemset(tab, 0, bytes));
    const double mb = bytes / 1e6;
    const uint64_t accesses = 64ull << 20;                                       // per launch
    const uint32_t threads = (uint32_t)(accesses / kPerThread);
    {
      const double ms = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_gather64, dim3(threads * 4 / 256), dim3(256), 0, nullptr, (const uint4 *)tab, bytes / 64 - 1, sink, (uint32_t)r); });
      emit("gather64", mb, ms, (double)accesses, "lines/s", 64.0 * accesses, 0);
    }
    {
      const double ms = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_gather16, dim3(threads / 256), dim3(256), 0, nullptr, (const uint4 *)tab, bytes / 64 - 1, sink, (uint32_t)r); });
      emit("gather16", mb, ms, (double)accesses, "lines/s", 16.0 * accesses, 0);
    }
    {
      const double ms = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_gather8, dim3(threads / 256), dim3(256), 0, nullptr, (const uint64_t *)tab, bytes / 8 - 1, sink, (uint32_t)r); });
      emit("gather8", mb, ms, (double)accesses, "lines/s", 8.0 * accesses, 0);
    }
    {
      const double ms = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_scatter8, dim3(threads / 256), dim3(256), 0, nullptr, (uint64_t *)tab, bytes / 8 - 1, (uint32_t)r); });
      emit("scatter8", mb, ms, (double)accesses, "lines/s", 0, 8.0 * accesses);
    }
    {
      const double ms = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_scatter16, dim3(threads / 256), dim3(256), 0, nullptr, (uint4 *)tab, bytes / 16 - 1, (uint32_t)r); });
      emit("scatter16", mb, ms, (double)accesses, "lines/s", 0, 16.0 * accesses);
    }
    {
      const double ms = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_atom_add64, dim3(threads / 256), dim3(256), 0, nullptr, (unsigned long long *)tab, bytes / 8 - 1, sink, (uint32_t)r); });
      emit("atom_add64", mb, ms, (double)accesses, "ops/s", 8.0 * accesses, 8.0 * accesses);
    }
    {
      const double ms = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_atom_add32, dim3(threads / 256), dim3(256), 0, nullptr, (uint32_t *)tab, bytes / 4 - 1, (uint32_t)r); });
      emit("atom_add32", mb, ms, (double)accesses, "ops/s", 4.0 * accesses, 4.0 * accesses);
    }
    CHECK(hipFree(tab));
  }

  // ---- VALU issue
  {
    const uint32_t rounds = 4096;
    const int blocks = cus * 8;
    const double wave_insts = (double)blocks * 4.0 * rounds * 8.0 * 3.0;       // three instructions per chain step
    const double ms = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, nullptr, sink, rounds, (uint32_t)r + 1u); });
    emit("valu_3op", 0, ms, wave_insts, "wave-instructions/s (v_xor + v_ffbl + v_min, eight independent chains)", 0, 0);
    const double ms2 = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, nullptr, sink, rounds, (uint32_t)r + 1u); });
    emit("valu_simple", 0, ms2, wave_insts, "wave-instructions/s (v_xor + v_and + v_add, eight independent chains)", 0, 0);
    const double ms3 = tm.best_ms([&](int r) { hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, nullptr, sink, rounds, (uint32_t)r + 1u); });
    emit("valu_bitscan", 0, ms3, wave_insts, "wave-instructions/s (v_xor + v_ffbl + v_ffbh, eight independent chains)", 0, 0);
  }
  json += "]}";
  if (argc > 1) {
    FILE * f = fopen(argv[1], "w");
    if (f != nullptr) { fputs(json.c_str(), f); fputs("\n", f); fclose(f); }
  }
  puts(json.c_str());
  return 0;
}
