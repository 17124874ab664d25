// scan.hip — the fused d >= 2 step: one (sub)seed against the whole unswarmed pool, on the GPU.
//
// The reference's greedy loop (src/algo.cc:384-676) builds, for every seed and sub-seed, a
// candidate list on the host (abundance rule + triangle-inequality prune
// `diffestimate <= radius + d`, src/algo.cc:423-431, 515-531), calls qgram_diff_fast, filters
// `qdiff <= d` on the host, calls search_do, and filters `diff <= d` on the host.  Done through
// the per-call seams B3/B4 that costs four host<->device transfers per step.  Here the pool
// state lives in HBM (est[] = q-gram estimate against the current swarm's initial seed,
// swarmed[]) and one step is three kernels and ONE small read-back:
//
//   k_scan_filter   8 lanes per pool amplicon: prune -> 128-byte signature gather -> popcount
//                   bound -> (first generation: store est) -> compact survivors
//   k_align<G>      the B4 kernel over the compacted targets (count read on the device)
//   k_scan_collect  diff <= d -> (id, diff) hit list, swarmed[id] = 1
//
// The unswarmed pool is always in ascending amplicon-id order in the reference (rotations in
// move_target_to_first_unswarmed_position keep the relative order, src/algo.cc:222-245), so
// "pool order" = id order and the host only needs the hits sorted by id.
#include "swa_internal.h"

#include <algorithm>

int swa_align_launch(swa_ctx * ctx, uint32_t query, const uint32_t * d_targets, const uint32_t * d_count,
                     uint32_t max_count, uint32_t * d_diffs, uint32_t * d_scores, uint32_t * d_alnlens);

namespace {

constexpr uint32_t kInlineHits = 1022;   // hits returned with the first read-back

struct ScanArgs {
  const ulonglong2 * sigs;
  const uint64_t * abundance;
  uint32_t * est;
  uint8_t * swarmed;
  uint32_t n, lo, seed;
  uint32_t first_generation, limit /* radius + d */, d, ncb;
  uint32_t * targets;
  uint32_t * counters;     // [0] targets [1] hits [2..3] unused
  unsigned long long * totals;   // [0] q-gram comparisons [1] aligned pairs [2] accepted
};

__global__ __launch_bounds__(256) void k_scan_filter(const ScanArgs a) {
  const uint32_t sub = threadIdx.x & 7u;
  const ulonglong2 mine = a.sigs[(uint64_t)a.seed * 8u + sub];
  const uint64_t seed_ab = a.abundance[a.seed];
  const uint32_t groups = gridDim.x * (blockDim.x >> 3);
  unsigned long long compared = 0;
  for (uint32_t i = a.lo + blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3); i < a.n; i += groups) {
    if (i == a.seed || a.swarmed[i] != 0) { continue; }
    if (a.first_generation == 0u && a.est[i] > a.limit) { continue; }          // algo.cc:521-522
    if (a.ncb == 0u && a.abundance[i] > seed_ab) { continue; }                  // algo.cc:427-428, 523-525
    const ulonglong2 other = a.sigs[(uint64_t)i * 8u + sub];
    uint32_t pop = (uint32_t)__popcll(mine.x ^ other.x) + (uint32_t)__popcll(mine.y ^ other.y);
    pop += __shfl_xor(pop, 1, 8);
    pop += __shfl_xor(pop, 2, 8);
    pop += __shfl_xor(pop, 4, 8);
    if (sub == 0u) {
      const uint32_t qd = (pop + 9u) / 10u;
      ++compared;
      if (a.first_generation != 0u) { a.est[i] = qd; }                          // algo.cc:442
      if (qd <= a.d) { a.targets[atomicAdd(&a.counters[0], 1u)] = i; }
    }
  }
  if (compared != 0ull) { atomicAdd(&a.totals[0], compared); }
}

__global__ __launch_bounds__(256) void k_scan_collect(const uint32_t * __restrict__ targets,
                                                      const uint32_t * __restrict__ diffs, uint32_t * counters,
                                                      uint32_t d, uint8_t * swarmed, uint32_t * __restrict__ hits,
                                                      unsigned long long * totals) {
  const uint32_t nt = counters[0];
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += gridDim.x * blockDim.x) {
    if (diffs[t] <= d) {
      const uint32_t at = atomicAdd(&counters[1], 1u);
      hits[2u * at] = targets[t];
      hits[2u * at + 1u] = diffs[t];
      swarmed[targets[t]] = 1;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&totals[1], (unsigned long long)nt); }
}

}  // namespace

extern "C" int swa_scan_begin(swa_ctx * ctx) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (!ctx->qgram_ready || !ctx->search_ready) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_scan_begin: call swa_qgram_build and swa_search_begin first");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t n = ctx->db.n;
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_est, n * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_swarmed, n));
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_targets, n * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_diffs, n * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_hits, (2 * n + 2 * kInlineHits + 8) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_counters, 64));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_scan_est.ptr, 0, n * sizeof(uint32_t), ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_scan_swarmed.ptr, 0, n, ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_scan_counters.ptr, 0, 64, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scan_ready = true;
  return SWA_OK;
}

extern "C" int swa_scan_step(swa_ctx * ctx, uint32_t seed, uint32_t lowest_unswarmed, int first_generation,
                             uint32_t radius, int no_cluster_breaking, uint32_t * hit_ids, uint32_t * hit_diffs,
                             uint32_t cap, uint32_t * nhits) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (!ctx->scan_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_scan_step: call swa_scan_begin first"); }
  if (seed >= ctx->db.n || nhits == nullptr || (cap != 0 && (hit_ids == nullptr || hit_diffs == nullptr))) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_scan_step: bad argument");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->db.n;
  auto * counters = static_cast<uint32_t *>(ctx->d_scan_counters.ptr);          // u32[4] then u64 totals[4]
  auto * totals = reinterpret_cast<unsigned long long *>(counters + 4);
  auto * hits = static_cast<uint32_t *>(ctx->d_scan_hits.ptr);
  ScanArgs a{};
  a.sigs = static_cast<const ulonglong2 *>(ctx->d_qgrams.ptr);
  a.abundance = ctx->db.abundance;
  a.est = static_cast<uint32_t *>(ctx->d_scan_est.ptr);
  a.swarmed = static_cast<uint8_t *>(ctx->d_scan_swarmed.ptr);
  a.n = n;
  a.lo = lowest_unswarmed < n ? lowest_unswarmed : n;
  a.seed = seed;
  a.first_generation = first_generation != 0 ? 1u : 0u;
  a.limit = radius + (uint32_t)ctx->resolution;
  a.d = (uint32_t)ctx->resolution;
  a.ncb = no_cluster_breaking != 0 ? 1u : 0u;
  a.targets = static_cast<uint32_t *>(ctx->d_scan_targets.ptr);
  a.counters = counters;
  a.totals = totals;
  SWA_HIP(ctx, hipMemsetAsync(counters, 0, 4 * sizeof(uint32_t), ctx->stream));
  if (a.first_generation != 0u) {                      // the initial seed joins its own swarm
    SWA_HIP(ctx, hipMemsetAsync(a.swarmed + seed, 1, 1, ctx->stream));
  }
  const uint32_t span = n - a.lo;
  if (span > 0) {
    uint64_t blocks = ((uint64_t)span + 31) / 32;
    const uint64_t gcap = uint64_t(ctx->num_cus) * 8;
    if (blocks > gcap) { blocks = gcap; }
    hipLaunchKernelGGL(k_scan_filter, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    SWA_HIP(ctx, hipGetLastError());
    SWA_TRY(swa_align_launch(ctx, seed, a.targets, counters, span, static_cast<uint32_t *>(ctx->d_scan_diffs.ptr),
                             nullptr, nullptr));
    hipLaunchKernelGGL(k_scan_collect, dim3(64), dim3(256), 0, ctx->stream, a.targets,
                       static_cast<const uint32_t *>(ctx->d_scan_diffs.ptr), counters, a.d, a.swarmed, hits + 2, totals);
    SWA_HIP(ctx, hipGetLastError());
  }
  // one read-back: [target count, hit count] + the first kInlineHits (id, diff) pairs
  SWA_HIP(ctx, hipMemcpyAsync(hits, counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
  ctx->scan_host.resize(2 + 2 * kInlineHits);
  SWA_HIP(ctx, hipMemcpyAsync(ctx->scan_host.data(), hits, ctx->scan_host.size() * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t got = ctx->scan_host[1];
  *nhits = got;
  if (got > kInlineHits) {
    ctx->scan_host.resize(2 + 2 * (size_t)got);
    SWA_HIP(ctx, hipMemcpyAsync(ctx->scan_host.data() + 2, hits + 2, 2 * (size_t)got * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (got > cap) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_scan_step: hit buffer too small"); }
  // pool order = ascending amplicon id
  ctx->scan_sorted.resize(got);
  const uint32_t * pairs = ctx->scan_host.data() + 2;
  for (uint32_t k = 0; k < got; ++k) { ctx->scan_sorted[k] = ((uint64_t)pairs[2 * k] << 32) | pairs[2 * k + 1]; }
  std::sort(ctx->scan_sorted.begin(), ctx->scan_sorted.end());
  for (uint32_t k = 0; k < got; ++k) {
    hit_ids[k] = (uint32_t)(ctx->scan_sorted[k] >> 32);
    hit_diffs[k] = (uint32_t)ctx->scan_sorted[k];
  }
  totals = nullptr;
  return SWA_OK;
}

extern "C" int swa_scan_totals(swa_ctx * ctx, uint64_t * out3) {
  if (ctx == nullptr || out3 == nullptr) { return SWA_E_ARG; }
  if (!ctx->scan_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_scan_totals: no scan state"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t t[4] = {};
  SWA_HIP(ctx, hipMemcpyAsync(t, static_cast<uint32_t *>(ctx->d_scan_counters.ptr) + 4, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  out3[0] = t[0]; out3[1] = t[1]; out3[2] = t[2];
  return SWA_OK;
}
