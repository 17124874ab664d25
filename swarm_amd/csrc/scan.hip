// scan.hip — the fused d >= 2 step: seeds against the whole unswarmed pool, on the GPU.
//
// The reference's greedy loop (src/algo.cc:384-676) builds, for every seed and sub-seed, a
// candidate list on the host (abundance rule + triangle-inequality prune
// `diffestimate <= radius + d`, src/algo.cc:423-431, 515-531), calls qgram_diff_fast, filters
// `qdiff <= d` on the host, calls search_do, and filters `diff <= d` on the host.  Done through
// the per-call seams B3/B4 that costs four host<->device transfers per step.  Here the pool
// state lives in HBM (est[] = q-gram estimate against the current swarm's initial seed,
// swarmed[]) and one launch sequence handles a whole BATCH of sub-seeds:
//
//   k_scan_filter   grid.y = seed; 8 lanes per pool amplicon: prune -> 128-byte signature
//                   gather -> popcount bound -> (first generation: store est) -> compact
//                   (seed index, target) pairs
//   k_align<G>      the B4 kernel over the compacted pairs (count read on the device)
//   k_scan_collect  diff <= d -> (seed index, id, diff) hit list, swarmed[id] = 1
//
// Why batching is exact: all sub-seeds of one generation are known when the generation
// starts (hits they produce belong to the next generation and queue up behind them,
// src/algo.cc:205-219); whether a pool amplicon is a hit of a sub-seed does not depend on the
// pool, only on the pair; so the sequential result is "the batch result minus targets an
// earlier sub-seed of the batch already took", which the host resolves in queue order.  The
// union of all hits is exactly the set that leaves the pool.
//
// The unswarmed pool is always in ascending amplicon-id order in the reference (the rotations
// of move_target_to_first_unswarmed_position keep the relative order, src/algo.cc:222-245), so
// "pool order" = id order and the host only needs the hits sorted by (seed index, id).
#include "swa_internal.h"

#include <algorithm>
#include <vector>

int swa_align_launch(swa_ctx * ctx, uint32_t query, const uint32_t * d_queries, const uint32_t * d_targets,
                     const uint32_t * d_count, uint32_t max_count, uint32_t * d_diffs, uint32_t * d_scores,
                     uint32_t * d_alnlens);

extern "C" int swa_scan_fetch(swa_ctx * ctx, uint32_t * hit_seedidx, uint32_t * hit_ids, uint32_t * hit_diffs,
                              uint32_t cap);

namespace {


struct ScanArgs {
  const ulonglong2 * sigs;
  const uint64_t * abundance;
  uint32_t * est;
  uint8_t * swarmed;
  uint32_t n, lo, nseeds;
  const uint32_t * seeds;       // [nseeds] amplicon ids
  const uint32_t * limits;      // [nseeds] radius + d
  uint32_t first_generation, d, ncb;
  uint32_t * t_query;           // compacted pairs: query amplicon id
  uint32_t * t_target;          //                  target amplicon id
  uint32_t * t_seedidx;         //                  index of the seed inside the batch
  uint32_t cap;                 // capacity of the pair arrays
  uint32_t * counters;          // [0] pairs [1] hits [2] pair overflow flag
  unsigned long long * totals;  // [0] (unused) [1] aligned pairs
  unsigned long long * compare_slots;   // q-gram comparisons, one partial sum per workgroup slot
  // candidate list of the current swarm: the pool amplicons whose estimate against the initial
  // seed is <= cand_bound.  Later generations can only ever look at those (algo.cc:521-522 prunes
  // on that same estimate), so they scan the list instead of the whole pool.
  uint32_t * cand;              // amplicon ids, unordered
  uint32_t * cand_count;
  uint32_t cand_bound;
};

// The comparison count is a statistic, and it must not cost anything: thousands of atomics on
// ONE address per launch serialise in L2 (~100 per us) and were 90 % of the pool scan's time.
// Every workgroup adds its sum to its own slot of a 4096-entry table (summed by swa_scan_totals).
constexpr uint32_t kCompareSlots = 4096;
__device__ __forceinline__ void add_comparisons(const ScanArgs & a, unsigned long long compared) {
  __shared__ unsigned long long block_sum;
  if (threadIdx.x == 0) { block_sum = 0ull; }
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) { compared += __shfl_xor(compared, o, 64); }
  if ((threadIdx.x & 63u) == 0u && compared != 0ull) { atomicAdd(&block_sum, compared); }
  __syncthreads();
  if (threadIdx.x == 0 && block_sum != 0ull) {
    atomicAdd(&a.compare_slots[(blockIdx.y * gridDim.x + blockIdx.x) & (kCompareSlots - 1u)], block_sum);
  }
}

__global__ __launch_bounds__(256) void k_scan_filter(const ScanArgs a) {
  const uint32_t sidx = blockIdx.y;
  const uint32_t seed = a.seeds[sidx];
  const uint32_t limit = a.limits[sidx];
  const uint32_t sub = threadIdx.x & 7u;
  const ulonglong2 mine = a.sigs[(uint64_t)seed * 8u + sub];
  const uint64_t seed_ab = a.abundance[seed];
  const uint32_t groups = gridDim.x * (blockDim.x >> 3);
  unsigned long long compared = 0;
  // kUnroll amplicons per turn with every load issued before the first test: the pool scan is a
  // stream of 128-byte signatures and must not serialise on "swarmed? -> abundance? -> signature"
  constexpr uint32_t kUnroll = 4;
  for (uint32_t i0 = a.lo + blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3); i0 < a.n; i0 += kUnroll * groups) {
    uint32_t idx[kUnroll];
    bool in[kUnroll];
    uint8_t sw[kUnroll];
    uint32_t es[kUnroll];
    uint64_t ab[kUnroll];
    ulonglong2 other[kUnroll];
#pragma unroll
    for (uint32_t u = 0; u < kUnroll; ++u) {
      idx[u] = i0 + u * groups;
      in[u] = idx[u] < a.n;
      const uint32_t j = in[u] ? idx[u] : seed;              // (an index that is always valid)
      sw[u] = a.swarmed[j];
      es[u] = a.first_generation == 0u ? a.est[j] : 0u;
      ab[u] = a.abundance[j];
      other[u] = a.sigs[(uint64_t)j * 8u + sub];
    }
#pragma unroll
    for (uint32_t u = 0; u < kUnroll; ++u) {
      const uint32_t i = idx[u];
      bool go = in[u] && i != seed && sw[u] == 0;
      go = go && !(a.first_generation == 0u && es[u] > limit);                  // algo.cc:521-522
      go = go && !(a.ncb == 0u && ab[u] > seed_ab);                             // algo.cc:427-428, 523-525
      uint32_t pop = (uint32_t)__popcll(mine.x ^ other[u].x) + (uint32_t)__popcll(mine.y ^ other[u].y);
      pop += __shfl_xor(pop, 1, 8);
      pop += __shfl_xor(pop, 2, 8);
      pop += __shfl_xor(pop, 4, 8);
      if (go && sub == 0u) {
        const uint32_t qd = (pop + 9u) / 10u;
        ++compared;
        if (a.first_generation != 0u) {
          a.est[i] = qd;                                                         // algo.cc:442
          if (qd <= a.cand_bound) { a.cand[atomicAdd(a.cand_count, 1u)] = i; }
        }
        if (qd <= a.d) {
          const uint32_t at = atomicAdd(&a.counters[0], 1u);
          if (at < a.cap) { a.t_query[at] = seed; a.t_target[at] = i; a.t_seedidx[at] = sidx; }
          else { a.counters[2] = 1u; }
        }
      }
    }
  }
  add_comparisons(a, compared);
}

// `mirror` is pinned host memory the GPU writes directly (zero copy): [0..3] = pairs, hits,
// overflow flag, 0, then the first mirror_cap hit triples — what the host needs after every
// batch without a single copy command.  The complete hit list stays in `hits` (device) for the
// rare batch with more hits than the mirror holds.  One workgroup: the header is written after
// the last hit.
__global__ __launch_bounds__(1024) void k_scan_collect(const uint32_t * __restrict__ t_target,
                                                       const uint32_t * __restrict__ t_seedidx,
                                                       const uint32_t * __restrict__ diffs, uint32_t * counters,
                                                       uint32_t cap, uint32_t d, uint8_t * swarmed,
                                                       uint32_t * __restrict__ hits, unsigned long long * totals,
                                                       uint32_t * mirror, uint32_t mirror_cap) {
  if (counters[2] == 0u) {                      // (pair overflow: the host grows the buffers and redoes the batch)
    const uint32_t nt = counters[0] < cap ? counters[0] : cap;
    for (uint32_t t = threadIdx.x; t < nt; t += blockDim.x) {
      if (diffs[t] <= d) {
        const uint32_t at = atomicAdd(&counters[1], 1u);
        const uint32_t sidx = t_seedidx[t], id = t_target[t], df = diffs[t];
        hits[3u * at] = sidx; hits[3u * at + 1u] = id; hits[3u * at + 2u] = df;
        if (at < mirror_cap) { mirror[4u + 3u * at] = sidx; mirror[5u + 3u * at] = id; mirror[6u + 3u * at] = df; }
        swarmed[id] = 1;
      }
    }
    if (threadIdx.x == 0) { atomicAdd(&totals[1], (unsigned long long)nt); }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    mirror[0] = counters[0]; mirror[1] = counters[1]; mirror[2] = counters[2]; mirror[3] = 0u;
    __threadfence_system();
  }
}

// later generations: the same tests as k_scan_filter over the swarm's candidate list
__global__ __launch_bounds__(256) void k_scan_filter_list(const ScanArgs a) {
  const uint32_t sidx = blockIdx.y;
  const uint32_t seed = a.seeds[sidx];
  const uint32_t limit = a.limits[sidx];
  const uint32_t sub = threadIdx.x & 7u;
  const ulonglong2 mine = a.sigs[(uint64_t)seed * 8u + sub];
  const uint64_t seed_ab = a.abundance[seed];
  const uint32_t groups = gridDim.x * (blockDim.x >> 3);
  const uint32_t ncand = *a.cand_count;
  unsigned long long compared = 0;
  for (uint32_t c = blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3); c < ncand; c += groups) {
    const uint32_t i = a.cand[c];
    if (i < a.lo || i == seed || a.swarmed[i] != 0) { continue; }
    if (a.est[i] > limit) { continue; }                                          // algo.cc:521-522
    if (a.ncb == 0u && a.abundance[i] > seed_ab) { continue; }                  // algo.cc:523-525
    const ulonglong2 other = a.sigs[(uint64_t)i * 8u + sub];
    uint32_t pop = (uint32_t)__popcll(mine.x ^ other.x) + (uint32_t)__popcll(mine.y ^ other.y);
    pop += __shfl_xor(pop, 1, 8);
    pop += __shfl_xor(pop, 2, 8);
    pop += __shfl_xor(pop, 4, 8);
    if (sub == 0u) {
      const uint32_t qd = (pop + 9u) / 10u;
      ++compared;
      if (qd <= a.d) {
        const uint32_t at = atomicAdd(&a.counters[0], 1u);
        if (at < a.cap) { a.t_query[at] = seed; a.t_target[at] = i; a.t_seedidx[at] = sidx; }
        else { a.counters[2] = 1u; }
      }
    }
  }
  add_comparisons(a, compared);
}

// a sub-seed's limit outgrew the list's bound: collect the list again from the stored estimates
__global__ __launch_bounds__(256) void k_scan_relist(const uint32_t * __restrict__ est, const uint8_t * __restrict__ swarmed,
                                                     uint32_t lo, uint32_t n, uint32_t bound, uint32_t * cand,
                                                     uint32_t * cand_count) {
  for (uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (swarmed[i] == 0 && est[i] <= bound) { cand[atomicAdd(cand_count, 1u)] = i; }
  }
}

// seeds + limits: pinned host memory -> HBM, by one workgroup.  (Letting every wave of the pool
// scan read them across PCIe cost ~80 us per launch.)  In a first generation the initial seed
// joins its own swarm here.
// Also resets the batch counters (and, for a first generation, the candidate list): the first
// kernel of every launch sequence, so that the sequence needs no memset commands.
__global__ void k_scan_stage(const uint32_t * __restrict__ pinned, uint32_t * __restrict__ staged, uint32_t words,
                             uint8_t * swarmed, uint32_t mark_first, uint32_t * counters, uint32_t * cand_count,
                             uint32_t reset_cand) {
  for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) { staged[i] = pinned[i]; }
  if (threadIdx.x < 4u) { counters[threadIdx.x] = 0u; }
  if (threadIdx.x == 0) {
    if (mark_first != 0u) { swarmed[pinned[0]] = 1; }
    if (reset_cand != 0u) { *cand_count = 0u; }
  }
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
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_counters, 64));
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_compares, kCompareSlots * sizeof(unsigned long long)));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_scan_compares.ptr, 0, kCompareSlots * sizeof(unsigned long long), ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_scan_est.ptr, 0, n * sizeof(uint32_t), ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_scan_swarmed.ptr, 0, n, ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_scan_counters.ptr, 0, 64, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scan_ready = true;
  return SWA_OK;
}

// One batch: every seeds[k] (with radius radii[k]) against the unswarmed pool.
// hit_* receive triples sorted by (seed index, target id); *nhits = number of triples.
extern "C" int swa_scan_batch(swa_ctx * ctx, uint32_t nseeds, const uint32_t * seeds, const uint32_t * radii,
                              uint32_t lowest_unswarmed, int first_generation, int no_cluster_breaking,
                              uint32_t * hit_seedidx, uint32_t * hit_ids, uint32_t * hit_diffs, uint32_t cap,
                              uint32_t * nhits) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (!ctx->scan_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_scan_batch: call swa_scan_begin first"); }
  if (nhits == nullptr || seeds == nullptr || radii == nullptr || nseeds == 0 || nseeds > 65535 ||
      (first_generation != 0 && nseeds != 1) ||
      (cap != 0 && (hit_ids == nullptr || hit_diffs == nullptr || hit_seedidx == nullptr))) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_scan_batch: bad argument");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->db.n;
  const uint32_t d = (uint32_t)ctx->resolution;
  for (uint32_t k = 0; k < nseeds; ++k) {
    if (seeds[k] >= n) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_scan_batch: seed out of range"); }
  }
  auto * counters = static_cast<uint32_t *>(ctx->d_scan_counters.ptr);          // u32[4] then u64 totals[4]
  auto * totals = reinterpret_cast<unsigned long long *>(counters + 4);

  // seeds + limits travel through pinned host memory the kernels read directly, the hits come
  // back the same way (k_scan_collect): no copy command in the per-generation sequence
  constexpr uint32_t kMirrorHits = 16384;
  const size_t pinned_words = 2 * 65536 + 4 + 3 * (size_t)kMirrorHits;
  if (ctx->h_scan_pinned == nullptr) {
    SWA_HIP(ctx, hipHostMalloc(&ctx->h_scan_pinned, pinned_words * sizeof(uint32_t), hipHostMallocDefault));
  }
  uint32_t * pin_seeds = static_cast<uint32_t *>(ctx->h_scan_pinned);
  uint32_t * mirror = pin_seeds + 2 * 65536;
  for (uint32_t k = 0; k < nseeds; ++k) { pin_seeds[k] = seeds[k]; pin_seeds[nseeds + k] = radii[k] + d; }
  const uint32_t lo = lowest_unswarmed < n ? lowest_unswarmed : n;
  const uint32_t span = n - lo;
  // pair capacity: a first guess that is grown (and the batch redone) if it ever overflows
  uint64_t pair_cap = ctx->scan_pair_cap;
  if (pair_cap == 0) { pair_cap = std::max<uint64_t>(1u << 16, (uint64_t)n); }
  uint32_t got = 0;
  ++ctx->scan_launches;
  for (;;) {
    SWA_TRY(swa_reserve(ctx, ctx->d_scan_targets, 3 * pair_cap * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_scan_diffs, pair_cap * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_scan_hits, (3 * pair_cap + 8) * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_scan_cand, (uint64_t)n * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_scan_seeds, 2 * (size_t)nseeds * sizeof(uint32_t)));
    ctx->scan_pair_cap = pair_cap;
    ScanArgs a{};
    a.sigs = static_cast<const ulonglong2 *>(ctx->d_qgrams.ptr);
    a.abundance = ctx->db.abundance;
    a.est = static_cast<uint32_t *>(ctx->d_scan_est.ptr);
    a.swarmed = static_cast<uint8_t *>(ctx->d_scan_swarmed.ptr);
    a.n = n; a.lo = lo; a.nseeds = nseeds;
    a.seeds = static_cast<const uint32_t *>(ctx->d_scan_seeds.ptr);
    a.limits = a.seeds + nseeds;
    a.first_generation = first_generation != 0 ? 1u : 0u;
    a.d = d;
    a.ncb = no_cluster_breaking != 0 ? 1u : 0u;
    a.t_query = static_cast<uint32_t *>(ctx->d_scan_targets.ptr);
    a.t_target = a.t_query + pair_cap;
    a.t_seedidx = a.t_target + pair_cap;
    a.cap = (uint32_t)std::min<uint64_t>(pair_cap, 0xFFFFFFFFull);
    a.counters = counters;
    a.totals = totals;
    a.compare_slots = static_cast<unsigned long long *>(ctx->d_scan_compares.ptr);
    a.cand = static_cast<uint32_t *>(ctx->d_scan_cand.ptr);
    a.cand_count = counters + 12;                      // (u32[4] counters, u64[4] totals, then the list length)
    auto * hits = static_cast<uint32_t *>(ctx->d_scan_hits.ptr);
    uint32_t max_limit = 0;
    for (uint32_t k = 0; k < nseeds; ++k) { max_limit = std::max(max_limit, radii[k] + d); }
    const bool relist = a.first_generation == 0u && max_limit > ctx->scan_cand_bound;
    hipLaunchKernelGGL(k_scan_stage, dim3(1), dim3(256), 0, ctx->stream, pin_seeds,
                       static_cast<uint32_t *>(ctx->d_scan_seeds.ptr), 2u * nseeds, a.swarmed, a.first_generation, counters,
                       a.cand_count, (a.first_generation != 0u || relist) ? 1u : 0u);
    if (span > 0) {
      uint64_t blocks = ((uint64_t)span + 31) / 32;
      const uint64_t gcap = std::max<uint64_t>(1, uint64_t(ctx->num_cus) * 8 / nseeds);
      if (blocks > gcap) { blocks = gcap; }
      if (a.first_generation != 0u) {
        // whole pool: estimates for everybody, and the candidate list of this swarm
        ctx->scan_cand_bound = 8u * d;
        a.cand_bound = ctx->scan_cand_bound;
        hipLaunchKernelGGL(k_scan_filter, dim3((unsigned)blocks, nseeds), dim3(256), 0, ctx->stream, a);
      } else {
        if (relist) {                                      // (rare) the radius outgrew the list: collect it again, wider
          ctx->scan_cand_bound = std::max(2u * ctx->scan_cand_bound, max_limit);
          hipLaunchKernelGGL(k_scan_relist, dim3((unsigned)std::min<uint64_t>(((uint64_t)span + 255) / 256, uint64_t(ctx->num_cus) * 8)),
                             dim3(256), 0, ctx->stream, a.est, a.swarmed, lo, n, ctx->scan_cand_bound, a.cand, a.cand_count);
        }
        a.cand_bound = ctx->scan_cand_bound;
        const uint64_t lblocks = std::max<uint64_t>(1, std::min<uint64_t>(gcap, 64));
        hipLaunchKernelGGL(k_scan_filter_list, dim3((unsigned)lblocks, nseeds), dim3(256), 0, ctx->stream, a);
      }
      SWA_HIP(ctx, hipGetLastError());
      const uint64_t max_pairs = std::min<uint64_t>((uint64_t)span * nseeds, a.cap);
      SWA_TRY(swa_align_launch(ctx, 0, a.t_query, a.t_target, counters, (uint32_t)max_pairs,
                               static_cast<uint32_t *>(ctx->d_scan_diffs.ptr), nullptr, nullptr));
      hipLaunchKernelGGL(k_scan_collect, dim3(1), dim3(1024), 0, ctx->stream, a.t_target, a.t_seedidx,
                         static_cast<const uint32_t *>(ctx->d_scan_diffs.ptr), counters, a.cap, d, a.swarmed, hits + 4,
                         totals, mirror, kMirrorHits);
      SWA_HIP(ctx, hipGetLastError());
    } else {
      mirror[0] = mirror[1] = mirror[2] = mirror[3] = 0u;   // nothing launched: no pairs, no hits
    }
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->scan_host.assign(mirror, mirror + 4);
    if (ctx->scan_host[2] != 0u) {
      // more (seed, target) pairs than the buffers hold.  Nothing irreversible happened (est
      // stores are idempotent, k_scan_collect marks nothing when the flag is set), so grow and
      // run the batch again.
      pair_cap = std::max<uint64_t>(2 * pair_cap, (uint64_t)ctx->scan_host[0] + 1024);
      continue;
    }
    got = ctx->scan_host[1];
    ctx->scan_host.resize(4 + 3 * (size_t)got);
    if (got <= kMirrorHits) {
      std::copy(mirror + 4, mirror + 4 + 3 * (size_t)got, ctx->scan_host.begin() + 4);
    } else {                                                 // more hits than the mirror holds: fetch them all
      SWA_HIP(ctx, hipMemcpyAsync(ctx->scan_host.data() + 4, hits + 4, 3 * (size_t)got * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, ctx->stream));
      SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    break;
  }
  *nhits = got;
  // queue order of the seeds, then pool (= id) order of the targets; kept in the context so
  // that a caller whose buffers were too small can fetch the result after growing them
  struct Hit { uint32_t sidx, id, diff; };
  const auto * triples = reinterpret_cast<const Hit *>(ctx->scan_host.data() + 4);
  ctx->scan_sorted.resize(got);
  for (uint32_t k = 0; k < got; ++k) { ctx->scan_sorted[k] = ((uint64_t)triples[k].sidx << 32) | triples[k].id; }
  std::vector<uint32_t> & perm = ctx->scan_perm;
  perm.resize(got);
  for (uint32_t k = 0; k < got; ++k) { perm[k] = k; }
  std::sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return ctx->scan_sorted[x] < ctx->scan_sorted[y]; });
  if (got > cap) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_scan_batch: hit buffer too small, use swa_scan_fetch"); }
  return swa_scan_fetch(ctx, hit_seedidx, hit_ids, hit_diffs, cap);
}

// copies the (sorted) hits of the most recent swa_scan_batch; cap must be >= its *nhits
extern "C" int swa_scan_fetch(swa_ctx * ctx, uint32_t * hit_seedidx, uint32_t * hit_ids, uint32_t * hit_diffs,
                              uint32_t cap) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  const uint32_t got = (uint32_t)ctx->scan_perm.size();
  if (got > cap) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_scan_fetch: hit buffer too small"); }
  struct Hit { uint32_t sidx, id, diff; };
  const auto * triples = reinterpret_cast<const Hit *>(ctx->scan_host.data() + 4);
  for (uint32_t k = 0; k < got; ++k) {
    const Hit & h = triples[ctx->scan_perm[k]];
    if (hit_seedidx != nullptr) { hit_seedidx[k] = h.sidx; }
    hit_ids[k] = h.id;
    hit_diffs[k] = h.diff;
  }
  return SWA_OK;
}

// single (sub)seed convenience form
extern "C" int swa_scan_step(swa_ctx * ctx, uint32_t seed, uint32_t lowest_unswarmed, int first_generation,
                             uint32_t radius, int no_cluster_breaking, uint32_t * hit_ids, uint32_t * hit_diffs,
                             uint32_t cap, uint32_t * nhits) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  ctx->scan_idx_tmp.resize(cap > 0 ? cap : 1);
  uint32_t * idx = ctx->scan_idx_tmp.data();   // (scan_batch does not touch scan_idx_tmp)
  return swa_scan_batch(ctx, 1, &seed, &radius, lowest_unswarmed, first_generation, no_cluster_breaking, idx, hit_ids,
                        hit_diffs, cap, nhits);
}

extern "C" int swa_scan_totals(swa_ctx * ctx, uint64_t * out3) {
  if (ctx == nullptr || out3 == nullptr) { return SWA_E_ARG; }
  if (!ctx->scan_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_scan_totals: no scan state"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t t[4] = {};
  std::vector<unsigned long long> slots(kCompareSlots);
  SWA_HIP(ctx, hipMemcpyAsync(t, static_cast<uint32_t *>(ctx->d_scan_counters.ptr) + 4, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipMemcpyAsync(slots.data(), ctx->d_scan_compares.ptr, kCompareSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t compared = 0;
  for (unsigned long long v : slots) { compared += v; }
  out3[0] = compared; out3[1] = t[1]; out3[2] = ctx->scan_launches;
  return SWA_OK;
}

// loads this translation unit's code object (see swa_ctx_warmup): an empty launch
__global__ void k_warm_scan() {}
void swa_warm_scan(swa_ctx * ctx) { hipLaunchKernelGGL(k_warm_scan, dim3(1), dim3(64), 0, ctx->stream); }
