// cluster_gpu.hip — the d = 1 agglomeration over the network that is still in HBM.
//
// The reference walks the network serially (src/algod1.cc:1175-1257: seeds in db order, a breadth-first
// queue per swarm, every generation sorted by id).  cluster_d1.cpp showed that the result is a pure function
// of the directed graph — swarm(v) = the smallest id that reaches v, generation(v) = its distance from that
// seed inside the swarm, parent(v) = the smallest id of the previous generation that points at v, members in
// (generation, id) order — and computes it with OpenMP sweeps.  Those sweeps are random accesses over 10^7
// nodes and 2 x 10^7 links: 0.36 s on a 256-core host, a few milliseconds here, and the 150 MB CSR no longer
// has to cross PCIe at all (only the four result arrays do).
//
//   k_label_sweep     label[v] = min(label[v], label[u]) along every link u -> v — and label[v] = label[label[v]]: pointer
//                     jumping — until nothing changes; only vertices whose label changed in the sweep before hand theirs on;
//                     sweeps launched ahead in batches, one behind the fixed point returns at once (no host round trip a sweep)
//   k_level_sweep     level-synchronous distances from the seeds + parents (atomicMin of the claiming ids), launched ahead alike
//   rocPRIM           seeds -> swarm numbers (exclusive scan); members ordered by one stable radix sort of
//                     (swarm << gbits | generation) over the bits that vary (23 of them at 10 M amplicons, 32-bit keys), the ids
//                     ascending as payload; k_swarm_bounds reads the swarms' first members off the sorted keys
// Round 6: 17.4 -> ~5 ms at 10 M for the device part + download (profiles/r06/NOTES.md).
#include "swa_internal.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>

namespace {

constexpr uint32_t kUnset = SWA_NO_AMPLICON;
constexpr uint32_t kSweepBatch = 10;     // label sweeps launched per host round trip (a sweep behind the fixed point returns at once)
constexpr uint32_t kSweepFlags = 1024;   // flags[s] = sweep s changed a label
constexpr uint32_t kLevelBatch = 32;     // generations launched per host round trip (a sweep behind an empty generation returns at once)

__global__ __launch_bounds__(256) void k_label_init(uint32_t * label, uint32_t * parent, uint32_t * stamp, uint32_t * gen, uint32_t n) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) { label[v] = v; parent[v] = kUnset; stamp[v] = 0u; gen[v] = kUnset; }
}

// label[v] = the smallest id known to reach v.  One sweep: every vertex takes its label's label (pointer jumping: whatever
// reaches l reaches everything l reaches — the sweeps needed fall from the depth of the deepest swarm towards its logarithm),
// and every ACTIVE vertex — its label changed in the sweep before, or just now — hands its label down its links.  stamp[v] =
// the last sweep that changed label[v]; a change made to v after v's own thread went by is handed on by the next sweep.
// Sweeps are launched ahead in batches: one behind a sweep that changed nothing (the fixed point) returns at once.
__global__ __launch_bounds__(256) void k_label_sweep(const uint64_t * __restrict__ offsets, const uint32_t * __restrict__ nb, uint32_t n,
                                                     uint32_t * label, uint32_t * stamp, uint32_t s, uint32_t * flags, uint32_t jump_all) {
  if (s > 1u && __hip_atomic_load(&flags[s - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) { return; }
  bool any = false;
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
    uint32_t lu = label[u];
    bool active = stamp[u] + 1u >= s;
    if (jump_all != 0u || active) {
      const uint32_t ll = label[lu];
      if (ll < lu) { atomicMin(&label[u], ll); lu = ll; stamp[u] = s; any = true; active = true; }
    }
    if (!active) { continue; }
    for (uint64_t e = offsets[u]; e < offsets[u + 1]; ++e) {
      const uint32_t v = nb[e];
      if (lu < label[v]) { atomicMin(&label[v], lu); stamp[v] = s; any = true; }
    }
  }
  if (any) { flags[s] = 1u; }
}

// ---- generations and parents: level-synchronous from all seeds at once -------------------------------------------------------
// A sweep per generation: the vertices of generation level - 1 claim their unclaimed neighbours of the same swarm.  The sweeps
// are launched ahead, a batch per host round trip; any[r] = the batch's r-th generation took somebody in — a sweep behind an
// empty generation returns at once.  (Frontier queues with workgroup-staged appends were built and measured this round: 2.3 ms
// for what these sweeps do in 1.1 at 10 M amplicons — the queue's random reads of label, offsets and the claimed vertices'
// own offsets cost more than streaming over gen[] does; lease r6d.)
__global__ __launch_bounds__(256) void k_level_init(const uint32_t * __restrict__ label, uint32_t * gen, uint8_t * is_seed, uint32_t n) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const bool seed = label[v] == v;
    gen[v] = seed ? 0u : kUnset;
    is_seed[v] = seed ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void k_level_sweep(const uint64_t * __restrict__ offsets, const uint32_t * __restrict__ nb, uint32_t n,
                                                     const uint32_t * __restrict__ label, uint32_t * gen, uint32_t * parent, uint32_t level,
                                                     uint32_t * any, uint32_t r) {
  if (r > 0u && __hip_atomic_load(&any[r - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) { return; }
  bool claimed = false;
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
    if (__hip_atomic_load(&gen[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != level - 1u) { continue; }
    const uint32_t lu = label[u];
    for (uint64_t e = offsets[u]; e < offsets[u + 1]; ++e) {
      const uint32_t v = nb[e];
      if (label[v] != lu) { continue; }
      const uint32_t gv = __hip_atomic_load(&gen[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gv == kUnset || gv == level) {
        if (gv == kUnset) { __hip_atomic_store(&gen[v], level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        atomicMin(&parent[v], u);
        claimed = true;
      }
    }
  }
  if (claimed) { any[r] = 1u; }
}

// sort keys: (swarm number << gbits | generation), ids ascending as payload — only the bits that vary are sorted
template <class K>
__global__ __launch_bounds__(256) void k_swarm_keys(const uint32_t * __restrict__ label, const uint32_t * __restrict__ gen,
                                                    const uint32_t * __restrict__ seed_rank, uint32_t n, uint32_t gbits, uint32_t * __restrict__ swarmid,
                                                    K * __restrict__ keys, uint32_t * __restrict__ ids) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const uint32_t sid = seed_rank[label[v]];
    swarmid[v] = sid;
    keys[v] = ((K)sid << gbits) | (K)gen[v];
    ids[v] = v;
  }
}

// begins[s] = where swarm s starts in the sorted order (every swarm holds its seed: none is empty); begins[nswarms] = n
template <class K>
__global__ __launch_bounds__(256) void k_swarm_bounds(const K * __restrict__ sorted, uint32_t n, uint32_t gbits, uint32_t nswarms, uint32_t * __restrict__ begins) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t s = (uint32_t)(sorted[i] >> gbits);
    if (i == 0u || (uint32_t)(sorted[i - 1u] >> gbits) != s) { begins[s] = i; }
    if (i == n - 1u) { begins[nswarms] = n; }
  }
}

struct widen_u8 { __host__ __device__ uint32_t operator()(uint8_t v) const { return (uint32_t)v; } };

int blocks_for(const swa_ctx * ctx, uint64_t items) {
  const uint64_t b = (items + 255) / 256, cap = (uint64_t)ctx->num_cus * 8;
  return (int)std::max<uint64_t>(1, std::min(b, cap));
}

uint32_t bit_width_u32(uint32_t v) { uint32_t b = 0; while (v != 0u) { ++b; v >>= 1; } return b; }

}  // namespace

// loads this translation unit's code object (see swa_ctx_warmup): an empty launch
void swa_warm_cluster(swa_ctx * ctx) {
  hipLaunchKernelGGL(k_label_init, dim3(1), dim3(64), 0, ctx->stream, static_cast<uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr),
                     static_cast<uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr), 0u);
}

// The network of the whole database computed into the context's own buffers (no host copy): what
// swa_d1_network does up to the download.  *total = number of links.
extern "C" int swa_d1_network_resident(swa_ctx * ctx, int no_cluster_breaking, uint64_t * total) {
  if (ctx == nullptr || total == nullptr) { return SWA_E_ARG; }
  if (!ctx->d1_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_network_resident: call swa_d1_index_build first"); }
  const uint32_t n = ctx->db.n;
  ctx->csr_ready = false;
  ctx->cluster_ready = false;
  SWA_TRY(swa_reserve(ctx, ctx->d_offsets_tmp, ((uint64_t)n + 1) * sizeof(uint64_t)));
  uint64_t cap = std::max<uint64_t>(ctx->d_nb_tmp.bytes / sizeof(uint32_t), 4ull * n + 1024);
  for (;;) {
    SWA_TRY(swa_reserve(ctx, ctx->d_nb_tmp, cap * sizeof(uint32_t)));
    const int rc = swa_d1_network_device(ctx, no_cluster_breaking, 0, n, static_cast<uint64_t *>(ctx->d_offsets_tmp.ptr),
                                         static_cast<uint32_t *>(ctx->d_nb_tmp.ptr), cap, total);
    if (rc == SWA_E_CAPACITY) { cap = *total + 1024; continue; }
    if (rc != SWA_OK) { return rc; }
    break;
  }
  ctx->csr_ready = true;
  ctx->csr_has_diffs = false;
  ctx->csr_total = *total;
  return SWA_OK;
}

// the resident CSR -> host (for -j and for callers that want the lists themselves)
extern "C" int swa_d1_network_fetch(swa_ctx * ctx, uint64_t * offsets, uint32_t * neighbours, uint64_t cap) {
  if (ctx == nullptr || offsets == nullptr) { return SWA_E_ARG; }
  if (!ctx->csr_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_network_fetch: no resident network"); }
  if (ctx->csr_total > cap) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_d1_network_fetch: neighbour buffer too small"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  swa_touch_pages(offsets, ((uint64_t)ctx->db.n + 1) * sizeof(uint64_t));
  swa_touch_pages(neighbours, ctx->csr_total * sizeof(uint32_t));
  SWA_HIP(ctx, hipMemcpyAsync(offsets, ctx->d_offsets_tmp.ptr, ((uint64_t)ctx->db.n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  if (ctx->csr_total != 0) {
    SWA_HIP(ctx, hipMemcpyAsync(neighbours, ctx->d_nb_tmp.ptr, ctx->csr_total * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}

// Clustering of the resident network.  All arrays have n entries (order: the members of swarm s are
// order[swarm_begin[s], swarm_begin[s + 1]), seed first, then by generation and id); swarm_begin needs
// swarm_cap + 1 entries: SWA_E_CAPACITY with *nswarms = the number needed when it is too small.
extern "C" int swa_d1_cluster_device(swa_ctx * ctx, uint32_t * swarmid, uint32_t * generation, uint32_t * parent, uint32_t * order,
                                     uint32_t * swarm_begin, uint32_t swarm_cap, uint32_t * nswarms) {
  // (swarmid / generation / parent may be null: they stay in HBM for swa_d1_cluster_fetch — a run that only writes the
  // swarms, the usual one, never looks at them)
  if (ctx == nullptr || order == nullptr || nswarms == nullptr) { return SWA_E_ARG; }
  ctx->cluster_ready = false;
  if (!ctx->csr_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_cluster_device: no resident network (swa_d1_network_resident)"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->db.n;
  const auto * offsets = static_cast<const uint64_t *>(ctx->d_offsets_tmp.ptr);
  const auto * nb = static_cast<const uint32_t *>(ctx->d_nb_tmp.ptr);
  // label | gen | parent | swarmid | ids_in | ids_out | seed_rank | begins (n + 2) | stamp, then keys in / out (u64 room), seed flags
  SWA_TRY(swa_reserve(ctx, ctx->d_cluster, (uint64_t)n * 4 * 10 + 64 + (uint64_t)n * 8 * 2 + n + 64));
  auto * label = static_cast<uint32_t *>(ctx->d_cluster.ptr);
  auto * gen = label + n, * par = gen + n, * sid = par + n, * ids_in = sid + n, * ids_out = ids_in + n, * seed_rank = ids_out + n;
  auto * begins = seed_rank + n, * stamp = begins + n + 4;
  auto * keys_in = reinterpret_cast<unsigned long long *>(stamp + n + (n & 1u));      // (8-byte aligned: 9 n + 4 words before it, one more when n is odd)
  auto * keys_out = keys_in + n;
  auto * is_seed = reinterpret_cast<uint8_t *>(keys_out + n);
  // control words: [0, kSweepFlags) sweep flags, then kLevelBatch "this generation took somebody in" flags
  SWA_TRY(swa_reserve(ctx, ctx->d_cluster_ctl, (kSweepFlags + kLevelBatch) * sizeof(uint32_t)));
  auto * flags = static_cast<uint32_t *>(ctx->d_cluster_ctl.ptr);
  auto * cnt = flags + kSweepFlags;
  uint32_t host_ctl[kLevelBatch];
  const dim3 g(blocks_for(ctx, n)), b(256);
  // SWARM_AMD_CLUSTER_TIMING: wall clock between the host's synchronisation points
  static const bool timing = getenv("SWARM_AMD_CLUSTER_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = now();
  auto lap = [&](const char * what, uint32_t count) {
    if (!timing) { return; }
    const double t = now();
    std::fprintf(stderr, "[cluster gpu] %-28s %8.3f ms  (%u)\n", what, t - t_last, count);
    t_last = t;
  };
  static const bool jump_all = [] { const char * e = getenv("SWA_CLUSTER_JUMP"); return e == nullptr || e[0] != 'a'; }();   // ("active": experiment)
  lap("buffers", 0);
  hipLaunchKernelGGL(k_label_init, g, b, 0, ctx->stream, label, par, stamp, gen, n);
  // ---- swarms: the smallest id that reaches every vertex
  for (uint32_t s = 1;;) {
    if (s == 1u) { SWA_HIP(ctx, hipMemsetAsync(flags, 0, kSweepFlags * sizeof(uint32_t), ctx->stream)); }
    const uint32_t last = std::min(s + kSweepBatch, kSweepFlags) - 1u;
    for (; s <= last; ++s) { hipLaunchKernelGGL(k_label_sweep, g, b, 0, ctx->stream, offsets, nb, n, label, stamp, s, flags, jump_all ? 1u : 0u); }
    uint32_t changed = 0;
    SWA_HIP(ctx, hipMemcpyAsync(&changed, flags + last, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lap("label sweeps, batch", s - 1u);
    if (changed == 0) { break; }                       // (the batch's last sweep changed nothing — or was behind one that did not)
    if (s == kSweepFlags) {                            // (a graph that needs a thousand sweeps: start the numbering again, everybody active)
      SWA_HIP(ctx, hipMemsetAsync(stamp, 0, (uint64_t)n * sizeof(uint32_t), ctx->stream));
      s = 1;
    }
  }
  // ---- swarm numbers = rank of the seed among the seeds (queued behind the first frontiers; read with them)
  size_t tmp_bytes = 0, need = 0;
  auto seeds32 = rocprim::make_transform_iterator(is_seed, widen_u8());
  (void)rocprim::exclusive_scan(nullptr, need, seeds32, seed_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream);
  tmp_bytes = std::max(tmp_bytes, need);
  (void)rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, ids_in, ids_out, (size_t)n, 0, 64, ctx->stream);
  tmp_bytes = std::max(tmp_bytes, need);
  (void)rocprim::radix_sort_pairs(nullptr, need, reinterpret_cast<uint32_t *>(keys_in), reinterpret_cast<uint32_t *>(keys_out), ids_in, ids_out, (size_t)n, 0, 32, ctx->stream);
  tmp_bytes = std::max(tmp_bytes, need);
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_hits, tmp_bytes + 16));
  // ---- generations and parents, and — queued with them — the swarm numbers = rank of the seed among the seeds
  auto * any = cnt;
  hipLaunchKernelGGL(k_level_init, g, b, 0, ctx->stream, label, gen, is_seed, n);
  need = tmp_bytes;
  SWA_HIP(ctx, rocprim::exclusive_scan(ctx->d_scan_hits.ptr, need, seeds32, seed_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
  uint32_t last_rank = 0, maxgen = 0;
  uint8_t last_seed = 0;
  SWA_HIP(ctx, hipMemcpyAsync(&last_rank, seed_rank + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipMemcpyAsync(&last_seed, is_seed + (n - 1), 1, hipMemcpyDeviceToHost, ctx->stream));
  for (uint32_t level0 = 1;; level0 += kLevelBatch) {   // (level0 = the generation the batch's first sweep hands out)
    SWA_HIP(ctx, hipMemsetAsync(any, 0, kLevelBatch * sizeof(uint32_t), ctx->stream));
    for (uint32_t r = 0; r < kLevelBatch; ++r) {
      hipLaunchKernelGGL(k_level_sweep, g, b, 0, ctx->stream, offsets, nb, n, label, gen, par, level0 + r, any, r);
    }
    SWA_HIP(ctx, hipMemcpyAsync(host_ctl, any, kLevelBatch * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lap("seeds, ranks, generations", level0 + kLevelBatch - 1u);
    for (uint32_t r = 0; r < kLevelBatch; ++r) { if (host_ctl[r] != 0u) { maxgen = level0 + r; } }
    if (host_ctl[kLevelBatch - 1u] == 0u) { break; }     // (the batch's last generation is empty: nobody is left to claim anybody)
  }
  ctx->cluster_maxgen = maxgen;
  *nswarms = last_rank + last_seed;
  if (*nswarms > swarm_cap || swarm_begin == nullptr) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_d1_cluster_device: swarm table too small"); }
  // ---- members by (swarm, generation, id): one stable radix sort over the bits that vary, ids ascending going in
  const uint32_t gbits = bit_width_u32(maxgen), sbits = bit_width_u32(*nswarms - 1u);
  const bool narrow = gbits + sbits <= 32u;
  need = tmp_bytes;
  if (narrow) {
    auto * k_in = reinterpret_cast<uint32_t *>(keys_in), * k_out = reinterpret_cast<uint32_t *>(keys_out);
    hipLaunchKernelGGL(k_swarm_keys<uint32_t>, g, b, 0, ctx->stream, label, gen, seed_rank, n, gbits, sid, k_in, ids_in);
    SWA_HIP(ctx, rocprim::radix_sort_pairs(ctx->d_scan_hits.ptr, need, k_in, k_out, ids_in, ids_out, (size_t)n, 0, std::max(1u, gbits + sbits), ctx->stream));
    hipLaunchKernelGGL(k_swarm_bounds<uint32_t>, g, b, 0, ctx->stream, k_out, n, gbits, *nswarms, begins);
  } else {
    hipLaunchKernelGGL(k_swarm_keys<unsigned long long>, g, b, 0, ctx->stream, label, gen, seed_rank, n, gbits, sid, keys_in, ids_in);
    SWA_HIP(ctx, rocprim::radix_sort_pairs(ctx->d_scan_hits.ptr, need, keys_in, keys_out, ids_in, ids_out, (size_t)n, 0, gbits + sbits, ctx->stream));
    hipLaunchKernelGGL(k_swarm_bounds<unsigned long long>, g, b, 0, ctx->stream, keys_out, n, gbits, *nswarms, begins);
  }
  if (swarmid != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(swarmid, sid, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  if (generation != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(generation, gen, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  if (parent != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(parent, par, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  SWA_HIP(ctx, hipGetLastError());
  if (timing) { SWA_HIP(ctx, hipStreamSynchronize(ctx->stream)); lap("keys, sort, bounds", gbits + sbits); }
  for (uint32_t * out : {swarmid, generation, parent, order}) { swa_touch_pages(out, (uint64_t)n * sizeof(uint32_t)); }   // (pinned pages: nothing to do)
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  if (timing) { for (auto & e : ev) { (void)hipEventCreate(&e); } (void)hipEventRecord(ev[0], ctx->stream); }
  SWA_HIP(ctx, hipMemcpyAsync(order, ids_out, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  lap("  order copy queued", 0);
  if (timing) { (void)hipEventRecord(ev[1], ctx->stream); }
  SWA_HIP(ctx, hipMemcpyAsync(swarm_begin, begins, ((uint64_t)*nswarms + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  lap("  bounds copy queued", 0);
  if (timing) { (void)hipEventRecord(ev[2], ctx->stream); }
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  lap("download order + bounds", *nswarms);
  if (timing) {
    float a = 0, b = 0;
    (void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]);
    std::fprintf(stderr, "[cluster gpu] by the stream's own events: order copy %.3f ms, bounds copy %.3f ms\n", a, b);
    for (auto & e : ev) { (void)hipEventDestroy(e); }
  }
  ctx->cluster_ready = true;
  return SWA_OK;
}

// what swa_d1_cluster_device left in HBM, for the callers that want it after all (-i, -s, -u, -w, --fastidious): swarm,
// generation and parent of every amplicon; any of the three may be null.  Valid until the next upload, index build,
// network or clustering call of this context.
extern "C" int swa_d1_cluster_fetch(swa_ctx * ctx, uint32_t * swarmid, uint32_t * generation, uint32_t * parent) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (!ctx->cluster_ready || !ctx->csr_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_cluster_fetch: no clustering in place (swa_d1_cluster_device)"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->db.n;
  const auto * label = static_cast<const uint32_t *>(ctx->d_cluster.ptr);
  const uint32_t * gen = label + n, * par = gen + n, * sid = par + n;
  for (uint32_t * out : {swarmid, generation, parent}) { swa_touch_pages(out, (uint64_t)n * sizeof(uint32_t)); }
  if (swarmid != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(swarmid, sid, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  if (generation != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(generation, gen, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  if (parent != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(parent, par, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}

// the deepest generation of the clustering in place (what the log calls "Max generations")
extern "C" uint32_t swa_d1_cluster_maxgen(const swa_ctx * ctx) { return ctx != nullptr && ctx->cluster_ready ? ctx->cluster_maxgen : 0u; }
