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
//   k_label_step      label[v] = min(label[v], label[u]) along every link u -> v — and label[v] = label[label[v]]: pointer
//                     jumping — until nothing changes
//   k_level_step      level-synchronous distances from the seeds + parents (atomicMin of the claiming ids)
//   rocPRIM           seeds -> swarm numbers (exclusive scan); members ordered by one stable radix sort of
//                     (swarm << 32 | generation) with the ids ascending as payload; swarm sizes -> begins (scan)
#include "swa_internal.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace {

constexpr uint32_t kUnset = SWA_NO_AMPLICON;

__global__ __launch_bounds__(256) void k_label_init(uint32_t * label, uint32_t * parent, uint32_t n) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) { label[v] = v; parent[v] = kUnset; }
}

__global__ __launch_bounds__(256) void k_label_step(const uint64_t * __restrict__ offsets, const uint32_t * __restrict__ nb, uint32_t n,
                                                    uint32_t * label, uint32_t * changed) {
  bool any = false;
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
    uint32_t lu = label[u];
    // pointer jumping (round 4): label[u] = l says "l reaches u"; whatever reaches l reaches u as well, so u may take l's
    // label at once instead of waiting for it to travel link by link — the sweeps until nothing changes fall from the
    // depth of the deepest swarm towards its logarithm
    const uint32_t ll = label[lu];
    if (ll < lu) { atomicMin(&label[u], ll); lu = ll; any = true; }
    for (uint64_t e = offsets[u]; e < offsets[u + 1]; ++e) {
      const uint32_t v = nb[e];
      if (lu < label[v]) { atomicMin(&label[v], lu); any = true; }
    }
  }
  if (any) { *changed = 1u; }
}

__global__ __launch_bounds__(256) void k_level_init(const uint32_t * __restrict__ label, uint32_t * gen, uint8_t * is_seed, uint32_t n) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const bool seed = label[v] == v;
    gen[v] = seed ? 0u : kUnset;
    is_seed[v] = seed ? 1 : 0;
  }
}

// nodes of generation level - 1 claim their unclaimed neighbours of the same swarm for `level`
__global__ __launch_bounds__(256) void k_level_step(const uint64_t * __restrict__ offsets, const uint32_t * __restrict__ nb, uint32_t n,
                                                    const uint32_t * __restrict__ label, uint32_t * gen, uint32_t * parent, uint32_t level,
                                                    uint32_t * grew) {
  bool any = false;
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
        any = true;
      }
    }
  }
  if (any) { *grew = 1u; }
}

__global__ __launch_bounds__(256) void k_swarm_keys(const uint32_t * __restrict__ label, const uint32_t * __restrict__ gen,
                                                    const uint32_t * __restrict__ seed_rank, uint32_t n, uint32_t * __restrict__ swarmid,
                                                    unsigned long long * __restrict__ keys, uint32_t * __restrict__ ids, uint32_t * sizes) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const uint32_t sid = seed_rank[label[v]];
    swarmid[v] = sid;
    keys[v] = ((unsigned long long)sid << 32) | gen[v];
    ids[v] = v;
    atomicAdd(&sizes[sid], 1u);
  }
}

struct widen_u8 { __host__ __device__ uint32_t operator()(uint8_t v) const { return (uint32_t)v; } };

int blocks_for(const swa_ctx * ctx, uint64_t items) {
  const uint64_t b = (items + 255) / 256, cap = (uint64_t)ctx->num_cus * 8;
  return (int)std::max<uint64_t>(1, std::min(b, cap));
}

}  // namespace

// loads this translation unit's code object (see swa_ctx_warmup): an empty launch
void swa_warm_cluster(swa_ctx * ctx) {
  hipLaunchKernelGGL(k_label_init, dim3(1), dim3(64), 0, ctx->stream, static_cast<uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr), 0u);
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
  // label | gen | parent | swarmid | ids_in | ids_out | seed_rank | sizes (n + 1) | begins (n + 2), then keys in/out, flags
  SWA_TRY(swa_reserve(ctx, ctx->d_cluster, (uint64_t)n * 4 * 9 + 64 + (uint64_t)n * 8 * 2 + n + 64));
  auto * label = static_cast<uint32_t *>(ctx->d_cluster.ptr);
  auto * gen = label + n, * par = gen + n, * sid = par + n, * ids_in = sid + n, * ids_out = ids_in + n, * seed_rank = ids_out + n;
  auto * sizes = seed_rank + n, * begins = sizes + n + 2;
  auto * keys_in = reinterpret_cast<unsigned long long *>(begins + n + 4 + ((n + 4) & 1u));
  auto * keys_out = keys_in + n;
  auto * is_seed = reinterpret_cast<uint8_t *>(keys_out + n);
  SWA_TRY(swa_reserve(ctx, ctx->d_flags, 16 * sizeof(uint32_t)));
  auto * flag = static_cast<uint32_t *>(ctx->d_flags.ptr) + 11;
  const dim3 g(blocks_for(ctx, n)), b(256);
  hipLaunchKernelGGL(k_label_init, g, b, 0, ctx->stream, label, par, n);
  for (;;) {
    SWA_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(uint32_t), ctx->stream));
    // a few sweeps per host round trip: convergence is monotone, extra sweeps are harmless
    for (int k = 0; k < 4; ++k) { hipLaunchKernelGGL(k_label_step, g, b, 0, ctx->stream, offsets, nb, n, label, flag); }
    uint32_t changed = 0;
    SWA_HIP(ctx, hipMemcpyAsync(&changed, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (changed == 0) { break; }
  }
  hipLaunchKernelGGL(k_level_init, g, b, 0, ctx->stream, label, gen, is_seed, n);
  for (uint32_t level = 1;; ++level) {
    SWA_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_level_step, g, b, 0, ctx->stream, offsets, nb, n, label, gen, par, level, flag);
    uint32_t grew = 0;
    SWA_HIP(ctx, hipMemcpyAsync(&grew, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (grew == 0) { ctx->cluster_maxgen = level - 1; break; }     // (the last level that took somebody in)
  }
  // swarm numbers = rank of the seed among the seeds
  size_t tmp_bytes = 0, need = 0;
  auto seeds32 = rocprim::make_transform_iterator(is_seed, widen_u8());
  (void)rocprim::exclusive_scan(nullptr, need, seeds32, seed_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream);
  tmp_bytes = std::max(tmp_bytes, need);
  (void)rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, ids_in, ids_out, (size_t)n, 0, 64, ctx->stream);
  tmp_bytes = std::max(tmp_bytes, need);
  (void)rocprim::exclusive_scan(nullptr, need, sizes, begins, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), ctx->stream);
  tmp_bytes = std::max(tmp_bytes, need);
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_hits, tmp_bytes + 16));
  need = tmp_bytes;
  SWA_HIP(ctx, rocprim::exclusive_scan(ctx->d_scan_hits.ptr, need, seeds32, seed_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(sizes, 0, ((uint64_t)n + 2) * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL(k_swarm_keys, g, b, 0, ctx->stream, label, gen, seed_rank, n, sid, keys_in, ids_in, sizes);
  need = tmp_bytes;
  SWA_HIP(ctx, rocprim::radix_sort_pairs(ctx->d_scan_hits.ptr, need, keys_in, keys_out, ids_in, ids_out, (size_t)n, 0, 64, ctx->stream));
  need = tmp_bytes;
  SWA_HIP(ctx, rocprim::exclusive_scan(ctx->d_scan_hits.ptr, need, sizes, begins, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), ctx->stream));
  // number of swarms = rank of a virtual seed behind the last amplicon = seed_rank[n-1] + is_seed[n-1]
  uint32_t last_rank = 0;
  uint8_t last_seed = 0;
  SWA_HIP(ctx, hipMemcpyAsync(&last_rank, seed_rank + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipMemcpyAsync(&last_seed, is_seed + (n - 1), 1, hipMemcpyDeviceToHost, ctx->stream));
  if (swarmid != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(swarmid, sid, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  if (generation != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(generation, gen, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  if (parent != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(parent, par, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  SWA_HIP(ctx, hipMemcpyAsync(order, ids_out, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipGetLastError());
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *nswarms = last_rank + last_seed;
  if (*nswarms > swarm_cap || swarm_begin == nullptr) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_d1_cluster_device: swarm table too small"); }
  SWA_HIP(ctx, hipMemcpyAsync(swarm_begin, begins, ((uint64_t)*nswarms + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
  if (swarmid != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(swarmid, sid, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  if (generation != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(generation, gen, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  if (parent != nullptr) { SWA_HIP(ctx, hipMemcpyAsync(parent, par, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}

// the deepest generation of the clustering in place (what the log calls "Max generations")
extern "C" uint32_t swa_d1_cluster_maxgen(const swa_ctx * ctx) { return ctx != nullptr && ctx->cluster_ready ? ctx->cluster_maxgen : 0u; }
