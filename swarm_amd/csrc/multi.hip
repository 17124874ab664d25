// multi.hip — the d = 1 path on several GPUs of one node, from C++ (no Python, no torch): one context, one
// stream and one host thread per GPU inside this process, RCCL over xGMI for the exchange.
//
// The reference fans its probe loop out over pthreads that share one hash table
// (src/algod1.cc:641-669, 1166-1167; src/utils/threads.h:145-162).  Here the database is replicated
// (288 GB per GPU is never the limit) and the PROBING is divided by ownership of anchor groups
// (swa_d1_set_ownership: rank r serves the groups whose key maps to r, with all their members), so every
// link of the network is found by exactly one rank.  Exchange, as SURVEY 8e / the north star name it: the
// per-rank hit counts are known on the host when the kernels return, then every rank's flat link list is
// gathered on rank 0 — the one consumer: grouped ncclSend / ncclRecv (round 2 all-gathered the whole network into every
// GPU) —, and the CSR is made there by the partition + row kernels of the single-GPU step (swa_d1_csr_from_lists).  Fastidious: the heavy amplicons are split
// (swa_d1_fastidious_shard), graft_cand is combined with swa_rccl().AllReduce(min) (src/algod1.cc:244-258 keeps
// the smallest heavy id), the two heavy-side counters add up.
//
// The same code runs with several contexts on ONE device (SWARM_AMD_DEVICES=0,0: what a one-GPU box can
// test); RCCL refuses two ranks on one GPU, so that configuration moves the same bytes with device-to-
// device copies instead.  Results are identical to a single GPU by construction and by test
// (tests/test_multi_gpu.py).
#include "swa_internal.h"

#include "rccl_late.h"

#include <algorithm>
#include <functional>
#include <set>
#include <thread>
#include <vector>

struct swa_multi {
  std::vector<swa_ctx *> ctx;
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;       // empty: in-process copies (several ranks on one device)
  std::vector<swa_dbuf> links, gathered;
  std::vector<swa_dbuf> routed, routed_counts, inbox;   // routed index build: a rank's outgoing id lists, their counts, what it received
  std::vector<uint64_t> link_cap;
  std::string err;
};

namespace {

// The ranks' shares of the d >= 2 graph arrive SORTED (swa_dn_graph_compute ends with a sort of its own share) and no key
// occurs in two of them (a pair is found through the first window it shares, and that window's group lives on one rank),
// so the whole is a merge, not a sort: element i of list r goes to i + (the number of smaller keys in every other list) —
// `world - 1` binary searches per element over lists that lie in L2 — where round 3 ran a 64-bit radix sort (8 passes).
// Equal keys across lists (which cannot happen) would still land in distinct places: upper bound below r, lower bound above.
struct MergeArgs {
  const unsigned long long * keys; const uint32_t * vals;     // the gathered lists, list r = [at[r], at[r + 1])
  uint64_t at[65];
  uint32_t world;
  unsigned long long * out_keys; uint32_t * out_vals;
};
__global__ __launch_bounds__(256) void k_merge_sorted_lists(const MergeArgs a) {
  const uint64_t all = a.at[a.world];
  for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < all; idx += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t r = 0;
    while (idx >= a.at[r + 1]) { ++r; }
    const unsigned long long key = a.keys[idx];
    uint64_t pos = idx - a.at[r];
    for (uint32_t s2 = 0; s2 < a.world; ++s2) {
      if (s2 == r) { continue; }
      uint64_t lo = a.at[s2], hi = a.at[s2 + 1];
      while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const unsigned long long k2 = a.keys[mid];
        if (s2 < r ? k2 <= key : k2 < key) { lo = mid + 1; } else { hi = mid; }
      }
      pos += lo - a.at[s2];
    }
    a.out_keys[pos] = key; a.out_vals[pos] = a.vals[idx];
  }
}

__global__ __launch_bounds__(256) void k_min_u32(uint32_t * __restrict__ acc, const uint32_t * __restrict__ other, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    acc[i] = min(acc[i], other[i]);
  }
}

int blocks_for(const swa_ctx * ctx, uint64_t items) {
  const uint64_t b = (items + 255) / 256, cap = (uint64_t)ctx->num_cus * 8;
  return (int)std::max<uint64_t>(1, std::min(b, cap));
}

// runs f(rank) on one host thread per rank; first non-zero status wins
int on_all(swa_multi * m, const std::function<int(int)> & f) {
  const int world = (int)m->ctx.size();
  std::vector<int> rc((size_t)world, SWA_OK);
  if (world == 1) { rc[0] = f(0); }
  else {
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r) { threads.emplace_back([&, r]() { rc[(size_t)r] = f(r); }); }
    for (auto & t : threads) { t.join(); }
  }
  for (int r = 0; r < world; ++r) {
    if (rc[(size_t)r] != SWA_OK) { m->err = "rank " + std::to_string(r) + ": " + swa_last_error(m->ctx[(size_t)r]); return rc[(size_t)r]; }
  }
  return SWA_OK;
}

int fail(swa_multi * m, int code, const std::string & msg) { m->err = msg; return code; }

#define NCCL_OK(m, expr)                                                                               \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess) { return fail((m), SWA_E_DEVICE, std::string(#expr) + ": " + swa_rccl().GetErrorString(r_)); } \
  } while (0)

// every rank's buffer src[r] (count[r] elements of `bytes_per` bytes) into dst[k] + prefix(r) on every rank k
int all_gather_v(swa_multi * m, const std::vector<const void *> & src, const std::vector<void *> & dst,
                 const std::vector<uint64_t> & count, size_t bytes_per) {
  const int world = (int)m->ctx.size();
  std::vector<uint64_t> at((size_t)world + 1, 0);
  for (int r = 0; r < world; ++r) { at[(size_t)r + 1] = at[(size_t)r] + count[(size_t)r]; }
  if (!m->comms.empty()) {
    NCCL_OK(m, swa_rccl().GroupStart());
    for (int k = 0; k < world; ++k) {
      for (int r = 0; r < world; ++r) {
        if (count[(size_t)r] == 0) { continue; }
        char * recv = static_cast<char *>(dst[(size_t)k]) + at[(size_t)r] * bytes_per;
        // (the send buffer only matters on the root)
        NCCL_OK(m, swa_rccl().Broadcast(k == r ? src[(size_t)r] : recv, recv, count[(size_t)r] * bytes_per, ncclUint8, r, m->comms[(size_t)k],
                                 m->ctx[(size_t)k]->stream));
      }
    }
    NCCL_OK(m, swa_rccl().GroupEnd());
  } else {
    for (int k = 0; k < world; ++k) {
      if (hipSetDevice(m->devices[(size_t)k]) != hipSuccess) { return fail(m, SWA_E_DEVICE, "hipSetDevice"); }
      for (int r = 0; r < world; ++r) {
        if (count[(size_t)r] == 0) { continue; }
        char * recv = static_cast<char *>(dst[(size_t)k]) + at[(size_t)r] * bytes_per;
        if (hipMemcpyAsync(recv, src[(size_t)r], count[(size_t)r] * bytes_per, hipMemcpyDefault, m->ctx[(size_t)k]->stream) != hipSuccess) {
          return fail(m, SWA_E_DEVICE, "device-to-device copy of a link list failed");
        }
      }
    }
  }
  for (int k = 0; k < world; ++k) {
    if (hipSetDevice(m->devices[(size_t)k]) != hipSuccess || hipStreamSynchronize(m->ctx[(size_t)k]->stream) != hipSuccess) {
      return fail(m, SWA_E_DEVICE, "synchronising the exchange failed");
    }
  }
  return SWA_OK;
}

// every rank's buffer src[r] (count[r] elements of `bytes_per` bytes) into dst + prefix(r) on rank 0 ONLY: the consumer
// of the gathered lists is the rank that builds the CSR, so nothing is sent to the others (an all-gather moved the
// whole network into every GPU: 0.9 GB per rank at 8 x 10 M amplicons).  RCCL: one ncclSend per rank, the matching
// ncclRecv on rank 0, all in one group; ranks sharing a device: device-to-device copies.
int gather_to_root_v(swa_multi * m, const std::vector<const void *> & src, void * dst, const std::vector<uint64_t> & count, size_t bytes_per) {
  const int world = (int)m->ctx.size();
  swa_ctx * c0 = m->ctx[0];
  std::vector<uint64_t> at((size_t)world + 1, 0);
  for (int r = 0; r < world; ++r) { at[(size_t)r + 1] = at[(size_t)r] + count[(size_t)r]; }
  if (!m->comms.empty()) { NCCL_OK(m, swa_rccl().GroupStart()); }
  for (int r = 0; r < world; ++r) {
    if (count[(size_t)r] == 0) { continue; }
    char * recv = static_cast<char *>(dst) + at[(size_t)r] * bytes_per;
    if (!m->comms.empty() && r != 0) {
      NCCL_OK(m, swa_rccl().Send(src[(size_t)r], count[(size_t)r] * bytes_per, ncclUint8, 0, m->comms[(size_t)r], m->ctx[(size_t)r]->stream));
      NCCL_OK(m, swa_rccl().Recv(recv, count[(size_t)r] * bytes_per, ncclUint8, r, m->comms[0], c0->stream));
    } else {
      if (hipSetDevice(c0->device) != hipSuccess ||
          hipMemcpyAsync(recv, src[(size_t)r], count[(size_t)r] * bytes_per, hipMemcpyDefault, c0->stream) != hipSuccess) {
        return fail(m, SWA_E_DEVICE, "device-to-device copy of a rank's list failed");
      }
    }
  }
  if (!m->comms.empty()) { NCCL_OK(m, swa_rccl().GroupEnd()); }
  for (int k = 0; k < world; ++k) {
    if (hipSetDevice(m->devices[(size_t)k]) != hipSuccess || hipStreamSynchronize(m->ctx[(size_t)k]->stream) != hipSuccess) {
      return fail(m, SWA_E_DEVICE, "synchronising the exchange failed");
    }
  }
  return SWA_OK;
}

}  // namespace

extern "C" int swa_multi_create(const int * devices, int ndevices, swa_multi ** out) {
  if (out == nullptr || devices == nullptr || ndevices < 1 || ndevices > 64) { return SWA_E_ARG; }
  auto * m = new swa_multi();
  *out = m;
  m->devices.assign(devices, devices + ndevices);
  // (per-rank buffers first: swa_multi_destroy walks them for every context that exists, also after a failure below)
  m->links.resize((size_t)ndevices);
  m->gathered.resize((size_t)ndevices);
  m->routed.resize((size_t)ndevices);
  m->routed_counts.resize((size_t)ndevices);
  m->inbox.resize((size_t)ndevices);
  m->link_cap.assign((size_t)ndevices, 0);
  for (int r = 0; r < ndevices; ++r) {
    swa_ctx * c = nullptr;
    const int rc = swa_ctx_create(devices[r], nullptr, &c);
    if (rc != SWA_OK) { m->err = "no usable gfx950 GPU with index " + std::to_string(devices[r]); return rc; }
    m->ctx.push_back(c);
  }
  const std::set<int> distinct(m->devices.begin(), m->devices.end());
  const char * force = getenv("SWARM_AMD_FORCE_RCCL");       // test hook: RCCL even for a single rank
  if ((int)distinct.size() == ndevices && (ndevices > 1 || (force != nullptr && force[0] == '1'))) {
    if (!swa_rccl().error.empty()) { m->err = swa_rccl().error; return SWA_E_DEVICE; }
    m->comms.resize((size_t)ndevices);
    const ncclResult_t r = swa_rccl().CommInitAll(m->comms.data(), ndevices, m->devices.data());
    if (r != ncclSuccess) { m->comms.clear(); m->err = std::string("ncclCommInitAll: ") + swa_rccl().GetErrorString(r); return SWA_E_DEVICE; }
  }
  return SWA_OK;
}

extern "C" void swa_multi_destroy(swa_multi * m) {
  if (m == nullptr) { return; }
  for (size_t r = 0; r < m->ctx.size(); ++r) {
    (void)hipSetDevice(m->devices[r]);
    swa_release(m->links[r]);
    swa_release(m->gathered[r]);
    swa_release(m->routed[r]);
    swa_release(m->routed_counts[r]);
    swa_release(m->inbox[r]);
  }
  for (auto & c : m->comms) { (void)swa_rccl().CommDestroy(c); }
  for (auto * c : m->ctx) { swa_ctx_destroy(c); }
  delete m;
}

extern "C" int swa_multi_size(const swa_multi * m) { return m != nullptr ? (int)m->ctx.size() : 0; }
extern "C" int swa_multi_uses_rccl(const swa_multi * m) { return m != nullptr && !m->comms.empty() ? 1 : 0; }
extern "C" swa_ctx * swa_multi_ctx(swa_multi * m, int rank) {
  return (m != nullptr && rank >= 0 && rank < (int)m->ctx.size()) ? m->ctx[(size_t)rank] : nullptr;
}
extern "C" const char * swa_multi_last_error(const swa_multi * m) { return m != nullptr ? m->err.c_str() : "null handle"; }

extern "C" int swa_multi_db_upload(swa_multi * m, const swa_db_view * host) {
  if (m == nullptr) { return SWA_E_ARG; }
  return on_all(m, [&](int r) { return swa_db_upload(m->ctx[(size_t)r], host); });
}

// Index build of all ranks WITHOUT any of them walking the whole database (swa_d1_route_slice_records /
// swa_d1_index_build_records): every rank keys its own slice, the finished key records travel all-to-all — grouped
// ncclSend / ncclRecv over RCCL, device-to-device copies when the ranks share a device —, every rank builds its indexes
// from what it received, starting at the partition (round 6: rounds 4-5 sent the ids and keyed them again at the owner,
// one random line fetch each).
// *routed = false when a destination region overflowed (never with hashed ownership): the caller builds the other way.
static int routed_index_build(swa_multi * m, std::vector<int> & dup, bool * routed) {
  const int world = (int)m->ctx.size();
  const uint32_t n = m->ctx[0]->db.n;
  *routed = false;
  std::vector<uint32_t> first((size_t)world + 1, 0);
  for (int r = 0; r <= world; ++r) { first[(size_t)r] = (uint32_t)((uint64_t)n * (uint64_t)r / (uint64_t)world); }
  const uint64_t cap = 3ull * ((uint64_t)n / (uint64_t)world + 1) / (2ull * (uint64_t)world) + 1024;
  const size_t lists = 2 * (size_t)world;
  std::vector<std::vector<uint32_t>> cnt((size_t)world, std::vector<uint32_t>(lists + 1, 0));
  int rc = on_all(m, [&](int r) {
    swa_ctx * c = m->ctx[(size_t)r];
    SWA_HIP(c, hipSetDevice(c->device));
    SWA_TRY(swa_d1_set_ownership(c, (uint32_t)r, (uint32_t)world));
    SWA_TRY(swa_reserve(c, m->routed[(size_t)r], lists * cap * sizeof(uint64_t)));
    SWA_TRY(swa_reserve(c, m->routed_counts[(size_t)r], (lists + 1) * sizeof(uint32_t)));
    SWA_TRY(swa_d1_route_slice_records(c, first[(size_t)r], first[(size_t)r + 1] - first[(size_t)r], (uint32_t)world,
                                       static_cast<uint64_t *>(m->routed[(size_t)r].ptr), cap, static_cast<uint32_t *>(m->routed_counts[(size_t)r].ptr)));
    SWA_HIP(c, hipMemcpy(cnt[(size_t)r].data(), m->routed_counts[(size_t)r].ptr, (lists + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return (int)SWA_OK;
  });
  if (rc != SWA_OK) { return rc; }
  for (int r = 0; r < world; ++r) { if (cnt[(size_t)r][lists] != 0) { return SWA_OK; } }
  // rank r receives, per index, the lists of the sources in rank order: inbox[r] = [index 0: s = 0, 1, ..][index 1: ..]
  std::vector<uint32_t> m0((size_t)world, 0), m1((size_t)world, 0);
  for (int r = 0; r < world; ++r) {
    for (int s2 = 0; s2 < world; ++s2) { m0[(size_t)r] += cnt[(size_t)s2][(size_t)r]; m1[(size_t)r] += cnt[(size_t)s2][(size_t)world + (size_t)r]; }
    swa_ctx * c = m->ctx[(size_t)r];
    if (hipSetDevice(c->device) != hipSuccess) { return fail(m, SWA_E_DEVICE, "hipSetDevice"); }
    rc = swa_reserve(c, m->inbox[(size_t)r], ((uint64_t)m0[(size_t)r] + m1[(size_t)r] + 1) * sizeof(uint64_t));
    if (rc != SWA_OK) { return fail(m, rc, swa_last_error(c)); }
  }
  if (!m->comms.empty()) { NCCL_OK(m, swa_rccl().GroupStart()); }
  for (int r = 0; r < world; ++r) {
    uint64_t at[2] = {0, m0[(size_t)r]};
    for (int s2 = 0; s2 < world; ++s2) {
      for (int index = 0; index < 2; ++index) {
        const size_t list = (size_t)index * (size_t)world + (size_t)r;
        const uint32_t count = cnt[(size_t)s2][list];
        if (count == 0) { continue; }
        const uint64_t * src = static_cast<const uint64_t *>(m->routed[(size_t)s2].ptr) + list * cap;
        uint64_t * dst = static_cast<uint64_t *>(m->inbox[(size_t)r].ptr) + at[index];
        at[index] += count;
        if (!m->comms.empty()) {
          NCCL_OK(m, swa_rccl().Send(src, count, ncclUint64, r, m->comms[(size_t)s2], m->ctx[(size_t)s2]->stream));
          NCCL_OK(m, swa_rccl().Recv(dst, count, ncclUint64, s2, m->comms[(size_t)r], m->ctx[(size_t)r]->stream));
        } else {
          if (hipSetDevice(m->devices[(size_t)r]) != hipSuccess ||
              hipMemcpyAsync(dst, src, (size_t)count * sizeof(uint64_t), hipMemcpyDefault, m->ctx[(size_t)r]->stream) != hipSuccess) {
            return fail(m, SWA_E_DEVICE, "device-to-device copy of a routed record list failed");
          }
        }
      }
    }
  }
  if (!m->comms.empty()) { NCCL_OK(m, swa_rccl().GroupEnd()); }
  for (int k = 0; k < world; ++k) {
    if (hipSetDevice(m->devices[(size_t)k]) != hipSuccess || hipStreamSynchronize(m->ctx[(size_t)k]->stream) != hipSuccess) {
      return fail(m, SWA_E_DEVICE, "synchronising the exchange of the routed record lists failed");
    }
  }
  *routed = true;
  return on_all(m, [&](int r) {
    swa_ctx * c = m->ctx[(size_t)r];
    const uint64_t * in = static_cast<const uint64_t *>(m->inbox[(size_t)r].ptr);
    return swa_d1_index_build_records(c, in, m0[(size_t)r], in + m0[(size_t)r], m1[(size_t)r], &dup[(size_t)r]);
  });
}

// the whole network of the database as a CSR on the host (buffers and capacity protocol of swa_d1_network)
extern "C" int swa_multi_d1_network(swa_multi * m, int no_cluster_breaking, uint64_t * offsets, uint32_t * neighbours, uint64_t cap,
                                    uint64_t * total, int * has_duplicates) {
  if (m == nullptr || offsets == nullptr || total == nullptr || (neighbours == nullptr && cap != 0)) { return SWA_E_ARG; }
  const int world = (int)m->ctx.size();
  const uint32_t n = m->ctx[0]->db.n;
  if (n == 0) { return fail(m, SWA_E_ARG, "swa_multi_d1_network: no database"); }
  // 1. every rank indexes and probes the anchor groups it owns
  std::vector<int> dup((size_t)world, 0);
  std::vector<uint64_t> count((size_t)world, 0);
  // (SWARM_AMD_MULTI_BUILD=streamed: every rank walks the whole database for the keys it owns, as before the routed build)
  const char * env_build = getenv("SWARM_AMD_MULTI_BUILD");
  bool routed = false;
  int rc = SWA_OK;
  if (world > 1 && !(env_build != nullptr && env_build[0] == 's')) {
    rc = routed_index_build(m, dup, &routed);
    if (rc != SWA_OK) {
      if (has_duplicates != nullptr) { *has_duplicates = 0; for (int d : dup) { *has_duplicates |= d; } }
      return rc;
    }
  }
  rc = on_all(m, [&](int r) {
    swa_ctx * c = m->ctx[(size_t)r];
    SWA_TRY(swa_d1_set_ownership(c, (uint32_t)r, (uint32_t)world));
    int rc2 = routed ? (int)SWA_OK : swa_d1_index_build(c, &dup[(size_t)r]);
    if (rc2 != SWA_OK) { return rc2; }
    if (m->link_cap[(size_t)r] == 0) { m->link_cap[(size_t)r] = 4ull * n / (uint64_t)world + 65536; }
    for (;;) {
      SWA_HIP(c, hipSetDevice(c->device));
      SWA_TRY(swa_reserve(c, m->links[(size_t)r], m->link_cap[(size_t)r] * sizeof(uint64_t)));
      rc2 = swa_d1_network_edges_device(c, no_cluster_breaking, 0, n, static_cast<uint64_t *>(m->links[(size_t)r].ptr),
                                        m->link_cap[(size_t)r], &count[(size_t)r]);
      if (rc2 == SWA_E_CAPACITY) { m->link_cap[(size_t)r] = count[(size_t)r] + 1024; continue; }
      if (rc2 == SWA_E_DUPLICATES) { dup[(size_t)r] = 1; }     // (identical sequences inside a prefix group this rank owns: the pair pass met them)
      return rc2;
    }
  });
  if (has_duplicates != nullptr) { *has_duplicates = 0; for (int d : dup) { *has_duplicates |= d; } }
  if (rc != SWA_OK) { return rc; }
  // 2. the link lists to rank 0, which builds the CSR (counts are on the host already)
  uint64_t all = 0;
  for (uint64_t c : count) { all += c; }
  *total = all;
  std::vector<const void *> src((size_t)world);
  for (int r = 0; r < world; ++r) { src[(size_t)r] = m->links[(size_t)r].ptr; }
  {
    swa_ctx * c = m->ctx[0];
    if (hipSetDevice(c->device) != hipSuccess) { return fail(m, SWA_E_DEVICE, "hipSetDevice"); }
    rc = swa_reserve(c, m->gathered[0], (all + 2) * sizeof(uint64_t));
    if (rc != SWA_OK) { return fail(m, rc, swa_last_error(c)); }
  }
  // (SWARM_AMD_MULTI_EXCHANGE=allgather: the literal all-gather of SURVEY 8e — every GPU ends with every list)
  const char * env_x = getenv("SWARM_AMD_MULTI_EXCHANGE");
  if (all != 0 && env_x != nullptr && env_x[0] == 'a') {
    std::vector<void *> dst((size_t)world);
    for (int r = 0; r < world; ++r) {
      swa_ctx * c = m->ctx[(size_t)r];
      if (hipSetDevice(c->device) != hipSuccess) { return fail(m, SWA_E_DEVICE, "hipSetDevice"); }
      rc = swa_reserve(c, m->gathered[(size_t)r], (all + 2) * sizeof(uint64_t));
      if (rc != SWA_OK) { return fail(m, rc, swa_last_error(c)); }
      dst[(size_t)r] = m->gathered[(size_t)r].ptr;
    }
    rc = all_gather_v(m, src, dst, count, sizeof(uint64_t));
    if (rc != SWA_OK) { return rc; }
  } else if (all != 0) {
    rc = gather_to_root_v(m, src, m->gathered[0].ptr, count, sizeof(uint64_t));
    if (rc != SWA_OK) { return rc; }
  }
  // 3. rank 0: the gathered lists -> CSR -> host.  The partition + row kernels of the single-GPU step (d1.hip: csr_from_chunks,
  // the ranks' lists as the chunks of its first level) where round 3 ran a 64-bit radix sort over all links.
  swa_ctx * c0 = m->ctx[0];
  if (hipSetDevice(c0->device) != hipSuccess) { return fail(m, SWA_E_DEVICE, "hipSetDevice"); }
  auto * keys_in = static_cast<unsigned long long *>(m->gathered[0].ptr);
  auto stage = [&]() -> int {
    c0->csr_ready = false;                                   // (d_offsets_tmp / d_nb_tmp now hold the gathered network)
    SWA_TRY(swa_reserve(c0, c0->d_offsets_tmp, ((uint64_t)n + 1) * sizeof(uint64_t)));
    SWA_TRY(swa_reserve(c0, c0->d_nb_tmp, (all + 1) * sizeof(uint32_t)));
    std::vector<uint64_t> starts((size_t)world, 0);
    for (int r = 1; r < world; ++r) { starts[(size_t)r] = starts[(size_t)r - 1] + count[(size_t)r - 1]; }
    SWA_TRY(swa_d1_csr_from_lists(c0, keys_in, starts.data(), count.data(), (uint32_t)world, static_cast<uint64_t *>(c0->d_offsets_tmp.ptr),
                                  static_cast<uint32_t *>(c0->d_nb_tmp.ptr), all + 1));
    SWA_HIP(c0, hipMemcpyAsync(offsets, c0->d_offsets_tmp.ptr, ((uint64_t)n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c0->stream));
    if (all != 0 && all <= cap) {
      SWA_HIP(c0, hipMemcpyAsync(neighbours, c0->d_nb_tmp.ptr, all * sizeof(uint32_t), hipMemcpyDeviceToHost, c0->stream));
    }
    SWA_HIP(c0, hipGetLastError());
    SWA_HIP(c0, hipStreamSynchronize(c0->stream));
    return SWA_OK;
  };
  rc = stage();
  if (rc != SWA_OK) { return fail(m, rc, swa_last_error(c0)); }
  if (all > cap) { return fail(m, SWA_E_CAPACITY, "swa_multi_d1_network: neighbour buffer too small"); }
  return SWA_OK;
}

// B2 on all GPUs: heavy amplicons split over the ranks, graft_cand combined with MIN, heavy counters summed
extern "C" int swa_multi_d1_fastidious(swa_multi * m, const uint8_t * is_light, uint64_t light_nt, uint32_t bloom_bits,
                                       uint32_t * graft_cand, uint64_t * counters) {
  if (m == nullptr || is_light == nullptr || graft_cand == nullptr || counters == nullptr) { return SWA_E_ARG; }
  const int world = (int)m->ctx.size();
  const uint32_t n = m->ctx[0]->db.n;
  std::vector<std::vector<uint32_t>> scratch((size_t)world);
  std::vector<std::vector<uint64_t>> cnt((size_t)world, std::vector<uint64_t>(8, 0));
  int rc = on_all(m, [&](int r) {
    swa_ctx * c = m->ctx[(size_t)r];
    if (!c->d1_ready) { int dup = 0; SWA_TRY(swa_d1_index_build(c, &dup)); }
    uint32_t * host = graft_cand;
    if (r != 0) { scratch[(size_t)r].resize(n); host = scratch[(size_t)r].data(); }   // (the per-rank host copy is not the result)
    return swa_d1_fastidious_shard(c, is_light, light_nt, bloom_bits, (uint32_t)r, (uint32_t)world, host, cnt[(size_t)r].data());
  });
  if (rc != SWA_OK) { return rc; }
  for (int k = 0; k < 5; ++k) { counters[k] = cnt[0][(size_t)k]; }
  for (int r = 1; r < world; ++r) { counters[1] += cnt[(size_t)r][1]; counters[2] += cnt[(size_t)r][2]; }
  if (world == 1 && m->comms.empty()) { return SWA_OK; }
  // element-wise minimum of the ranks' graft_cand arrays, which are still in HBM (d_graft)
  if (!m->comms.empty()) {
    NCCL_OK(m, swa_rccl().GroupStart());
    for (int r = 0; r < world; ++r) {
      NCCL_OK(m, swa_rccl().AllReduce(m->ctx[(size_t)r]->d_graft.ptr, m->ctx[(size_t)r]->d_graft.ptr, n, ncclUint32, ncclMin, m->comms[(size_t)r],
                               m->ctx[(size_t)r]->stream));
    }
    NCCL_OK(m, swa_rccl().GroupEnd());
  } else {
    swa_ctx * c0 = m->ctx[0];
    if (hipSetDevice(c0->device) != hipSuccess) { return fail(m, SWA_E_DEVICE, "hipSetDevice"); }
    rc = swa_reserve(c0, m->gathered[0], (uint64_t)n * sizeof(uint32_t));
    if (rc != SWA_OK) { return fail(m, rc, swa_last_error(c0)); }
    for (int r = 1; r < world; ++r) {
      if (hipMemcpyAsync(m->gathered[0].ptr, m->ctx[(size_t)r]->d_graft.ptr, (uint64_t)n * sizeof(uint32_t), hipMemcpyDefault, c0->stream) != hipSuccess) {
        return fail(m, SWA_E_DEVICE, "device-to-device copy of graft candidates failed");
      }
      hipLaunchKernelGGL(k_min_u32, dim3(blocks_for(c0, n)), dim3(256), 0, c0->stream, static_cast<uint32_t *>(c0->d_graft.ptr),
                         static_cast<const uint32_t *>(m->gathered[0].ptr), (uint64_t)n);
    }
  }
  swa_ctx * c0 = m->ctx[0];
  if (hipSetDevice(c0->device) != hipSuccess ||
      hipMemcpyAsync(graft_cand, c0->d_graft.ptr, (uint64_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c0->stream) != hipSuccess ||
      hipStreamSynchronize(c0->stream) != hipSuccess) {
    return fail(m, SWA_E_DEVICE, "download of the combined graft candidates failed");
  }
  return SWA_OK;
}

// ---- d >= 2 on several GPUs: the bulk graph (dn_graph.hip) divided by ownership of window groups ------------------
// Replaces the scan fan-out of src/scan.cc:221-256 under the loop of src/algo.cc:505-602: every rank finds the pairs of
// the window groups it owns (swa_dn_set_ownership) and aligns them; the accepted (query, target, diff) triples travel
// to rank 0 — ncclSend / ncclRecv over RCCL, device-to-device copies when ranks share a device —, which sorts them
// into the CSR of swa_dn_graph.  The greedy walk over it stays on the host (cluster_dn.cpp).
extern "C" int swa_multi_dn_begin(swa_multi * m, uint64_t mismatch, uint64_t gapopen, uint64_t gapextend, uint64_t d) {
  if (m == nullptr) { return SWA_E_ARG; }
  return on_all(m, [&](int r) {
    swa_ctx * c = m->ctx[(size_t)r];
    SWA_TRY(swa_qgram_build(c));
    return swa_search_begin(c, mismatch, gapopen, gapextend, d);
  });
}

extern "C" int swa_multi_dn_graph_supported(swa_multi * m) {
  if (m == nullptr) { return 0; }
  for (swa_ctx * c : m->ctx) { if (swa_dn_graph_supported(c) == 0) { return 0; } }
  return 1;
}

extern "C" int swa_multi_dn_graph(swa_multi * m, int no_cluster_breaking, uint64_t * offsets, uint32_t * neighbours, uint8_t * diffs,
                                  uint64_t cap, uint64_t * total) {
  if (m == nullptr || offsets == nullptr || total == nullptr || (cap != 0 && (neighbours == nullptr || diffs == nullptr))) { return SWA_E_ARG; }
  const int world = (int)m->ctx.size();
  int rc = on_all(m, [&](int r) {
    swa_ctx * c = m->ctx[(size_t)r];
    SWA_TRY(swa_dn_set_ownership(c, (uint32_t)r, (uint32_t)world));
    SWA_TRY(swa_dn_graph_compute(c, no_cluster_breaking));
    // (the share's final sort is still queued on this rank's stream, and the streams of the contexts do not order against
    // each other: the copies below run on rank 0's stream when the ranks share a device — ADVICE r03)
    SWA_HIP(c, hipStreamSynchronize(c->stream));
    return (int)SWA_OK;
  });
  // the shares are computed: a later single-context call on any of these contexts sees the whole database again
  for (swa_ctx * c : m->ctx) { (void)swa_dn_set_ownership(c, 0, 1); }
  if (rc != SWA_OK) { return rc; }
  swa_ctx * c0 = m->ctx[0];
  uint64_t all = 0;
  std::vector<uint64_t> at((size_t)world + 1, 0);
  for (int r = 0; r < world; ++r) { at[(size_t)r] = all; all += m->ctx[(size_t)r]->dn_edges; }
  at[(size_t)world] = all;
  if (hipSetDevice(c0->device) != hipSuccess) { return fail(m, SWA_E_DEVICE, "hipSetDevice"); }
  // rank 0's buffers: [all keys | sorted keys] and [all diffs | sorted diffs]
  rc = swa_reserve(c0, m->gathered[0], (2 * all + 2) * sizeof(uint64_t));
  if (rc == SWA_OK) { rc = swa_reserve(c0, m->links[0], (2 * all + 2) * sizeof(uint32_t)); }
  if (rc != SWA_OK) { return fail(m, rc, swa_last_error(c0)); }
  auto * keys = static_cast<unsigned long long *>(m->gathered[0].ptr);
  auto * vals = static_cast<uint32_t *>(m->links[0].ptr);
  if (!m->comms.empty()) { NCCL_OK(m, swa_rccl().GroupStart()); }
  for (int r = 0; r < world; ++r) {
    swa_ctx * c = m->ctx[(size_t)r];
    const uint64_t cnt = c->dn_edges;
    if (cnt == 0) { continue; }
    const auto * k_src = static_cast<const unsigned long long *>(c->d_dn_keys.ptr) + c->dn_work;
    const auto * v_src = static_cast<const uint32_t *>(c->d_dn_vals.ptr) + c->dn_work;
    if (!m->comms.empty() && r != 0) {
      NCCL_OK(m, swa_rccl().Send(k_src, cnt, ncclUint64, 0, m->comms[(size_t)r], c->stream));
      NCCL_OK(m, swa_rccl().Recv(keys + at[(size_t)r], cnt, ncclUint64, r, m->comms[0], c0->stream));
      NCCL_OK(m, swa_rccl().Send(v_src, cnt, ncclUint32, 0, m->comms[(size_t)r], c->stream));
      NCCL_OK(m, swa_rccl().Recv(vals + at[(size_t)r], cnt, ncclUint32, r, m->comms[0], c0->stream));
    } else {
      if (hipSetDevice(c0->device) != hipSuccess ||
          hipMemcpyAsync(keys + at[(size_t)r], k_src, cnt * sizeof(uint64_t), hipMemcpyDefault, c0->stream) != hipSuccess ||
          hipMemcpyAsync(vals + at[(size_t)r], v_src, cnt * sizeof(uint32_t), hipMemcpyDefault, c0->stream) != hipSuccess) {
        return fail(m, SWA_E_DEVICE, "device-to-device copy of a rank's share of the graph failed");
      }
    }
  }
  if (!m->comms.empty()) { NCCL_OK(m, swa_rccl().GroupEnd()); }
  for (int k = 0; k < world; ++k) {
    if (hipSetDevice(m->devices[(size_t)k]) != hipSuccess || hipStreamSynchronize(m->ctx[(size_t)k]->stream) != hipSuccess) {
      return fail(m, SWA_E_DEVICE, "synchronising the exchange of the graph failed");
    }
  }
  if (hipSetDevice(c0->device) != hipSuccess) { return fail(m, SWA_E_DEVICE, "hipSetDevice"); }
  auto stage = [&]() -> int {
    if (all != 0) {
      MergeArgs ma{};
      ma.keys = keys; ma.vals = vals; ma.world = (uint32_t)world;
      for (int r = 0; r <= world; ++r) { ma.at[r] = at[(size_t)r]; }
      ma.out_keys = keys + all + 1; ma.out_vals = vals + all + 1;
      hipLaunchKernelGGL(k_merge_sorted_lists, dim3(blocks_for(c0, all)), dim3(256), 0, c0->stream, ma);
      SWA_HIP(c0, hipGetLastError());
    }
    return swa_dn_graph_emit(c0, all != 0 ? keys + all + 1 : nullptr, all != 0 ? vals + all + 1 : nullptr, all, offsets, neighbours, diffs, cap, total);
  };
  rc = stage();
  if (rc != SWA_OK) { return fail(m, rc, swa_last_error(c0)); }
  return SWA_OK;
}

// out3 = {q-gram comparisons, aligned pairs, kernel launches} over all ranks
extern "C" int swa_multi_dn_graph_totals(swa_multi * m, uint64_t * out3) {
  if (m == nullptr || out3 == nullptr) { return SWA_E_ARG; }
  out3[0] = out3[1] = out3[2] = 0;
  for (swa_ctx * c : m->ctx) {
    uint64_t t[3] = {0, 0, 0};
    (void)swa_dn_graph_totals(c, t);
    out3[0] += t[0]; out3[1] += t[1]; out3[2] = std::max(out3[2], t[2]);
  }
  return SWA_OK;
}
