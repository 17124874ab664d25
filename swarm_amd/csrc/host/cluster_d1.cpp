// cluster_d1.cpp — host side of the d = 1 path: greedy single-linkage agglomeration
// over the GPU-returned neighbour lists, fastidious bookkeeping and grafting, writers.
//
// Behaviour mirrors algo_d1_run's host part (src/algod1.cc:1185-1280 clustering,
// 214-336 grafting, 755-1095 writers) so that every output file is byte-identical.
// Shape is this repo's own: swarms are contiguous ranges of one discovery-order array
// (the reference threads a linked list through ampinfo[].next); the light swarms grafted onto a
// heavy one form a chain through the swarm table.
#include "hostdb.h"
#include "nw_host.h"
#include "out.h"

#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <omp.h>
#include <parallel/algorithm>
#include <string>
#include <vector>

struct swa_d1_result {
  uint32_t n = 0;
  swa_vec<uint32_t> swarmid, parent, generation;   // (swa_vec: sized without a serial fill, see hostdb.h)
  swa_vec<uint32_t> order;                  // amplicons in discovery order, swarm after swarm
  struct Swarm {                            // plain data: the table of 1.6 M swarms (10 M amplicons) is filled by all threads
    uint64_t mass, sumlen;
    uint32_t seed, size, singletons, maxgen;
    uint32_t begin, end;                    // own members = order[begin, end)
    // light swarms grafted onto this one, in graft order: a chain through the swarm table
    // (head / tail on the heavy swarm, next on each light one; the reference threads a similar
    // list through ampinfo[].next)
    uint32_t graft_head, graft_tail, graft_next;
    uint32_t attached;                      // this (light) swarm hangs on another one
  };
  static constexpr Swarm empty_swarm() {
    return Swarm{0, 0, 0, 0, 0, 0, 0, 0, SWA_NO_AMPLICON, SWA_NO_AMPLICON, SWA_NO_AMPLICON, 0};
  }
  swa_vec<Swarm> swarms;
  swa_vec<uint32_t> graft_cand;             // per amplicon, after swa_d1_graft
  uint64_t swarmcount_adjusted = 0;
  uint32_t largest = 0, maxgen = 0;
  std::string error;
  // The clustering made on the GPU (swa_d1_cluster_resident) brings home only what the swarms file needs — the member
  // order and the swarms' bounds.  swarmid / parent / generation / graft_cand and the per-swarm sums (mass, length,
  // singletons, deepest generation) are fetched and computed when somebody asks: -i, -s, -u, -w, --fastidious and the
  // accessors (need_details).  A plain `swarm -d 1 -o` never does.
  swa_ctx * lazy_ctx = nullptr;
  const swa_hostdb * lazy_db = nullptr;
  bool details = true;
  // swa_d1_result_prepare: order and the download area of the swarms' bounds sized ahead — and pinned, when the runtime agreed
  swa_vec<uint32_t> begin_tmp;
  bool prepared = false, pinned_order = false, pinned_begin = false;
  ~swa_d1_result() {
    if (pinned_order) { swa_host_unpin(order.data()); }
    if (pinned_begin) { swa_host_unpin(begin_tmp.data()); }
  }
};

namespace {

// v[i] = value for all i, by all threads (first touch of a freshly sized array included)
template <class V, class T>
void fill_parallel(V & v, size_t n, T value) {
  v.resize(n);
  auto * p = v.data();
#pragma omp parallel for schedule(static) if (n >= 100000) num_threads(swa_host_team())
  for (int64_t i = 0; i < (int64_t)n; ++i) { p[i] = value; }
}

template <typename F>
void for_each_member(const swa_d1_result * r, const swa_d1_result::Swarm & s, F && f) {
  for (uint32_t k = s.begin; k < s.end; ++k) { f(r->order[k]); }
  for (uint32_t g = s.graft_head; g != SWA_NO_AMPLICON; g = r->swarms[g].graft_next) {
    const auto & l = r->swarms[g];
    for (uint32_t k = l.begin; k < l.end; ++k) { f(r->order[k]); }
  }
}

// what a swarm costs a writer that walks its members: its own and those of the light swarms grafted onto it (which print
// nothing themselves) — swa_format_in_weighted_pieces
inline uint64_t printed_members(const swa_d1_result * r, size_t k) {
  const auto & s = r->swarms[k];
  return s.attached != 0 ? 0u : 1u + std::max<uint32_t>(s.size, s.end - s.begin);
}

constexpr uint32_t kParallelOutputFrom = 200000;   // amplicons from which the writers format on several threads
constexpr uint32_t kParallelUclustFrom = 200;      // -u aligns every member against its seed: worth it much earlier

// number of every swarm in the output files (attached = grafted swarms have none: they are printed
// with the swarm they were grafted to)
std::vector<uint32_t> output_numbers(const swa_d1_result * r) {
  std::vector<uint32_t> number(r->swarms.size());
  uint32_t next = 0;
  for (size_t k = 0; k < r->swarms.size(); ++k) { number[k] = next; if (r->swarms[k].attached == 0) { ++next; } }
  return number;
}

// swarmid / parent / generation from HBM and the sums of every swarm (see swa_d1_result); false when the context that holds
// them has moved on
bool need_details(const swa_d1_result * cr) {
  if (cr->details) { return true; }
  if (cr->lazy_ctx == nullptr) { return false; }
  auto * r = const_cast<swa_d1_result *>(cr);
  const uint32_t n = r->n;
  const swa_hostdb * db = r->lazy_db;
  r->swarmid.resize(n); r->parent.resize(n); r->generation.resize(n);
  // (the fetch faults the arrays' pages in on the library's worker threads before the copies land: swa_touch_pages)
  fill_parallel(r->graft_cand, n, (uint32_t)SWA_NO_AMPLICON);
  if (swa_d1_cluster_fetch(r->lazy_ctx, r->swarmid.data(), r->generation.data(), r->parent.data()) != SWA_OK) {
    r->error = swa_last_error(r->lazy_ctx);
    return false;
  }
  const auto & gen = r->generation;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(swa_host_team())
  for (int64_t s = 0; s < (int64_t)r->swarms.size(); ++s) {
    auto & sw = r->swarms[(size_t)s];
    for (uint32_t k = sw.begin; k < sw.end; ++k) {
      // (members of consecutive swarms are consecutive in `order`: what the sums read 16 members ahead is asked for now)
      if (k + 16u < n) { const uint32_t f = r->order[k + 16u]; __builtin_prefetch(&db->abundance[f]); __builtin_prefetch(&db->seqlen[f]); __builtin_prefetch(&gen[f]); }
      const uint32_t a = r->order[k];
      sw.mass += db->abundance[a];
      sw.sumlen += db->seqlen[a];
      if (db->abundance[a] == 1) { ++sw.singletons; }
      sw.maxgen = std::max(sw.maxgen, gen[a]);
    }
  }
  r->details = true;
  r->lazy_ctx = nullptr;                                    // (nothing of the context is needed from here on)
  return true;
}

// what an entry point returns when the details could not be had: the fetch's own failure (SWA_E_DEVICE, text in
// swa_d1_result_error) — never SWA_E_ARG, which the command line reports as "Unable to open ... file" 
int details_failed(const swa_d1_result * r) { return r->error.empty() ? SWA_E_ARG : SWA_E_DEVICE; }

}  // namespace

// ---- the same clustering, computed without the serial walk --------------------------------
// The greedy loop above is equivalent to three order-free statements (i -> j = "j is in i's
// neighbour list"; ids are db order):
//   swarm(v)      = seed(v) = the smallest id among v and everything that reaches v.
//                   (That id m is a seed: anything claiming m earlier would reach v and be smaller.
//                   When m's turn comes nothing on a path m -> ... -> v is taken, for the same
//                   reason, so m's walk claims v.)
//   generation(v) = the distance from seed(v) to v (every path from the seed to v stays inside
//                   the swarm, again because a node taken earlier would make v's seed smaller).
//   parent(v)     = the smallest id u with u -> v and generation(u) = generation(v) - 1 (a
//                   generation is expanded in ascending id order and the first to reach v keeps it).
// Members are listed by (generation, id), swarms by seed id.  Each statement is a data-parallel
// fixed point / sweep over the CSR, which is what the many cores of a GPU host are for: used for
// large inputs, and checked against the serial walk in tests/test_host_logic.py.
namespace {

inline void atomic_min_u32(uint32_t * addr, uint32_t value) {
  uint32_t seen = __atomic_load_n(addr, __ATOMIC_RELAXED);
  while (value < seen && !__atomic_compare_exchange_n(addr, &seen, value, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

void cluster_by_fixed_points(const swa_hostdb * db, const uint64_t * offsets, const uint32_t * neighbours,
                             swa_d1_result * r) {
  const uint32_t n = db->n;
  const int64_t n64 = (int64_t)n;
  constexpr uint32_t kUnset = SWA_NO_AMPLICON;
  const bool timing = std::getenv("SWARM_AMD_CLUSTER_TIMING") != nullptr;
  double t_last = omp_get_wtime();
  auto lap = [&](const char * what) {
    if (!timing) { return; }
    const double now = omp_get_wtime();
    std::fprintf(stderr, "[cluster] %-24s %8.3f ms\n", what, 1000.0 * (now - t_last));
    t_last = now;
  };
  // 1. seed(v): push the smaller label along every edge until nothing changes
  swa_vec<uint32_t> label(n);
  swa_vec<uint8_t> active, next_active;
  fill_parallel(active, n, (uint8_t)1);
  fill_parallel(next_active, n, (uint8_t)0);
#pragma omp parallel for schedule(static) num_threads(swa_host_team())
  for (int64_t v = 0; v < n64; ++v) { label[(size_t)v] = (uint32_t)v; }
  for (bool changed = true; changed;) {
    changed = false;
#pragma omp parallel for schedule(dynamic, 8192) reduction(|| : changed) num_threads(swa_host_team())
    for (int64_t u = 0; u < n64; ++u) {
      if (active[(size_t)u] == 0) { continue; }
      active[(size_t)u] = 0;
      const uint32_t lu = __atomic_load_n(&label[(size_t)u], __ATOMIC_RELAXED);
      for (uint64_t e = offsets[u]; e < offsets[u + 1]; ++e) {
        const uint32_t v = neighbours[e];
        if (lu < __atomic_load_n(&label[v], __ATOMIC_RELAXED)) {
          atomic_min_u32(&label[v], lu);
          __atomic_store_n(&next_active[v], (uint8_t)1, __ATOMIC_RELAXED);
          changed = true;
        }
      }
    }
    active.swap(next_active);              // (next_active is all zero again: every visited flag was cleared)
  }
  lap("labels");
  // 2. generation(v): level-synchronous distances from the seeds inside their swarms;
  //    parent(v): smallest id of the previous level that points at v
  auto & gen = r->generation;
  auto & parent = r->parent;
#pragma omp parallel for schedule(static) num_threads(swa_host_team())
  for (int64_t v = 0; v < n64; ++v) { gen[(size_t)v] = label[(size_t)v] == (uint32_t)v ? 0u : kUnset; parent[(size_t)v] = kUnset; }
  // Frontier by frontier (not a sweep over all amplicons per level: swarms can be hundreds of generations
  // deep): the nodes of level - 1 are a list; a node is claimed for `level` by exactly one thread (compare and
  // swap on its generation), which appends it to that thread's piece of the next list; parents are the
  // smallest claiming or co-claiming id, by atomic minimum.
  {
    const int nthreads = std::max(1, swa_host_team());
    std::vector<std::vector<uint32_t>> piece((size_t)nthreads);
    std::vector<uint32_t> frontier, next;
    // level 0 = the seeds
#pragma omp parallel num_threads(swa_host_team())
    {
      auto & mine = piece[(size_t)omp_get_thread_num()];
      mine.clear();
#pragma omp for schedule(static) nowait
      for (int64_t v = 0; v < n64; ++v) { if (label[(size_t)v] == (uint32_t)v) { mine.push_back((uint32_t)v); } }
    }
    auto gather_pieces = [&](std::vector<uint32_t> & into) {
      std::vector<size_t> at((size_t)nthreads + 1, 0);
      for (int t = 0; t < nthreads; ++t) { at[(size_t)t + 1] = at[(size_t)t] + piece[(size_t)t].size(); }
      into.resize(at[(size_t)nthreads]);
#pragma omp parallel for schedule(static, 1) num_threads(swa_host_team())
      for (int t = 0; t < nthreads; ++t) { std::copy(piece[(size_t)t].begin(), piece[(size_t)t].end(), into.begin() + (std::ptrdiff_t)at[(size_t)t]); }
    };
    gather_pieces(frontier);
    for (uint32_t level = 1; !frontier.empty(); ++level) {
      const int64_t fsize = (int64_t)frontier.size();
#pragma omp parallel num_threads(swa_host_team())
      {
        auto & mine = piece[(size_t)omp_get_thread_num()];
        mine.clear();
#pragma omp for schedule(dynamic, 2048) nowait
        for (int64_t k = 0; k < fsize; ++k) {
          const uint32_t u = frontier[(size_t)k];
          const uint32_t lu = label[u];
          for (uint64_t e = offsets[u]; e < offsets[u + 1]; ++e) {
            const uint32_t v = neighbours[e];
            if (label[v] != lu) { continue; }
            uint32_t gv = __atomic_load_n(&gen[v], __ATOMIC_RELAXED);
            if (gv == kUnset) {
              if (__atomic_compare_exchange_n(&gen[v], &gv, level, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { mine.push_back(v); gv = level; }
            }
            if (gv == level) { atomic_min_u32(&parent[v], u); }   // (gv holds the winner's value after a lost exchange)
          }
        }
      }
      gather_pieces(next);
      frontier.swap(next);
    }
  }
  lap("generations + parents");
  // 3. swarms in seed order, members by (generation, id)
  // swarm number of every seed = number of seeds before it: per-block counts, prefix, numbering
  swa_vec<uint32_t> sid_of_seed(n);
  const int blocks = std::max(1, swa_host_team());
  std::vector<uint32_t> seeds_before((size_t)blocks + 1, 0);
#pragma omp parallel for schedule(static, 1) num_threads(swa_host_team())
  for (int b = 0; b < blocks; ++b) {
    uint32_t c = 0;
    for (int64_t v = n64 * b / blocks; v < n64 * (b + 1) / blocks; ++v) { c += label[(size_t)v] == (uint32_t)v ? 1u : 0u; }
    seeds_before[(size_t)b + 1] = c;
  }
  for (int b = 0; b < blocks; ++b) { seeds_before[(size_t)b + 1] += seeds_before[(size_t)b]; }
  const uint32_t nswarms = seeds_before[(size_t)blocks];
#pragma omp parallel for schedule(static, 1) num_threads(swa_host_team())
  for (int b = 0; b < blocks; ++b) {
    uint32_t next = seeds_before[(size_t)b];
    for (int64_t v = n64 * b / blocks; v < n64 * (b + 1) / blocks; ++v) {
      if (label[(size_t)v] == (uint32_t)v) { sid_of_seed[(size_t)v] = next++; }
    }
  }
  fill_parallel(r->swarms, nswarms, swa_d1_result::empty_swarm());
  std::vector<uint32_t> size(nswarms, 0);
#pragma omp parallel for schedule(static) num_threads(swa_host_team())
  for (int64_t v = 0; v < n64; ++v) {
    const uint32_t sid = sid_of_seed[label[(size_t)v]];
    r->swarmid[(size_t)v] = sid;
    __atomic_fetch_add(&size[sid], 1u, __ATOMIC_RELAXED);
  }
  uint32_t at = 0;
  for (uint32_t s = 0; s < nswarms; ++s) {
    auto & sw = r->swarms[s];
    sw.begin = at;
    at += size[s];
    sw.end = sw.begin;                      // filled below (used as the cursor)
  }
  lap("swarm table");
  r->order.resize(n);
#pragma omp parallel for schedule(static) num_threads(swa_host_team())
  for (int64_t v = 0; v < n64; ++v) {
    auto & sw = r->swarms[r->swarmid[(size_t)v]];
    r->order[__atomic_fetch_add(&sw.end, 1u, __ATOMIC_RELAXED)] = (uint32_t)v;
  }
  lap("placement");
  uint32_t largest = 0, maxgen = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(max : largest) reduction(max : maxgen) num_threads(swa_host_team())
  for (int64_t s = 0; s < (int64_t)nswarms; ++s) {
    auto & sw = r->swarms[(size_t)s];
    std::sort(r->order.begin() + sw.begin, r->order.begin() + sw.end, [&](uint32_t x, uint32_t y) {
      return gen[x] != gen[y] ? gen[x] < gen[y] : x < y;
    });
    sw.seed = r->order[sw.begin];
    sw.size = sw.end - sw.begin;
    for (uint32_t k = sw.begin; k < sw.end; ++k) {
      const uint32_t a = r->order[k];
      sw.mass += db->abundance[a];
      sw.sumlen += db->seqlen[a];
      if (db->abundance[a] == 1) { ++sw.singletons; }
      sw.maxgen = std::max(sw.maxgen, gen[a]);
    }
    largest = std::max(largest, sw.size);
    maxgen = std::max(maxgen, sw.maxgen);
  }
  r->largest = largest;
  r->maxgen = maxgen;
  lap("member order + sums");
}

}  // namespace

// ---- the same result from the network that is still in HBM (cluster_gpu.hip) ------------------
// swa_d1_network_resident left the CSR on the device; the three order-free statements above are evaluated
// there and only swarm / generation / parent / member order come back.  The per-swarm sums stay here.
static int cluster_resident(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result ** out, swa_d1_result * prepared = nullptr);

extern "C" int swa_d1_result_prepare(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result ** out) {
  if (ctx == nullptr || db == nullptr || out == nullptr) { return SWA_E_ARG; }
  auto * r = new swa_d1_result();
  *out = r;
  r->n = db->n;
  r->prepared = true;
  if (db->n == 0) { return SWA_OK; }
  r->order.resize(db->n);
  r->begin_tmp.resize((size_t)db->n + 1);
  // (pinning faults the pages in and maps them for the device; when it is refused the downloads are staged copies as before)
  const char * how = std::getenv("SWARM_AMD_PIN_RESULTS");     // (experiments: "touch" = pages faulted in, not pinned)
  if (how != nullptr && std::strcmp(how, "touch") == 0) {
    std::memset(r->order.data(), 0, (size_t)db->n * sizeof(uint32_t));
    return SWA_OK;
  }
  r->pinned_order = swa_host_pin(ctx, r->order.data(), (size_t)db->n * sizeof(uint32_t)) == SWA_OK;
  // (the swarms' bounds are a few hundred kilobytes of an array sized for the worst case: not pinned, not touched — 40 MB of
  // pages this process would otherwise fault in and take apart again)
  if (std::getenv("SWARM_AMD_CLUSTER_TIMING") != nullptr) { std::fprintf(stderr, "[cluster] result arrays pinned: order %d, bounds %d\n", (int)r->pinned_order, (int)r->pinned_begin); }
  return SWA_OK;
}

extern "C" int swa_d1_cluster_resident_prepared(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result * prepared) {
  if (prepared == nullptr || !prepared->prepared || db == nullptr || prepared->n != db->n) { return SWA_E_ARG; }
  swa_d1_result * same = prepared;
  return cluster_resident(ctx, db, &same, prepared);
}

// The result owns everything it returns: swarm / generation / parent are fetched and the sums made before this returns, and
// the context may be destroyed or used for the next database afterwards.
extern "C" int swa_d1_cluster_resident(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result ** out) {
  const int rc = cluster_resident(ctx, db, out);
  if (rc != SWA_OK) { return rc; }
  return swa_d1_result_detach(*out);
}

// The command line's form: the details stay in HBM until an accessor, the grafting or one of -i -s -u -w asks for them (a
// plain `swarm -d 1 -o` never does: 160 MB of downloads and a pass over all members less).  The CALLER keeps `ctx` alive and
// its clustering untouched until swa_d1_result_detach (or the result's release).
extern "C" int swa_d1_cluster_resident_lazy(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result ** out) {
  return cluster_resident(ctx, db, out);
}

extern "C" int swa_d1_result_detach(swa_d1_result * r) {
  if (r == nullptr) { return SWA_E_ARG; }
  if (r->n == 0) { r->lazy_ctx = nullptr; return SWA_OK; }
  return need_details(r) ? SWA_OK : details_failed(r);
}

extern "C" const char * swa_d1_result_error(const swa_d1_result * r) { return r == nullptr ? "" : r->error.c_str(); }

static int cluster_resident(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result ** out, swa_d1_result * prepared) {
  if (ctx == nullptr || db == nullptr || out == nullptr) { return SWA_E_ARG; }
  auto * r = prepared != nullptr ? prepared : new swa_d1_result();
  *out = r;
  const uint32_t n = db->n;
  r->n = n;
  const bool timing = std::getenv("SWARM_AMD_CLUSTER_TIMING") != nullptr;
  double t_last = omp_get_wtime();
  auto lap = [&](const char * what) {
    if (!timing) { return; }
    const double now = omp_get_wtime();
    std::fprintf(stderr, "[cluster] %-24s %8.3f ms\n", what, 1000.0 * (now - t_last));
    t_last = now;
  };
  if (n == 0) { return SWA_OK; }
  if (prepared == nullptr) {
    r->order.resize(n);                                     // (every entry is written by the download below)
    r->begin_tmp.resize((size_t)n + 1);
  }
  swa_vec<uint32_t> & begin = r->begin_tmp;
  lap("result arrays");
  uint32_t nswarms = 0;
  const int rc = swa_d1_cluster_device(ctx, nullptr, nullptr, nullptr, r->order.data(), begin.data(), n, &nswarms);
  if (rc != SWA_OK) { r->error = swa_last_error(ctx); return rc; }
  lap("device + download");
  r->lazy_ctx = ctx;
  r->lazy_db = db;
  r->details = false;
  r->swarms.resize(nswarms);
  uint32_t largest = 0;
#pragma omp parallel for schedule(static) reduction(max : largest) num_threads(swa_host_team())
  for (int64_t s = 0; s < (int64_t)nswarms; ++s) {
    auto & sw = r->swarms[(size_t)s];
    sw = swa_d1_result::empty_swarm();
    sw.begin = begin[(size_t)s];
    sw.end = begin[(size_t)s + 1];
    sw.seed = r->order[sw.begin];
    sw.size = sw.end - sw.begin;
    largest = std::max(largest, sw.size);
  }
  r->largest = largest;
  r->maxgen = swa_d1_cluster_maxgen(ctx);
  r->swarmcount_adjusted = r->swarms.size();
  lap("swarm table");
  return SWA_OK;
}

// ---- clustering (src/algod1.cc:1185-1280, process_seed 673-718) ------------------------
extern "C" int swa_d1_cluster(const swa_hostdb * db, const uint64_t * offsets, const uint32_t * neighbours,
                              swa_d1_result ** out) {
  if (db == nullptr || out == nullptr || (db->n > 0 && (offsets == nullptr))) { return SWA_E_ARG; }
  auto * r = new swa_d1_result();
  *out = r;
  const uint32_t n = db->n;
  r->n = n;
  fill_parallel(r->swarmid, n, (uint32_t)SWA_NO_AMPLICON);
  fill_parallel(r->parent, n, (uint32_t)SWA_NO_AMPLICON);
  fill_parallel(r->generation, n, 0u);
  fill_parallel(r->graft_cand, n, (uint32_t)SWA_NO_AMPLICON);
  // large inputs on a many-core host: the order-free formulation (SWARM_AMD_CLUSTER=serial|parallel
  // overrides the choice)
  {
    const char * mode = std::getenv("SWARM_AMD_CLUSTER");
    const bool force_parallel = mode != nullptr && std::strcmp(mode, "parallel") == 0;
    const bool force_serial = mode != nullptr && std::strcmp(mode, "serial") == 0;
    if (force_parallel || (!force_serial && n >= 500000 && swa_host_team() >= 8)) {
      cluster_by_fixed_points(db, offsets, neighbours, r);
      r->swarmcount_adjusted = r->swarms.size();
      return SWA_OK;
    }
  }
  r->order.reserve(n);
  // The walk is a chain of dependent random reads (row of s -> ids -> swarmid of each id), so the
  // loop touches as little as possible per amplicon: the visited test reads swarmid[] only,
  // parent / generation go to arrays aligned with `order` (sequential writes) and are scattered
  // afterwards in one pass of independent writes, and the rows of the next generation are
  // prefetched while the current one is expanded.
  std::vector<uint32_t> order_parent, order_generation;
  order_parent.reserve(n);
  order_generation.reserve(n);
  std::vector<uint64_t> fresh;                // (id << 32 | parent): sorting it sorts by id
  for (uint32_t seed = 0; seed < n; ++seed) {
    if (r->swarmid[seed] != SWA_NO_AMPLICON) { continue; }
    const uint32_t sid = (uint32_t)r->swarms.size();
    swa_d1_result::Swarm sw = swa_d1_result::empty_swarm();
    sw.seed = seed;
    sw.begin = (uint32_t)r->order.size();
    r->swarmid[seed] = sid;
    r->order.push_back(seed);
    order_parent.push_back(SWA_NO_AMPLICON);
    order_generation.push_back(0);
    uint32_t gen_begin = sw.begin;            // current generation = order[gen_begin, gen_end)
    uint32_t gen_end = gen_begin + 1;
    uint32_t gen = 0;
    while (gen_begin < gen_end) {
      fresh.clear();
      for (uint32_t k = gen_begin; k < gen_end; ++k) {
        const uint32_t s = r->order[k];
        const uint64_t row_end = offsets[s + 1];
        for (uint64_t e = offsets[s]; e < row_end; ++e) {
          if (e + 8 < row_end) { __builtin_prefetch(&r->swarmid[neighbours[e + 8]]); }
          const uint32_t amp = neighbours[e];
          if (r->swarmid[amp] == SWA_NO_AMPLICON) {
            r->swarmid[amp] = sid;
            __builtin_prefetch(&offsets[amp]);
            fresh.push_back(((uint64_t)amp << 32) | s);
          }
        }
      }
      // each generation joins the swarm sorted by db index (= abundance, then header)
      std::sort(fresh.begin(), fresh.end());
      for (const uint64_t f : fresh) {
        r->order.push_back((uint32_t)(f >> 32));
        order_parent.push_back((uint32_t)f);
        order_generation.push_back(gen + 1);
      }
      gen_begin = gen_end;
      gen_end = (uint32_t)r->order.size();
      if (!fresh.empty()) { ++gen; }
    }
    sw.end = (uint32_t)r->order.size();
    sw.size = sw.end - sw.begin;
    sw.maxgen = gen;
    r->largest = std::max(r->largest, sw.size);
    r->maxgen = std::max(r->maxgen, sw.maxgen);
    r->swarms.push_back(sw);
  }
  // per-amplicon results and per-swarm sums: independent gathers / scatters over `order`
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t a = r->order[k];
    r->parent[a] = order_parent[k];
    r->generation[a] = order_generation[k];
  }
  for (auto & sw : r->swarms) {
    for (uint32_t k = sw.begin; k < sw.end; ++k) {
      const uint32_t a = r->order[k];
      sw.mass += db->abundance[a];
      sw.sumlen += db->seqlen[a];
      if (db->abundance[a] == 1) { ++sw.singletons; }
    }
  }
  r->swarmcount_adjusted = r->swarms.size();
  return SWA_OK;
}

extern "C" void swa_d1_result_free(swa_d1_result * r) { delete r; }

extern "C" void swa_d1_result_summary(const swa_d1_result * r, uint64_t * out4) {
  out4[0] = r->swarmcount_adjusted;
  out4[1] = r->largest;
  out4[2] = r->maxgen;
  out4[3] = r->swarms.size();
}

extern "C" const uint32_t * swa_d1_result_swarmid(const swa_d1_result * r) { return need_details(r) ? r->swarmid.data() : nullptr; }
extern "C" const uint32_t * swa_d1_result_parent(const swa_d1_result * r) { return need_details(r) ? r->parent.data() : nullptr; }
extern "C" const uint32_t * swa_d1_result_generation(const swa_d1_result * r) { return need_details(r) ? r->generation.data() : nullptr; }

// ---- fastidious bookkeeping (src/algod1.cc:1291-1328) -----------------------------------
extern "C" int swa_d1_light_flags(const swa_d1_result * r, int64_t boundary, uint8_t * is_light, uint64_t * stats5) {
  if (!need_details(r) && r->n != 0) { return details_failed(r); }      // (without the masses every swarm would look light)
  uint64_t light_swarms = 0, light_amps = 0, light_nt = 0;
  const int64_t nswarms = (int64_t)r->swarms.size();
  // (every amplicon belongs to one swarm: the flags are independent writes; 3 M swarms at 10 M amplicons with 30 % light ones)
#pragma omp parallel for schedule(static) reduction(+ : light_swarms, light_amps, light_nt) if (r->n >= kParallelOutputFrom) num_threads(swa_host_team())
  for (int64_t i = 0; i < nswarms; ++i) {
    const auto & s = r->swarms[(size_t)i];
    const bool light = s.mass < (uint64_t)boundary;
    if (light) { ++light_swarms; light_amps += s.size; light_nt += s.sumlen; }
    for (uint32_t k = s.begin; k < s.end; ++k) { is_light[r->order[k]] = light ? 1 : 0; }
  }
  stats5[0] = light_swarms;
  stats5[1] = light_amps;
  stats5[2] = light_nt;
  stats5[3] = r->swarms.size() - light_swarms;
  stats5[4] = r->n - light_amps;
  return SWA_OK;
}

// ---- grafting (src/algod1.cc:214-241 attach, 274-336 attach_candidates) -----------------
// The attach loop of swa_d1_graft (below: its sequential form, which IS the reference's, src/algod1.cc:214-241, 274-336) over
// millions of sorted (parent, child) pairs, by all threads.  What the sequential walk decides, restated without the walk:
//   * a light swarm hangs on the parent of the FIRST pair, in sorted order, whose child is one of its members (later pairs
//     find it attached: their child's candidate is withdrawn) — a minimum over pair numbers, per light swarm;
//   * a heavy swarm's chain holds the light swarms it won in the order of those winning pairs, and its sums grow by theirs.
// So: the winning pair of every light swarm by an atomic minimum; the winners, in pair order, keyed (heavy swarm, rank among
// the winners) and sorted; every heavy swarm's run of that list linked and summed by one thread.  (3.2 M pairs, 1.5 M grafts at
// 10 M amplicons with 30 % light ones: the sequential walk was 45 of the 61 ms this phase took.)
static uint32_t graft_sorted_pairs_in_parallel(swa_d1_result * r, const uint64_t * pairs, size_t npairs, int blocks) {
  const size_t nswarms = r->swarms.size();
  swa_vec<uint32_t> first_pair;
  fill_parallel(first_pair, nswarms, 0xFFFFFFFFu);                       // (npairs <= n < 2^32 - 1)
  const int64_t np64 = (int64_t)npairs;
#pragma omp parallel for schedule(static) num_threads(swa_host_team())
  for (int64_t at = 0; at < np64; ++at) {
    if (at + 16 < np64) { __builtin_prefetch(&r->swarmid[(uint32_t)pairs[at + 16]]); }
    uint32_t * slot = &first_pair[r->swarmid[(uint32_t)pairs[at]]];
    uint32_t seen = __atomic_load_n(slot, __ATOMIC_RELAXED);
    while ((uint32_t)at < seen && !__atomic_compare_exchange_n(slot, &seen, (uint32_t)at, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
  }
  // the winners in pair order (a block's share goes behind the shares of the blocks before it); the losers' candidates withdrawn
  std::vector<uint64_t> before((size_t)blocks + 1, 0);
#pragma omp parallel for schedule(static, 1) num_threads(swa_host_team())
  for (int b = 0; b < blocks; ++b) {
    uint64_t c = 0;
    for (int64_t at = np64 * b / blocks; at < np64 * (b + 1) / blocks; ++at) {
      const uint32_t child = (uint32_t)pairs[at];
      if (first_pair[r->swarmid[child]] == (uint32_t)at) { ++c; } else { r->graft_cand[child] = SWA_NO_AMPLICON; }
    }
    before[(size_t)b + 1] = c;
  }
  for (int b = 0; b < blocks; ++b) { before[(size_t)b + 1] += before[(size_t)b]; }
  const size_t nwin = before[(size_t)blocks];
  swa_vec<uint64_t> keyed(nwin);                                        // heavy swarm << 32 | rank among the winners
  swa_vec<uint32_t> light_of(nwin);                                     // the light swarm of every winner, by rank
#pragma omp parallel for schedule(static, 1) num_threads(swa_host_team())
  for (int b = 0; b < blocks; ++b) {
    uint64_t rank = before[(size_t)b];
    for (int64_t at = np64 * b / blocks; at < np64 * (b + 1) / blocks; ++at) {
      const uint32_t parent = (uint32_t)(pairs[at] >> 32), child = (uint32_t)pairs[at];
      const uint32_t light_id = r->swarmid[child];
      if (first_pair[light_id] != (uint32_t)at) { continue; }
      keyed[rank] = ((uint64_t)r->swarmid[parent] << 32) | rank;
      light_of[rank] = light_id;
      ++rank;
    }
  }
  __gnu_parallel::sort(keyed.begin(), keyed.end());
  // every heavy swarm's run: linked in order, summed — runs are found from their first entries, one thread per run
  uint32_t largest = r->largest;
  const int64_t nw64 = (int64_t)nwin;
#pragma omp parallel for schedule(dynamic, 4096) reduction(max : largest) num_threads(swa_host_team())
  for (int64_t i = 0; i < nw64; ++i) {
    const uint32_t heavy_id = (uint32_t)(keyed[(size_t)i] >> 32);
    if (i > 0 && (uint32_t)(keyed[(size_t)i - 1] >> 32) == heavy_id) { continue; }      // (not the first of its run)
    auto & heavy = r->swarms[heavy_id];
    for (int64_t j = i; j < nw64 && (uint32_t)(keyed[(size_t)j] >> 32) == heavy_id; ++j) {
      const uint32_t light_id = light_of[(uint32_t)keyed[(size_t)j]];
      auto & light = r->swarms[light_id];
      if (heavy.graft_head == SWA_NO_AMPLICON) { heavy.graft_head = light_id; }
      else { r->swarms[heavy.graft_tail].graft_next = light_id; }
      heavy.graft_tail = light_id;
      heavy.size += light.size;
      heavy.singletons += light.singletons;
      heavy.mass += light.mass;
      heavy.sumlen += light.sumlen;                 // maxgen untouched, like the reference
      light.attached = 1;
    }
    largest = std::max(largest, heavy.size);
  }
  r->largest = largest;
  r->swarmcount_adjusted -= nwin;
  return (uint32_t)nwin;
}

extern "C" uint32_t swa_d1_graft(swa_d1_result * r, const uint32_t * graft_cand) {
  if (!need_details(r)) { return 0; }
  // (parent << 32 | child): sorting the packed pairs is the (parent, child) order of
  // src/algod1.cc:263-271, and a plain integer sort runs on all cores
  const int64_t n64 = (int64_t)r->n;
  const int blocks = std::max(1, swa_host_team());
  std::vector<uint64_t> before((size_t)blocks + 1, 0);        // candidates in the blocks before each block
#pragma omp parallel for schedule(static, 1) num_threads(swa_host_team())
  for (int b = 0; b < blocks; ++b) {
    uint64_t c = 0;
    for (int64_t i = n64 * b / blocks; i < n64 * (b + 1) / blocks; ++i) {
      r->graft_cand[(size_t)i] = graft_cand[i];
      c += graft_cand[i] != SWA_NO_AMPLICON ? 1u : 0u;
    }
    before[(size_t)b + 1] = c;
  }
  for (int b = 0; b < blocks; ++b) { before[(size_t)b + 1] += before[(size_t)b]; }
  swa_vec<uint64_t> pairs(before[(size_t)blocks]);
#pragma omp parallel for schedule(static, 1) num_threads(swa_host_team())
  for (int b = 0; b < blocks; ++b) {
    uint64_t at = before[(size_t)b];
    for (int64_t i = n64 * b / blocks; i < n64 * (b + 1) / blocks; ++i) {
      if (graft_cand[i] != SWA_NO_AMPLICON) { pairs[at++] = ((uint64_t)graft_cand[i] << 32) | (uint64_t)i; }
    }
  }
  __gnu_parallel::sort(pairs.begin(), pairs.end());
  const size_t npairs = pairs.size();
  static const bool serial_env = std::getenv("SWARM_AMD_SERIAL_GRAFT") != nullptr;
  if (npairs >= 50000 && blocks >= 3 && !serial_env) { return graft_sorted_pairs_in_parallel(r, pairs.data(), npairs, blocks); }
  uint32_t grafts = 0;
  // (a chain of dependent random reads — swarm of the child, swarm of the parent, both swarms' records — the ids of the
  // pairs 16 ahead and the records of the pairs 8 ahead are asked for early)
  for (size_t at = 0; at < npairs; ++at) {
    if (at + 16 < npairs) {
      __builtin_prefetch(&r->swarmid[(uint32_t)pairs[at + 16]]);
      __builtin_prefetch(&r->swarmid[(uint32_t)(pairs[at + 16] >> 32)]);
    }
    if (at + 8 < npairs) {
      __builtin_prefetch(&r->swarms[r->swarmid[(uint32_t)pairs[at + 8]]], 1);
      __builtin_prefetch(&r->swarms[r->swarmid[(uint32_t)(pairs[at + 8] >> 32)]], 1);
    }
    const uint64_t packed = pairs[at];
    const uint32_t parent = (uint32_t)(packed >> 32), child = (uint32_t)packed;
    auto & light = r->swarms[r->swarmid[child]];
    if (light.attached != 0) {
      r->graft_cand[child] = SWA_NO_AMPLICON;       // this light swarm already hangs somewhere
      continue;
    }
    auto & heavy = r->swarms[r->swarmid[parent]];
    const uint32_t light_id = r->swarmid[child];
    if (heavy.graft_head == SWA_NO_AMPLICON) { heavy.graft_head = light_id; }
    else { r->swarms[heavy.graft_tail].graft_next = light_id; }
    heavy.graft_tail = light_id;
    heavy.size += light.size;
    heavy.singletons += light.singletons;
    heavy.mass += light.mass;
    heavy.sumlen += light.sumlen;                   // maxgen untouched, like the reference
    light.attached = 1;
    r->largest = std::max(r->largest, heavy.size);
    --r->swarmcount_adjusted;
    ++grafts;
  }
  return grafts;
}

// ---- writers ------------------------------------------------------------------------------
// -o / -r  (src/algod1.cc:790-846)
extern "C" int swa_d1_write_swarms(const swa_d1_result * r, const swa_hostdb * db, const char * path, int mothur,
                                   int usearch, int64_t append_abundance, int64_t differences) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  if (mothur) { o.str("swarm_"); o.u64((uint64_t)differences); o.put('\t'); o.u64(r->swarmcount_adjusted); }
  // A member's identifier is three dependent random reads away — order[k] -> its entry pointer -> the entry -> the
  // header text — and the members of consecutive swarms are consecutive in `order`: what the walk will need 24, 16 and 8
  // members ahead is asked for now (without that: ~100 ms of cache misses at 10 M amplicons on 16 threads, lease r5f).
  const uint32_t * order = r->order.data();
  auto format_range = [&](BufOut & sink, size_t begin, size_t end) {
    // pass 1: the ids this piece prints, in printing order, and where every printed swarm ends.  Own members are a run
    // of `order`; grafted light swarms (--fastidious: millions of one-member swarms hung on the heavy ones) are a chain
    // through the swarm table — the next link's record is asked for while this one's members are copied.
    std::vector<uint32_t> ids, ends;
    if (begin < end) { ids.reserve((size_t)r->swarms[end - 1].end - r->swarms[begin].begin + 64); ends.reserve(end - begin); }
    for (size_t k = begin; k < end; ++k) {
      const auto & s = r->swarms[k];
      if (s.attached != 0) { continue; }
      ids.insert(ids.end(), order + s.begin, order + s.end);
      for (uint32_t g = s.graft_head; g != SWA_NO_AMPLICON;) {
        const auto & l = r->swarms[g];
        if (l.graft_next != SWA_NO_AMPLICON) { __builtin_prefetch(&r->swarms[l.graft_next]); }
        ids.insert(ids.end(), order + l.begin, order + l.end);
        g = l.graft_next;
      }
      ends.push_back((uint32_t)ids.size());
    }
    // pass 2: a member's identifier is three dependent random reads away — its entry pointer, the entry, the header
    // text: what the walk needs 24, 16 and 8 members ahead is asked for now (without that: ~100 ms of cache misses at
    // 10 M amplicons on 16 threads, lease r5f)
    sink.reserve(std::min<size_t>(ids.size() * (db->longest_header + 2u) + 64, (size_t)8 << 20));
    const size_t total = ids.size();
    size_t at = 0;
    for (const uint32_t stop : ends) {
      bool first = true;
      for (; at < stop; ++at) {
        if (at + 24 < total) { __builtin_prefetch(&db->ent[ids[at + 24]]); }
        if (at + 16 < total) { __builtin_prefetch(db->ent[ids[at + 16]]); }
        if (at + 8 < total) { __builtin_prefetch(db->hdr(ids[at + 8])); }
        if (mothur) { sink.put(first ? '\t' : ','); }
        else if (!first) { sink.put(' '); }
        first = false;
        swa_out::id(sink, db, ids[at], usearch != 0, append_abundance);
      }
      if (!mothur) { sink.put('\n'); }
    }
  };
  swa_format_in_weighted_pieces(o, r->swarms.size(), r->n >= kParallelOutputFrom, [&](size_t k) { return printed_members(r, k); }, format_range);
  if (mothur) { o.put('\n'); }
  return SWA_OK;
}

// -s  (src/algod1.cc:1040-1062): maxgen is printed twice for d = 1
extern "C" int swa_d1_write_stats(const swa_d1_result * r, const swa_hostdb * db, const char * path, int usearch) {
  if (!need_details(r)) { return details_failed(r); }
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  for (const auto & s : r->swarms) {
    if (s.attached != 0) { continue; }
    o.u64(s.size); o.put('\t'); o.u64(s.mass); o.put('\t');
    swa_out::id_noabundance(o, db, s.seed, usearch != 0);
    o.put('\t'); o.u64(db->abundance[s.seed]); o.put('\t'); o.u64(s.singletons);
    o.put('\t'); o.u64(s.maxgen); o.put('\t'); o.u64(s.maxgen); o.put('\n');
  }
  return SWA_OK;
}

// -i  (src/algod1.cc:985-1037)
extern "C" int swa_d1_write_structure(const swa_d1_result * r, const swa_hostdb * db, const char * path, int usearch) {
  if (!need_details(r)) { return details_failed(r); }
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  const std::vector<uint32_t> number = output_numbers(r);
  swa_format_in_weighted_pieces(o, r->swarms.size(), r->n >= kParallelOutputFrom, [&](size_t k) { return printed_members(r, k); }, [&](BufOut & sink, size_t begin, size_t end) {
    for (size_t k = begin; k < end; ++k) {
      const auto & s = r->swarms[k];
      if (s.attached != 0) { continue; }
      const uint32_t cluster_no = number[k];
      for_each_member(r, s, [&](uint32_t a) {
        if (a == s.seed) { return; }
        const uint32_t gp = r->graft_cand[a];
        if (gp != SWA_NO_AMPLICON) {
          swa_out::id_noabundance(sink, db, gp, usearch != 0);
          sink.put('\t');
          swa_out::id_noabundance(sink, db, a, usearch != 0);
          sink.put('\t'); sink.u64(2); sink.put('\t'); sink.u64(cluster_no + 1); sink.put('\t'); sink.u64(r->generation[gp] + 1);
          sink.put('\n');
        }
        const uint32_t p = r->parent[a];
        if (p != SWA_NO_AMPLICON) {
          swa_out::id_noabundance(sink, db, p, usearch != 0);
          sink.put('\t');
          swa_out::id_noabundance(sink, db, a, usearch != 0);
          sink.put('\t'); sink.u64(1); sink.put('\t'); sink.u64(cluster_no + 1); sink.put('\t'); sink.u64(r->generation[a]);
          sink.put('\n');
        }
      });
    }
  });
  return SWA_OK;
}

// -w  (src/algod1.cc:935-982 + db_fprintseq src/db.cc:925-943)
extern "C" int swa_d1_write_seeds(const swa_d1_result * r, const swa_hostdb * db, const char * path, int usearch) {
  if (!need_details(r)) { return details_failed(r); }
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  std::vector<uint32_t> idx(r->swarms.size());
  std::iota(idx.begin(), idx.end(), 0u);
  std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
    const auto & a = r->swarms[x];
    const auto & b = r->swarms[y];
    if (a.mass != b.mass) { return a.mass > b.mass; }
    return std::strcmp(swa_out::hdr(db, a.seed), swa_out::hdr(db, b.seed)) < 0;
  });
  swa_format_in_pieces(o, idx.size(), r->n >= kParallelOutputFrom, [&](BufOut & sink, size_t begin, size_t end) {
    std::string line;
    for (size_t i = begin; i < end; ++i) {
      const auto & s = r->swarms[idx[i]];
      if (s.attached != 0) { continue; }
      sink.put('>');
      swa_out::id_new_abundance(sink, db, s.seed, s.mass, usearch != 0);
      sink.put('\n');
      swa_out::sequence(sink, db, s.seed, line);
    }
  });
  return SWA_OK;
}

// -j  (src/algod1.cc:755-788): rows ascending by target id (the C ABI already sorts them)
extern "C" int swa_d1_write_network(const swa_hostdb * db, const uint64_t * offsets, const uint32_t * neighbours,
                                    const char * path, int usearch, int64_t append_abundance) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  std::vector<uint32_t> row;
  for (uint32_t i = 0; i < db->n; ++i) {
    row.assign(neighbours + offsets[i], neighbours + offsets[i + 1]);
    std::sort(row.begin(), row.end());
    for (uint32_t j : row) {
      swa_out::id(o, db, i, usearch != 0, append_abundance);
      o.put('\t');
      swa_out::id(o, db, j, usearch != 0, append_abundance);
      o.put('\n');
    }
  }
  return SWA_OK;
}

// -u  (src/algod1.cc:849-932): members in swarm order, each aligned against the seed
extern "C" int swa_d1_write_uclust(const swa_d1_result * r, const swa_hostdb * db, const char * path, int usearch,
                                   int64_t append_abundance, uint64_t mismatch, uint64_t gapopen, uint64_t gapextend) {
  if (!need_details(r)) { return details_failed(r); }
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  const std::vector<uint32_t> number = output_numbers(r);
  // one alignment per member: by far the most expensive writer, and every swarm is independent
  swa_format_in_weighted_pieces(o, r->swarms.size(), r->n >= kParallelUclustFrom, [&](size_t k) { return printed_members(r, k); }, [&](BufOut & sink, size_t begin, size_t end) {
    swa_nw_scratch scratch;
    for (size_t k = begin; k < end; ++k) {
      const auto & s = r->swarms[k];
      if (s.attached != 0) { continue; }
      const uint32_t cluster_no = number[k];
      sink.str("C\t"); sink.u64(cluster_no); sink.put('\t'); sink.u64(s.size); sink.str("\t*\t*\t*\t*\t*\t");
      swa_out::id(sink, db, s.seed, usearch != 0, append_abundance);
      sink.str("\t*\n");
      sink.str("S\t"); sink.u64(cluster_no); sink.put('\t'); sink.u64(db->seqlen[s.seed]); sink.str("\t*\t*\t*\t*\t*\t");
      swa_out::id(sink, db, s.seed, usearch != 0, append_abundance);
      sink.str("\t*\n");
      for_each_member(r, s, [&](uint32_t a) {
        if (a == s.seed) { return; }
        const uint64_t nwdiff = swa_nw_align(db->words(a), db->seqlen[a],
                                             db->words(s.seed), db->seqlen[s.seed], mismatch, gapopen,
                                             gapextend, scratch);
        const double columns = (double)scratch.ops.size();
        const double percentid = 100.0 * (columns - (double)nwdiff) / columns;
        sink.str("H\t"); sink.u64(cluster_no); sink.put('\t'); sink.u64(db->seqlen[a]); sink.put('\t'); sink.fixed1(percentid);
        sink.str("\t+\t0\t0\t");
        if (nwdiff > 0) { const std::string cigar = swa_cigar(scratch.ops); sink.write(cigar.data(), cigar.size()); }
        else { sink.put('='); }
        sink.put('\t');
        swa_out::id(sink, db, a, usearch != 0, append_abundance);
        sink.put('\t');
        swa_out::id(sink, db, s.seed, usearch != 0, append_abundance);
        sink.put('\n');
      });
    }
  });
  return SWA_OK;
}
