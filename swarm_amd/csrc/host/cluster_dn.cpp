// cluster_dn.cpp — host side of the d >= 2 path: the greedy agglomeration of algo_run
// (src/algo.cc:384-676) driven by the GPU's fused scan step (swa_scan_step), plus writers.
//
// What stays on the host is exactly the sequential, order-defining part: which amplicon seeds
// next, where an accepted hit is queued (generation, then id: find_correct_position_in_list,
// src/algo.cc:205-219), and the per-swarm statistics.  The reference keeps one pool array and
// rotates accepted hits into a "swarmed" prefix; the unswarmed remainder always stays in
// ascending id order, so here the pool is just a per-amplicon flag on the GPU and each swarm
// is a small queue on the host.  Outputs are byte-identical (-o -r -s -i -w -u).
#include "hostdb.h"
#include "nw_host.h"
#include "out.h"

#include <algorithm>
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct swa_dn_result {
  struct Member { uint32_t id, generation, radius; };
  struct Swarm {
    uint32_t begin = 0, end = 0;          // members = order[begin, end), seed first
    uint64_t mass = 0;
    uint32_t singletons = 0, maxgen = 1, maxradius = 0;
    uint32_t link_begin = 0, link_end = 0;   // this swarm's accepted pairs inside links[]
  };
  struct Link { uint32_t parent, child, diff, swarm, generation; };
  // (swa_vec: sized without a serial fill — the device route writes every entry from all threads, hostdb.h)
  swa_vec<Member> order;
  std::vector<Swarm> swarms;
  swa_vec<Link> links;                    // the -i lines, in acceptance order
  uint64_t largest = 0, maxgenerations = 0;
  int64_t differences = 0;
  uint64_t pen_mismatch = 18, pen_gapopen = 24, pen_gapextend = 13;
  std::string error;
};

// The greedy agglomeration of algo_run (src/algo.cc:384-602) over the complete graph of pairs within d differences
// (swa_dn_graph: row q = the targets t with diff(q, t) <= d that the abundance rule lets q take, ascending, with
// their diffs).  Same order-defining rules as the loop in swa_dn_cluster: seeds by lowest unswarmed id, a sub-seed's
// hits in pool (= id) order, queue kept sorted by generation then id (src/algo.cc:205-219).
// (multi != nullptr: the graph comes from several GPUs, swa_multi_dn_graph)
static int cluster_over_graph(swa_ctx * ctx, swa_multi * multi, const swa_hostdb * db, int no_cluster_breaking, swa_dn_result * r) {
  const uint32_t n = db->n;
  std::vector<uint64_t> off((size_t)n + 1);
  std::vector<uint32_t> nb;
  std::vector<uint8_t> df;
  uint64_t total = 0;
  auto graph = [&](uint32_t * nbp, uint8_t * dfp, uint64_t cap) {
    return multi != nullptr ? swa_multi_dn_graph(multi, no_cluster_breaking, off.data(), nbp, dfp, cap, &total)
                            : swa_dn_graph(ctx, no_cluster_breaking, off.data(), nbp, dfp, cap, &total);
  };
  int rc = graph(nullptr, nullptr, 0);
  if (rc == SWA_E_CAPACITY) {
    nb.resize(total); df.resize(total);
    rc = graph(nb.data(), df.data(), total);
  }
  if (rc != SWA_OK) { r->error = multi != nullptr ? swa_multi_last_error(multi) : swa_last_error(ctx); return rc; }
  std::vector<uint8_t> swarmed(n, 0);
  std::vector<swa_dn_result::Member> queue;
  r->order.reserve(n);
  r->links.reserve(n);
  for (uint32_t seed = 0; seed < n; ++seed) {
    if (swarmed[seed]) { continue; }
    const uint32_t swarm_no = (uint32_t)r->swarms.size() + 1;
    swa_dn_result::Swarm sw;
    sw.link_begin = (uint32_t)r->links.size();
    swarmed[seed] = 1;
    queue.clear();
    queue.push_back({seed, 0, 0});
    size_t next = 0;
    while (next < queue.size()) {
      const swa_dn_result::Member sub = queue[next];
      ++next;
      for (uint64_t e = off[sub.id]; e < off[sub.id + 1]; ++e) {
        const uint32_t id = nb[e];
        if (swarmed[id]) { continue; }
        swarmed[id] = 1;
        const uint32_t diff = df[e];
        size_t pos = queue.size();
        while (pos > next && queue[pos - 1].id > id && queue[pos - 1].generation > sub.generation) { --pos; }
        const swa_dn_result::Member m{id, sub.generation + 1, sub.radius + diff};
        queue.insert(queue.begin() + (std::ptrdiff_t)pos, m);
        sw.maxgen = std::max(sw.maxgen, m.generation);
        sw.maxradius = std::max(sw.maxradius, m.radius);
        r->links.push_back({sub.id, id, diff, swarm_no, m.generation});
      }
    }
    sw.link_end = (uint32_t)r->links.size();
    sw.begin = (uint32_t)r->order.size();
    for (const auto & m : queue) {
      r->order.push_back(m);
      sw.mass += db->abundance[m.id];
      if (db->abundance[m.id] == 1) { ++sw.singletons; }
    }
    sw.end = (uint32_t)r->order.size();
    r->largest = std::max<uint64_t>(r->largest, sw.end - sw.begin);
    r->maxgenerations = std::max<uint64_t>(r->maxgenerations, sw.maxgen);
    r->swarms.push_back(sw);
  }
  return SWA_OK;
}

// The same with the walk itself on the GPU: the graph stays in HBM (swa_dn_graph_resident), swa_d1_cluster_device labels
// swarms, generations and parents on it (the greedy loop above is a pure function of the directed graph: a swarm = what
// its seed reaches, a member's generation = its distance from the seed, its parent = the smallest id of the previous
// generation that points at it — cluster_gpu.hip), and only five arrays of n entries come back instead of the graph.
// Radii and the -i lines (acceptance order: sub-seeds in queue order, each one's hits by id) follow from the parents.
// SWARM_AMD_TIMING=1: milliseconds since the first stamp of the clustering phase at every milestone, on stderr
static void dn_stamp(const char * what) {
  static const bool on = std::getenv("SWARM_AMD_TIMING") != nullptr;
  static const auto t0 = std::chrono::steady_clock::now();
  if (on) { std::fprintf(stderr, "[dn %8.3f ms] %s\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what); }
}

static int cluster_on_device(swa_ctx * ctx, const swa_hostdb * db, int no_cluster_breaking, swa_dn_result * r) {
  const uint32_t n = db->n;
  uint64_t total = 0;
  dn_stamp("graph: start");
  int rc = swa_dn_graph_resident(ctx, no_cluster_breaking, &total);
  if (rc != SWA_OK) { r->error = swa_last_error(ctx); return rc; }
  dn_stamp("graph: resident");
  // (no serial zero fill of arrays the downloads overwrite: 40 MB of them at 1 M amplicons were a third of the 11 ms the
  // tables below took, lease r6j; their pages faulted in by the team instead of the copying thread)
  swa_vec<uint32_t> swarmid(n), generation(n), parent(n), order(n), begins((size_t)n + 1);
  swa_vec<uint8_t> pdiff(n);
  {
    char * const arrays[] = {reinterpret_cast<char *>(swarmid.data()), reinterpret_cast<char *>(generation.data()), reinterpret_cast<char *>(parent.data()),
                             reinterpret_cast<char *>(order.data())};
    const int64_t pages = (int64_t)(((size_t)n * sizeof(uint32_t) + 4095) / 4096);
#pragma omp parallel for schedule(static) num_threads(swa_host_team()) if (pages >= 256)
    for (int64_t k = 0; k < pages; ++k) { for (char * a : arrays) { a[k * 4096] = 0; } }
  }
  uint32_t nswarms = 0;
  rc = swa_d1_cluster_device(ctx, swarmid.data(), generation.data(), parent.data(), order.data(), begins.data(), n, &nswarms);
  dn_stamp("walk on the device: labels, generations, parents, order");
  if (rc == SWA_OK) { rc = swa_dn_parent_diffs(ctx, pdiff.data()); }
  if (rc != SWA_OK) { r->error = swa_last_error(ctx); return rc; }
  dn_stamp("parent differences");
  swa_vec<uint32_t> radius(n), pos_of(n);                   // (both written for a member before any of its children reads them)
  r->order.resize(n);
  r->swarms.resize(nswarms);
  r->links.resize((size_t)n - nswarms);
  // Swarm by swarm, independently: swarm s owns order[begins[s], begins[s + 1]) and — one link per member but the seed —
  // links[begins[s] - s, begins[s + 1] - s - 1).  (31 ms of a 91 ms clustering phase at 1 M x 400 as one thread's loop;
  // the sums r->largest / maxgenerations are folded behind it.)
  uint64_t largest = 0, maxgenerations = 0;
  bool broken = false;
#pragma omp parallel for schedule(dynamic, 64) reduction(max : largest, maxgenerations) reduction(|| : broken) num_threads(swa_host_team())
  for (uint32_t s = 0; s < nswarms; ++s) {
    std::vector<uint32_t> fill;
    swa_dn_result::Swarm & sw = r->swarms[s];
    sw.begin = begins[s]; sw.end = begins[s + 1];
    uint32_t link_at = begins[s] - s;
    sw.link_begin = link_at;
    const uint32_t size = sw.end - sw.begin;
    for (uint32_t at = sw.begin; at < sw.end; ++at) {       // (generation, id) order: a parent comes before its children
      const uint32_t v = order[at];
      const uint32_t g = generation[v];
      if (g != 0 && pdiff[v] == 0xFF) { broken = true; }
      const uint32_t rad = g == 0 ? 0u : radius[parent[v]] + pdiff[v];
      radius[v] = rad;
      pos_of[v] = at;
      r->order[at] = {v, g, rad};
      sw.mass += db->abundance[v];
      if (db->abundance[v] == 1) { ++sw.singletons; }
      sw.maxgen = std::max(sw.maxgen, g);
      sw.maxradius = std::max(sw.maxradius, rad);
    }
    if (size > 1) {
      // the links by the parent's place in the queue (counting sort, stable: a parent's children stay in id order)
      fill.assign((size_t)size + 1, 0);
      for (uint32_t at = sw.begin + 1; at < sw.end; ++at) { ++fill[pos_of[parent[order[at]]] - sw.begin + 1]; }
      for (uint32_t k = 0; k < size; ++k) { fill[k + 1] += fill[k]; }
      for (uint32_t at = sw.begin + 1; at < sw.end; ++at) {
        const uint32_t v = order[at], p = parent[v];
        r->links[link_at + fill[pos_of[p] - sw.begin]++] = {p, v, pdiff[v], s + 1, generation[v]};
      }
      link_at += size - 1;
    }
    sw.link_end = link_at;
    largest = std::max<uint64_t>(largest, size);
    maxgenerations = std::max<uint64_t>(maxgenerations, sw.maxgen);
  }
  if (broken) { r->error = "d >= 2 clustering on the GPU: a parent without a link to its child"; return SWA_E_DEVICE; }
  r->largest = std::max<uint64_t>(r->largest, largest);
  r->maxgenerations = std::max<uint64_t>(r->maxgenerations, maxgenerations);
  dn_stamp("swarm tables on the host");
  return SWA_OK;
}

// d >= 2 on several GPUs (SWARM_AMD_DEVICES): the bulk graph shared out by ownership of window groups; a database the
// graph route cannot serve (a sequence too short for d + 1 windows) is clustered by rank 0 alone with the fused scan.
extern "C" int swa_dn_cluster_multi(swa_multi * multi, const swa_hostdb * db, int64_t differences, int no_cluster_breaking,
                                    uint64_t mismatch, uint64_t gapopen, uint64_t gapextend, swa_dn_result ** out) {
  if (multi == nullptr || db == nullptr || out == nullptr || differences < 2) { return SWA_E_ARG; }
  const char * route = std::getenv("SWARM_AMD_DN");
  const bool want_scan = route != nullptr && std::strcmp(route, "scan") == 0;
  if (db->n != 0 && !want_scan) {
    int rc = swa_multi_dn_begin(multi, mismatch, gapopen, gapextend, (uint64_t)differences);
    if (rc != SWA_OK) {                                       // (no silent single-rank run behind a failed start: ADVICE r03)
      auto * r = new swa_dn_result();
      *out = r;
      r->differences = differences;
      r->error = swa_multi_last_error(multi);
      return rc;
    }
    if (swa_multi_dn_graph_supported(multi) != 0) {
      auto * r = new swa_dn_result();
      *out = r;
      r->differences = differences;
      r->pen_mismatch = mismatch; r->pen_gapopen = gapopen; r->pen_gapextend = gapextend;
      return cluster_over_graph(swa_multi_ctx(multi, 0), multi, db, no_cluster_breaking, r);
    }
  }
  return swa_dn_cluster(swa_multi_ctx(multi, 0), db, differences, no_cluster_breaking, mismatch, gapopen, gapextend, out);
}

extern "C" int swa_dn_cluster(swa_ctx * ctx, const swa_hostdb * db, int64_t differences, int no_cluster_breaking,
                              uint64_t mismatch, uint64_t gapopen, uint64_t gapextend, swa_dn_result ** out) {
  if (ctx == nullptr || db == nullptr || out == nullptr || differences < 2) { return SWA_E_ARG; }
  auto * r = new swa_dn_result();
  *out = r;
  r->differences = differences;
  r->pen_mismatch = mismatch; r->pen_gapopen = gapopen; r->pen_gapextend = gapextend;
  const uint32_t n = db->n;
  if (n == 0) { return SWA_OK; }
  dn_stamp("clustering: start");
  int rc = swa_qgram_build(ctx);
  dn_stamp("q-gram signatures");
  if (rc == SWA_OK) { rc = swa_search_begin(ctx, mismatch, gapopen, gapextend, (uint64_t)differences); }
  if (rc == SWA_OK) { rc = swa_scan_begin(ctx); }
  dn_stamp("search / scan begin");
  if (rc != SWA_OK) { r->error = swa_last_error(ctx); return rc; }

  // Bulk route (dn_graph.hip): when every sequence has room for d + 1 windows the GPU returns the whole graph of
  // pairs within d differences at once and the greedy loop below runs over that CSR, like d = 1.
  // SWARM_AMD_DN=scan keeps the fused scan (one launch sequence per swarm generation), =graph insists on the graph.
  const char * route = std::getenv("SWARM_AMD_DN");
  const bool want_scan = route != nullptr && std::strcmp(route, "scan") == 0;
  const bool want_graph = route != nullptr && std::strcmp(route, "graph") == 0;
  if (!want_scan && swa_dn_graph_supported(ctx) != 0) {
    // (SWARM_AMD_DN_WALK=host: the graph downloaded and walked on the host — comparison switch)
    const char * walk = std::getenv("SWARM_AMD_DN_WALK");
    // (differences of 255 are what the device walk's parent-difference bytes use for "no link": d >= 255 walks on the host)
    if ((walk != nullptr && std::strcmp(walk, "host") == 0) || differences >= 255) { return cluster_over_graph(ctx, nullptr, db, no_cluster_breaking, r); }
    (void)swa_dn_set_ownership(ctx, 0, 1);                    // (a context that served one rank's share before: the whole graph)
    return cluster_on_device(ctx, db, no_cluster_breaking, r);
  }
  if (want_graph) { r->error = "SWARM_AMD_DN=graph: a sequence is too short for d + 1 windows"; return SWA_E_ARG; }

  std::vector<uint8_t> swarmed(n, 0);
  std::vector<uint32_t> hit_ids(n), hit_diffs(n), hit_sidx(n), batch_ids, batch_radii;
  std::vector<swa_dn_result::Member> queue;
  r->order.reserve(n);
  for (uint32_t seed = 0; seed < n; ++seed) {
    if (swarmed[seed]) { continue; }                       // next initial seed = lowest unswarmed id
    const uint32_t swarm_no = (uint32_t)r->swarms.size() + 1;   // 1-based like the reference
    swa_dn_result::Swarm sw;
    sw.link_begin = (uint32_t)r->links.size();
    swarmed[seed] = 1;
    queue.clear();
    queue.push_back({seed, 0, 0});
    uint32_t nh = 0;
    rc = swa_scan_step(ctx, seed, seed + 1, 1, 0, no_cluster_breaking, hit_ids.data(), hit_diffs.data(), n, &nh);
    if (rc != SWA_OK) { r->error = swa_last_error(ctx); return rc; }
    for (uint32_t k = 0; k < nh; ++k) {                    // first generation, pool (= id) order
      swarmed[hit_ids[k]] = 1;
      queue.push_back({hit_ids[k], 1, hit_diffs[k]});
      sw.maxradius = std::max(sw.maxradius, hit_diffs[k]);
      r->links.push_back({seed, hit_ids[k], hit_diffs[k], swarm_no, 1});
    }
    size_t next = 1;                                       // queue[next..) = swarmed but not yet seeded
    while (next < queue.size()) {
      // All sub-seeds of the current generation go to the GPU in one batch: their hits belong
      // to the next generation and queue up behind them, so the batch is complete when the
      // generation starts.  Hits are computed against the pool as of now; a target found by
      // several sub-seeds is kept for the first one in queue order, exactly what the
      // one-by-one loop of the reference does (src/algo.cc:505-602).
      const uint32_t generation = queue[next].generation;
      size_t batch_end = next;
      while (batch_end < queue.size() && queue[batch_end].generation == generation && batch_end - next < 65535) { ++batch_end; }
      const uint32_t nb = (uint32_t)(batch_end - next);
      batch_ids.resize(nb);
      batch_radii.resize(nb);
      for (uint32_t k = 0; k < nb; ++k) { batch_ids[k] = queue[next + k].id; batch_radii[k] = queue[next + k].radius; }
      rc = swa_scan_batch(ctx, nb, batch_ids.data(), batch_radii.data(), seed + 1, 0, no_cluster_breaking,
                          hit_sidx.data(), hit_ids.data(), hit_diffs.data(), (uint32_t)hit_ids.size(), &nh);
      if (rc == SWA_E_CAPACITY) {
        hit_sidx.resize(nh); hit_ids.resize(nh); hit_diffs.resize(nh);
        rc = swa_scan_fetch(ctx, hit_sidx.data(), hit_ids.data(), hit_diffs.data(), nh);
      }
      if (rc != SWA_OK) { r->error = swa_last_error(ctx); return rc; }
      const size_t batch_first = next;
      uint32_t h = 0;
      for (uint32_t b = 0; b < nb; ++b) {
        const swa_dn_result::Member sub = queue[batch_first + b];   // members of this generation never move
        ++next;
        for (; h < nh && hit_sidx[h] == b; ++h) {
          const uint32_t id = hit_ids[h];
          if (swarmed[id]) { continue; }                   // an earlier sub-seed of the batch took it
          swarmed[id] = 1;
          // keep the unseeded part of the queue ordered by generation, then id (algo.cc:205-219)
          size_t pos = queue.size();
          while (pos > next && queue[pos - 1].id > id && queue[pos - 1].generation > sub.generation) { --pos; }
          const swa_dn_result::Member m{id, sub.generation + 1, sub.radius + hit_diffs[h]};
          queue.insert(queue.begin() + (std::ptrdiff_t)pos, m);
          sw.maxgen = std::max(sw.maxgen, m.generation);
          sw.maxradius = std::max(sw.maxradius, m.radius);
          r->links.push_back({sub.id, id, hit_diffs[h], swarm_no, m.generation});
        }
      }
    }
    sw.link_end = (uint32_t)r->links.size();
    sw.begin = (uint32_t)r->order.size();
    for (const auto & m : queue) {
      r->order.push_back(m);
      sw.mass += db->abundance[m.id];
      if (db->abundance[m.id] == 1) { ++sw.singletons; }
    }
    sw.end = (uint32_t)r->order.size();
    r->largest = std::max<uint64_t>(r->largest, sw.end - sw.begin);
    r->maxgenerations = std::max<uint64_t>(r->maxgenerations, sw.maxgen);
    r->swarms.push_back(sw);
  }
  return SWA_OK;
}

extern "C" void swa_dn_result_free(swa_dn_result * r) { delete r; }
extern "C" const char * swa_dn_result_error(const swa_dn_result * r) { return r != nullptr ? r->error.c_str() : ""; }

// {number of swarms, largest swarm, max generations} — the log's summary (src/algo.cc:699-705)
extern "C" void swa_dn_result_summary(const swa_dn_result * r, uint64_t * out3) {
  out3[0] = r->swarms.size();
  out3[1] = r->largest;
  out3[2] = r->maxgenerations;
}

// -o / -r  (src/algo.cc:258-325)
extern "C" int swa_dn_write_swarms(const swa_dn_result * r, const swa_hostdb * db, const char * path, int mothur,
                                   int usearch, int64_t append_abundance) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  if (!r->order.empty()) {
    if (mothur) { o.str("swarm_"); o.u64((uint64_t)r->differences); o.put('\t'); o.u64(r->swarms.size()); o.put('\t'); }
    // Pieces of equal member counts, formatted by the team, written in order (out.h) — a member's identifier is three
    // dependent random reads away (its entry pointer, the entry, the header text), asked for 24 / 16 / 8 members ahead:
    // 1 M members took one thread 86 ms (lease r6j); the bytes are those of the serial loop.
    const auto * order = r->order.data();
    const size_t total = r->order.size();
    auto format_range = [&](BufOut & sink, size_t begin, size_t end) {
      for (size_t si = begin; si < end; ++si) {
        const auto & s = r->swarms[si];
        if (si != 0) { sink.put(mothur ? '\t' : '\n'); }
        for (uint32_t k = s.begin; k < s.end; ++k) {
          if ((size_t)k + 24 < total) { __builtin_prefetch(&db->ent[order[k + 24].id]); }
          if ((size_t)k + 16 < total) { __builtin_prefetch(db->ent[order[k + 16].id]); }
          if ((size_t)k + 8 < total) { __builtin_prefetch(db->hdr(order[k + 8].id)); }
          if (k != s.begin) { sink.put(mothur ? ',' : ' '); }
          swa_out::id(sink, db, order[k].id, usearch != 0, append_abundance);
        }
      }
    };
    swa_format_in_weighted_pieces(o, r->swarms.size(), total >= 200000, [&](size_t k) { return (uint64_t)(r->swarms[k].end - r->swarms[k].begin); }, format_range);
    o.put('\n');
  }
  return SWA_OK;
}

// -s  (src/algo.cc:662-674)
extern "C" int swa_dn_write_stats(const swa_dn_result * r, const swa_hostdb * db, const char * path, int usearch) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  for (const auto & s : r->swarms) {
    const uint32_t seed = r->order[s.begin].id;
    o.u64(s.end - s.begin); o.put('\t'); o.u64(s.mass); o.put('\t');
    swa_out::id_noabundance(o, db, seed, usearch != 0);
    o.put('\t'); o.u64(db->abundance[seed]); o.put('\t'); o.u64(s.singletons); o.put('\t'); o.u64(s.maxgen);
    o.put('\t'); o.u64(s.maxradius); o.put('\n');
  }
  return SWA_OK;
}

// -i  (src/algo.cc:473-488, 573-589): one line per accepted pair, in acceptance order
extern "C" int swa_dn_write_structure(const swa_dn_result * r, const swa_hostdb * db, const char * path, int usearch) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  for (const auto & l : r->links) {
    swa_out::id_noabundance(o, db, l.parent, usearch != 0);
    o.put('\t');
    swa_out::id_noabundance(o, db, l.child, usearch != 0);
    o.put('\t'); o.u64(l.diff); o.put('\t'); o.u64(l.swarm); o.put('\t'); o.u64(l.generation); o.put('\n');
  }
  return SWA_OK;
}

// -w  (src/algo.cc:122-203).  Seeds sorted by decreasing mass; ties: the reference's
// comparator is `std::strcmp(lhs, rhs) == -1` (not `< 0`), which is libc-dependent and not a
// strict weak order — reproduced as is, on the same libc/libstdc++, for byte-identical output.
extern "C" int swa_dn_write_seeds(const swa_dn_result * r, const swa_hostdb * db, const char * path, int usearch) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  struct Seed { uint64_t mass; uint32_t seed; };
  std::vector<Seed> seeds;
  for (const auto & s : r->swarms) { seeds.push_back({s.mass, r->order[s.begin].id}); }
  std::sort(seeds.begin(), seeds.end(), [&](const Seed & a, const Seed & b) {
    if (a.mass > b.mass) { return true; }
    if (a.mass < b.mass) { return false; }
    return std::strcmp(swa_out::hdr(db, a.seed), swa_out::hdr(db, b.seed)) == -1;
  });
  std::string line;
  for (const auto & s : seeds) {
    o.put('>');
    swa_out::id_new_abundance(o, db, s.seed, s.mass, usearch != 0);
    o.put('\n');
    swa_out::sequence(o, db, s.seed, line);
  }
  return SWA_OK;
}

// -u  (src/algo.cc:608-659): every member, in the order it was accepted (the reference's
// hits[] array), aligned against its swarm's seed with the scalar aligner
extern "C" int swa_dn_write_uclust(const swa_dn_result * r, const swa_hostdb * db, const char * path, int usearch,
                                   int64_t append_abundance) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  // one alignment per member — the most expensive writer by far (the reference's is serial, too),
  // and every swarm is independent: formatted by several threads, written in order
  swa_format_in_weighted_pieces(o, r->swarms.size(), r->order.size() >= 200, [&](size_t k) -> uint64_t { return 1u + (r->swarms[k].end - r->swarms[k].begin); }, [&](BufOut & sink, size_t begin, size_t end) {
    swa_nw_scratch scratch;
    for (size_t cluster_no = begin; cluster_no < end; ++cluster_no) {
      const auto & s = r->swarms[cluster_no];
      const uint32_t seed = r->order[s.begin].id;
      sink.str("C\t"); sink.u64(cluster_no); sink.put('\t'); sink.u64(s.end - s.begin); sink.str("\t*\t*\t*\t*\t*\t");
      swa_out::id(sink, db, seed, usearch != 0, append_abundance);
      sink.str("\t*\n");
      sink.str("S\t"); sink.u64(cluster_no); sink.put('\t'); sink.u64(db->seqlen[seed]); sink.str("\t*\t*\t*\t*\t*\t");
      swa_out::id(sink, db, seed, usearch != 0, append_abundance);
      sink.str("\t*\n");
      for (uint32_t k = s.link_begin; k < s.link_end; ++k) {
        const uint32_t hit = r->links[k].child;
        const uint64_t nwdiff = swa_nw_align(db->words(hit), db->seqlen[hit],
                                             db->words(seed), db->seqlen[seed], r->pen_mismatch,
                                             r->pen_gapopen, r->pen_gapextend, scratch);
        const double columns = (double)scratch.ops.size();
        const double percentid = 100.0 * (columns - (double)nwdiff) / columns;
        sink.str("H\t"); sink.u64(cluster_no); sink.put('\t'); sink.u64(db->seqlen[hit]); sink.put('\t'); sink.fixed1(percentid);
        sink.str("\t+\t0\t0\t");
        if (nwdiff > 0) { const std::string cigar = swa_cigar(scratch.ops); sink.write(cigar.data(), cigar.size()); }
        else { sink.put('='); }
        sink.put('\t');
        swa_out::id(sink, db, hit, usearch != 0, append_abundance);
        sink.put('\t');
        swa_out::id(sink, db, seed, usearch != 0, append_abundance);
        sink.put('\n');
      }
    }
  });
  return SWA_OK;
}
