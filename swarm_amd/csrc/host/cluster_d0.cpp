// cluster_d0.cpp — host side of d = 0 (dereplication): clusters of identical sequences from
// the GPU's "first identical amplicon" array, the reference's ordering, and the writers.
//
// Behaviour mirrors src/derep.cc: members of a cluster in db order, its first member is the
// seed (276-354); clusters by decreasing mass, ties by the seed's db index (67-90); the output
// formats of 107-273, including the constant columns (0 generations / radius in -s, "0 <n> 0"
// in -i, 100.0 % and "=" in -u).  Shape is this repo's own: one members array grouped by
// cluster (counting sort over the seeds) instead of a chain threaded through nextseqtab[].
#include "hostdb.h"
#include "out.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace { constexpr size_t kParallelOutputFrom = 200000; }   // amplicons from which -o / -w are formatted by several threads

struct swa_d0_result {
  struct Cluster {
    uint64_t mass = 0;
    uint32_t seed = 0, size = 0, singletons = 0;
    uint32_t begin = 0;                     // members = members[begin, begin + size), db order, seed first
  };
  std::vector<Cluster> clusters;            // output order
  std::vector<uint32_t> members;
  uint64_t heaviest = 0;
  uint32_t largest = 0;
};

extern "C" int swa_d0_cluster(const swa_hostdb * db, const uint32_t * first_identical, swa_d0_result ** out) {
  if (db == nullptr || out == nullptr || (db->n > 0 && first_identical == nullptr)) { return SWA_E_ARG; }
  auto * r = new swa_d0_result();
  *out = r;
  const uint32_t n = db->n;
  std::vector<uint32_t> cluster_of(n, SWA_NO_AMPLICON);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t f = first_identical[i];
    if (f > i || (f != i && first_identical[f] != f)) { delete r; *out = nullptr; return SWA_E_ARG; }
    if (f == i) {
      cluster_of[i] = (uint32_t)r->clusters.size();
      swa_d0_result::Cluster c;
      c.seed = i;
      r->clusters.push_back(c);
    }
    auto & c = r->clusters[cluster_of[f]];
    ++c.size;
    c.mass += db->abundance[i];
    if (db->abundance[i] == 1) { ++c.singletons; }
  }
  uint32_t at = 0;
  for (auto & c : r->clusters) {
    c.begin = at;
    at += c.size;
    r->heaviest = std::max(r->heaviest, c.mass);
    r->largest = std::max(r->largest, c.size);
  }
  r->members.resize(n);
  std::vector<uint32_t> fill(r->clusters.size(), 0);
  for (uint32_t i = 0; i < n; ++i) {        // ascending i => db order inside every cluster
    const uint32_t c = cluster_of[first_identical[i]];
    r->members[r->clusters[c].begin + fill[c]++] = i;
  }
  std::sort(r->clusters.begin(), r->clusters.end(), [](const swa_d0_result::Cluster & a, const swa_d0_result::Cluster & b) {
    if (a.mass != b.mass) { return a.mass > b.mass; }
    return a.seed < b.seed;
  });
  return SWA_OK;
}

extern "C" void swa_d0_result_free(swa_d0_result * r) { delete r; }

// out3 = {clusters, largest (members), heaviest (mass)}: the log lines of src/derep.cc:405-410
extern "C" void swa_d0_result_summary(const swa_d0_result * r, uint64_t * out3) {
  out3[0] = r->clusters.size();
  out3[1] = r->largest;
  out3[2] = r->heaviest;
}

// -o / -r  (src/derep.cc:207-273)
extern "C" int swa_d0_write_swarms(const swa_d0_result * r, const swa_hostdb * db, const char * path, int mothur,
                                   int usearch, int64_t append_abundance, int64_t differences) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  if (mothur) { o.str("swarm_"); o.u64((uint64_t)differences); o.put('\t'); o.u64(r->clusters.size()); }
  swa_format_in_weighted_pieces(o, r->clusters.size(), r->members.size() >= kParallelOutputFrom, [&](size_t ci) -> uint64_t { return 1u + r->clusters[ci].size; }, [&](BufOut & sink, size_t begin, size_t end) {
    for (size_t ci = begin; ci < end; ++ci) {
      const auto & c = r->clusters[ci];
      for (uint32_t k = 0; k < c.size; ++k) {
        if (mothur) { sink.put(k == 0 ? '\t' : ','); }
        else if (k != 0) { sink.put(' '); }
        swa_out::id(sink, db, r->members[c.begin + k], usearch != 0, append_abundance);
      }
      if (!mothur) { sink.put('\n'); }
    }
  });
  if (mothur) { o.put('\n'); }
  return SWA_OK;
}

// -w  (src/derep.cc:188-204)
extern "C" int swa_d0_write_seeds(const swa_d0_result * r, const swa_hostdb * db, const char * path, int usearch) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  swa_format_in_pieces(o, r->clusters.size(), r->members.size() >= kParallelOutputFrom, [&](BufOut & sink, size_t begin, size_t end) {
    std::string line;
    for (size_t ci = begin; ci < end; ++ci) {
      const auto & c = r->clusters[ci];
      sink.put('>');
      swa_out::id_new_abundance(sink, db, c.seed, c.mass, usearch != 0);
      sink.put('\n');
      swa_out::sequence(sink, db, c.seed, line);
    }
  });
  return SWA_OK;
}

// -s  (src/derep.cc:107-124)
extern "C" int swa_d0_write_stats(const swa_d0_result * r, const swa_hostdb * db, const char * path, int usearch) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  for (const auto & c : r->clusters) {
    o.u64(c.size); o.put('\t'); o.u64(c.mass); o.put('\t');
    swa_out::id_noabundance(o, db, c.seed, usearch != 0);
    o.put('\t'); o.u64(db->abundance[c.seed]); o.put('\t'); o.u64(c.singletons); o.str("\t0\t0\n");
  }
  return SWA_OK;
}

// -i  (src/derep.cc:127-148)
extern "C" int swa_d0_write_structure(const swa_d0_result * r, const swa_hostdb * db, const char * path, int usearch) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  uint64_t cluster_no = 0;
  for (const auto & c : r->clusters) {
    ++cluster_no;
    for (uint32_t k = 1; k < c.size; ++k) {
      swa_out::id_noabundance(o, db, c.seed, usearch != 0);
      o.put('\t');
      swa_out::id_noabundance(o, db, r->members[c.begin + k], usearch != 0);
      o.str("\t0\t"); o.u64(cluster_no); o.str("\t0\n");
    }
  }
  return SWA_OK;
}

// -u  (src/derep.cc:151-185)
extern "C" int swa_d0_write_uclust(const swa_d0_result * r, const swa_hostdb * db, const char * path, int usearch,
                                   int64_t append_abundance) {
  BufOut o(path);
  if (!o.ok()) { return SWA_E_ARG; }
  uint64_t cluster_no = 0;
  for (const auto & c : r->clusters) {
    o.str("C\t"); o.u64(cluster_no); o.put('\t'); o.u64(c.size); o.str("\t*\t*\t*\t*\t*\t");
    swa_out::id(o, db, c.seed, usearch != 0, append_abundance);
    o.str("\t*\n");
    o.str("S\t"); o.u64(cluster_no); o.put('\t'); o.u64(db->seqlen[c.seed]); o.str("\t*\t*\t*\t*\t*\t");
    swa_out::id(o, db, c.seed, usearch != 0, append_abundance);
    o.str("\t*\n");
    for (uint32_t k = 1; k < c.size; ++k) {
      const uint32_t a = r->members[c.begin + k];
      o.str("H\t"); o.u64(cluster_no); o.put('\t'); o.u64(db->seqlen[a]); o.str("\t100.0\t+\t0\t0\t=\t");
      swa_out::id(o, db, a, usearch != 0, append_abundance);
      o.put('\t');
      swa_out::id(o, db, c.seed, usearch != 0, append_abundance);
      o.put('\n');
    }
    ++cluster_no;
  }
  return SWA_OK;
}
