// tools/repro/anchor_race.hip — stand-alone reproducer for the round-1 "wrong suffix-anchor keys"
// failure: the EXACT kernels of d1.hip (this file includes it), launched in
// the order of the dropped table-on-demand build, with the anchor key tables checked word for word
// against keys computed on the host.  No Python, no library: one binary, one stream.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o anchor_race anchor_race.hip \
//         ../../swarm_amd/csrc/ctx.hip ../../swarm_amd/csrc/cluster_gpu.hip ../../swarm_amd/csrc/host_tables.cpp
//   ./anchor_race [n=1000000] [rounds=20]
//
// Flows (each `rounds` times, fresh anchor buffers every round like a fresh context):
//   head      hashes, table + Bloom, duplicate check, ranks, HOST SYNC, anchor indexes   (what ships)
//   lean      hashes, ranks, anchor indexes                (the dropped build: no sync, no table)
//   lean+sync hashes, HOST SYNC, ranks, anchor indexes
//   lean-nomalloc   as lean, but the anchor buffers are allocated once, before the first round
// A mismatch prints which amplicons got a wrong key and what the wrong key corresponds to.
#include "../../swarm_amd/csrc/d1.hip"

#include <chrono>
#include <random>

namespace {

uint64_t h_mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

uint64_t h_anchor_key(const uint64_t * seq, uint32_t len, int which) {
  uint64_t v;
  if (which == 0) { v = seq[0]; }
  else {
    const uint32_t start = len - 32u;
    const uint32_t w = start >> 5, sh = (start & 31u) << 1;
    v = seq[w] >> sh;
    if (sh != 0u) { v |= seq[w + 1] << (64u - sh); }
  }
  return h_mix64(v ^ (which ? 0x9E3779B97F4A7C15ull : 0ull)) & 0x7FFFFFFFFFFFFFFFull;
}

// what a table slot keeps of a key: the high 32 bits of its mix (the low bits choose the slot)
uint64_t h_anchor_tag(const uint64_t * seq, uint32_t len, int which) { return h_mix64(h_anchor_key(seq, len, which)) >> 32; }

#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); exit(2); } } while (0)

struct Db {
  uint32_t n = 0, longest = 0;
  std::vector<uint64_t> seqs, seq_off, abundance;
  std::vector<uint32_t> seqlen;
};

void make_db(Db & db, uint32_t n) {
  std::mt19937_64 rng(12345);
  const uint32_t ncent = n / 50 + 1;
  std::vector<std::vector<uint8_t>> cent(ncent);
  for (auto & c : cent) {
    c.resize(140 + rng() % 19);
    for (auto & b : c) { b = (uint8_t)(rng() & 3u); }
  }
  db.n = n;
  db.seq_off.resize(n + 1);
  db.seqlen.resize(n);
  db.abundance.resize(n);
  uint64_t at = 0;
  for (uint32_t i = 0; i < n; ++i) {
    std::vector<uint8_t> s = cent[rng() % ncent];
    const uint32_t p = (uint32_t)(rng() % s.size());
    switch (rng() % 3) {
      case 0: s[p] = (uint8_t)((s[p] + 1 + rng() % 3) & 3u); break;
      case 1: s.erase(s.begin() + p); break;
      default: s.insert(s.begin() + p, (uint8_t)(rng() & 3u)); break;
    }
    // a second edit keeps most sequences distinct (duplicates are harmless here: nothing is clustered)
    s[(p * 7 + 13) % s.size()] ^= (uint8_t)(1 + rng() % 3);
    const uint32_t len = (uint32_t)s.size();
    const uint32_t nw = (len + 31u) / 32u;
    db.seq_off[i] = at;
    db.seqlen[i] = len;
    db.abundance[i] = (uint64_t)(n - i);                     // descending
    db.longest = std::max(db.longest, len);
    db.seqs.resize(at + nw, 0);
    for (uint32_t q = 0; q < len; ++q) { db.seqs[at + (q >> 5)] |= (uint64_t)s[q] << ((q & 31u) << 1); }
    at += nw;
  }
  db.seq_off[n] = at;
  db.seqs.resize(at + 2, 0);
}

struct Anchor {
  unsigned long long * slots[2] = {};
  uint32_t * counts[2] = {}, * pos_of[2] = {}, * slot_of[2] = {}, * members[2] = {};
  uint4 * minfo[2] = {};
  uint64_t * fp[2] = {};
  uint64_t * offsets[2] = {};
  swa_item * items[2] = {};
  uint32_t * acounters = nullptr;
  uint64_t * scan_tmp = nullptr;
};

void alloc_anchor(Anchor & a, uint32_t n, uint64_t asize, uint32_t tiles) {
  for (int w = 0; w < 2; ++w) {
    CK(hipMalloc(&a.slots[w], asize * 8)); CK(hipMalloc(&a.pos_of[w], (uint64_t)n * 4));
    CK(hipMalloc(&a.minfo[w], (uint64_t)n * 16)); CK(hipMalloc(&a.fp[w], (uint64_t)n * 8));
    CK(hipMalloc(&a.offsets[w], (asize + 1) * 8)); CK(hipMalloc(&a.slot_of[w], (uint64_t)n * 4));
    CK(hipMalloc(&a.members[w], (uint64_t)n * 4)); CK(hipMalloc(&a.items[w], ((uint64_t)n + 128) * sizeof(swa_item)));
  }
  CK(hipMalloc(&a.acounters, 64 * 4));
  CK(hipMalloc(&a.scan_tmp, (uint64_t)tiles * 8));
}

void free_anchor(Anchor & a) {
  for (int w = 0; w < 2; ++w) {
    CK(hipFree(a.slots[w])); CK(hipFree(a.pos_of[w])); CK(hipFree(a.minfo[w])); CK(hipFree(a.fp[w])); CK(hipFree(a.offsets[w]));
    CK(hipFree(a.slot_of[w])); CK(hipFree(a.members[w])); CK(hipFree(a.items[w]));
  }
  CK(hipFree(a.acounters)); CK(hipFree(a.scan_tmp));
  a = Anchor{};
}

}  // namespace

int main(int argc, char ** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 1000000u;
  const int rounds = argc > 2 ? atoi(argv[2]) : 20;
  Db db;
  make_db(db, n);
  hipStream_t stream;
  CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  hipDeviceProp_t prop{};
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  auto grid = [&](uint64_t items) { uint64_t b = (items + 255) / 256; const uint64_t cap = (uint64_t)cus * 8; return (int)std::max<uint64_t>(1, std::min(b, cap)); };

  uint64_t * d_seqs, * d_seq_off, * d_abund, * d_zob, * d_seqhash, * d_bloom, * d_pat;
  uint32_t * d_seqlen, * d_rank, * d_flags, * d_rank_tmp;
  swa_aux * d_aux;
  swa_slot * d_table;
  const uint32_t zlen = db.longest + 2;
  std::vector<uint64_t> zob, pat;
  swa_zobrist_table(zlen, zob);
  swa_bloom_patterns(1024, 8, pat);
  const uint64_t tsize = swa_hashtable_size(n);
  const uint64_t bwords = (tsize < 8 ? 8 : tsize) >> 3;
  CK(hipMalloc(&d_seqs, db.seqs.size() * 8)); CK(hipMalloc(&d_seq_off, ((uint64_t)n + 1) * 8)); CK(hipMalloc(&d_abund, (uint64_t)n * 8));
  CK(hipMalloc(&d_seqlen, (uint64_t)n * 4)); CK(hipMalloc(&d_zob, zob.size() * 8)); CK(hipMalloc(&d_seqhash, (uint64_t)n * 8));
  CK(hipMalloc(&d_aux, (uint64_t)n * sizeof(swa_aux))); CK(hipMalloc(&d_rank, (uint64_t)n * 4)); CK(hipMalloc(&d_rank_tmp, ((uint64_t)n / 256 + 1) * 4)); CK(hipMalloc(&d_flags, 64));
  CK(hipMalloc(&d_table, tsize * sizeof(swa_slot))); CK(hipMalloc(&d_bloom, bwords * 8)); CK(hipMalloc(&d_pat, 1024 * 8));
  CK(hipMemcpyAsync(d_seqs, db.seqs.data(), db.seqs.size() * 8, hipMemcpyHostToDevice, stream));
  CK(hipMemcpyAsync(d_seq_off, db.seq_off.data(), ((uint64_t)n + 1) * 8, hipMemcpyHostToDevice, stream));
  CK(hipMemcpyAsync(d_seqlen, db.seqlen.data(), (uint64_t)n * 4, hipMemcpyHostToDevice, stream));
  CK(hipMemcpyAsync(d_abund, db.abundance.data(), (uint64_t)n * 8, hipMemcpyHostToDevice, stream));
  CK(hipMemcpyAsync(d_zob, zob.data(), zob.size() * 8, hipMemcpyHostToDevice, stream));
  CK(hipMemcpyAsync(d_pat, pat.data(), 1024 * 8, hipMemcpyHostToDevice, stream));
  CK(hipStreamSynchronize(stream));

  uint64_t asize = 64;
  while (asize < 2ull * n) { asize <<= 1; }
  const uint32_t tiles = (uint32_t)((asize + kScanTile - 1) / kScanTile);
  std::vector<uint64_t> want[2];
  for (int w = 0; w < 2; ++w) {
    want[w].resize(n);
    for (uint32_t i = 0; i < n; ++i) { want[w][i] = h_anchor_tag(db.seqs.data() + db.seq_off[i], db.seqlen[i], w); }
  }
  printf("device: %s, %d CUs; n = %u, longest %u, anchor slots %llu\n", prop.name, cus, n, db.longest, (unsigned long long)asize);

  const char * names[4] = {"head", "lean", "lean+sync", "lean-nomalloc"};
  std::vector<unsigned long long> h_keys(asize);
  std::vector<uint32_t> h_slot(n);
  int total_bad = 0;
  for (int flow = 0; flow < 4; ++flow) {
    Anchor a;
    if (flow == 3) { alloc_anchor(a, n, asize, tiles); }
    int bad_rounds = 0;
    for (int r = 0; r < rounds; ++r) {
      CK(hipMemsetAsync(d_flags, 0, 64, stream));
      k_seqhash<true><<<grid(n), 256, 4ull * zlen * 8, stream>>>(d_seqs, d_seq_off, d_seqlen, d_zob, zlen, n, d_seqhash, d_aux, nullptr, nullptr, 32u);
      if (flow == 0) {
        k_table_clear<<<grid(tsize), 256, 0, stream>>>(d_table, tsize, d_bloom, bwords);
        k_table_insert<<<grid(n), 256, 0, stream>>>(d_seqhash, n, nullptr, d_table, tsize - 1, (unsigned long long *)d_bloom, bwords - 1, d_pat);
        k_dup_check<<<grid(n), 256, 0, stream>>>(d_seqs, d_seq_off, d_seqlen, d_seqhash, 0, n, d_table, tsize - 1, d_flags);
      }
      if (flow == 2) { CK(hipStreamSynchronize(stream)); }
      k_abundance_rank<<<grid(n), 256, 0, stream>>>(d_abund, n, d_rank, d_flags, d_rank_tmp);
      k_abundance_rank_carry<<<1, 256, 0, stream>>>(d_rank_tmp, (n + 255u) / 256u);
      k_abundance_rank_fill<<<grid(n), 256, 0, stream>>>(n, d_rank, d_rank_tmp);
      if (flow == 0) {
        uint32_t fl[2];
        CK(hipMemcpyAsync(fl, d_flags, 8, hipMemcpyDeviceToHost, stream));
        CK(hipStreamSynchronize(stream));
      }
      if (flow != 3) { alloc_anchor(a, n, asize, tiles); }     // swa_reserve of a fresh context: while kernels run
      CK(hipMemsetAsync(a.acounters, 0, 64 * 4, stream));
      {
        AnchorBuildArgs b{};
        AnchorScatterArgs sc{};
        b.seqs = d_seqs; b.seq_off = d_seq_off; b.seqlen = d_seqlen; b.n = n; b.first = 0; b.count = n; b.amask = asize - 1; b.probe_limit = asize - 1;
        b.fingerprint = a.fp[0]; b.owner_rank = 0; b.owner_world = 1; b.flags = d_flags; b.minlen = kMinAnchoredLen;
        for (int which = 0; which < 2; ++which) {
          b.slots[which] = a.slots[which]; b.slot_of[which] = a.slot_of[which]; b.pos_of[which] = a.pos_of[which];
          sc.slot_of[which] = a.slot_of[which]; sc.pos_of[which] = a.pos_of[which]; sc.offsets[which] = a.offsets[which];
          sc.minfo[which] = a.minfo[which];
        }
        sc.fingerprint = a.fp[0]; sc.member_fingerprint = a.fp[1]; sc.seqlen = d_seqlen; sc.rank = d_rank; sc.seq_off = d_seq_off; sc.n = n;
        k_anchor_clear<<<grid(asize), 256, 0, stream>>>(a.slots[0], a.slots[1], asize);
        k_anchor_place<true><<<grid(n), 256, 0, stream>>>(b);
        for (int which = 0; which < 2; ++which) {
          k_scan_tiles<unsigned long long><<<tiles, kScanBlock, 0, stream>>>(a.slots[which], (uint32_t)asize, a.scan_tmp);
          k_scan_sums<<<1, kScanBlock, 0, stream>>>(a.scan_tmp, tiles);
          k_scan_apply<unsigned long long><<<tiles, kScanBlock, 0, stream>>>(a.slots[which], (uint32_t)asize, a.scan_tmp, a.offsets[which]);
        }
        k_anchor_scatter<<<grid(n), 256, 0, stream>>>(sc);
        for (int which = 0; which < 2; ++which) {
          k_anchor_items<<<grid(asize), 256, 0, stream>>>(a.slots[which], a.offsets[which], asize, a.items[which], a.acounters + which,
                                                         a.items[which] + (n / 2 + 64), a.acounters + 3 + which, 64u);
        }
      }
      CK(hipGetLastError());
      CK(hipStreamSynchronize(stream));
      int bad_this = 0;
      for (int which = 0; which < 2; ++which) {
        CK(hipMemcpy(h_keys.data(), a.slots[which], asize * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_slot.data(), a.slot_of[which], (uint64_t)n * 4, hipMemcpyDeviceToHost));
        uint32_t wrong = 0, shown = 0, first_bad = 0, last_bad = 0;
        for (uint32_t i = 0; i < n; ++i) {
          const uint32_t s = h_slot[i];
          const uint64_t got = s < asize ? h_keys[s] >> 32 : ~0ull;      // the slot's tag
          if (got == want[which][i]) { continue; }
          if (wrong == 0) { first_bad = i; }
          last_bad = i;
          ++wrong;
          if (shown < 6) {
            ++shown;
            // what does the wrong key correspond to?
            const char * what = "unexplained";
            char buf[96];
            for (int64_t j = (int64_t)i - 512; j <= (int64_t)i + 512 && what[0] == 'u'; ++j) {
              if (j < 0 || j >= (int64_t)n) { continue; }
              for (int w2 = 0; w2 < 2; ++w2) {
                if (h_anchor_tag(db.seqs.data() + db.seq_off[j], db.seqlen[j], w2) == got) { snprintf(buf, sizeof buf, "key %d of amplicon i%+lld", w2, (long long)(j - (int64_t)i)); what = buf; }
              }
            }
            for (uint32_t l2 = 32; l2 <= db.longest && what[0] == 'u'; ++l2) {
              if (l2 <= db.seqlen[i] + 32 && h_anchor_tag(db.seqs.data() + db.seq_off[i], l2, which) == got) { snprintf(buf, sizeof buf, "own sequence read with len %u (true %u)", l2, db.seqlen[i]); what = buf; }
            }
            printf("    which %d amplicon %u (block %u, wave %u of its block) slot %u: got %016llx want %016llx — %s\n", which, i, i / 256, (i % 256) / 64, s,
                   (unsigned long long)got, (unsigned long long)want[which][i], what);
          }
        }
        if (wrong != 0) { printf("  %s round %d which %d: %u wrong keys, amplicons %u..%u\n", names[flow], r, which, wrong, first_bad, last_bad); bad_this = 1; }
      }
      bad_rounds += bad_this;
      if (flow != 3) { free_anchor(a); }
    }
    if (flow == 3) { free_anchor(a); }
    printf("%-14s %d / %d rounds with wrong anchor keys\n", names[flow], bad_rounds, rounds);
    total_bad += bad_rounds;
  }
  return total_bad != 0 ? 1 : 0;
}
