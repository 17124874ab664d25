/*
 * oracle/swarm_oracle.c — TEST INFRASTRUCTURE ONLY (see swarm_oracle.h).
 *
 * Single-threaded plain-C restatement of swarm 3.1.6's neighbour-finding path.
 * Written from the behaviour of the reference (citations = /root/reference/…);
 * data structures and control flow are this repo's own.
 */
#include "swarm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------
 * std::mt19937_64 — the published MT19937-64 recurrence (Matsumoto & Nishimura),
 * which is what libstdc++'s std::mt19937_64 implements.  The reference draws its
 * Zobrist values and Bloom patterns from `static std::mt19937_64 rand_64(1)`
 * (src/utils/pseudo_rng.h:30-31), one private instance per translation unit.
 * ------------------------------------------------------------------------------ */
void orc_mt64_seed(orc_mt64 * g, uint64_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 312; ++i) {
    g->mt[i] = 6364136223846793005ULL * (g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) + (uint64_t)i;
  }
  g->idx = 312;
}

uint64_t orc_mt64_next(orc_mt64 * g) {
  if (g->idx >= 312) {
    for (int i = 0; i < 312; ++i) {
      const uint64_t x = (g->mt[i] & 0xFFFFFFFF80000000ULL) | (g->mt[(i + 1) % 312] & 0x7FFFFFFFULL);
      g->mt[i] = g->mt[(i + 156) % 312] ^ (x >> 1) ^ ((x & 1ULL) ? 0xB5026F5AA96619E9ULL : 0ULL);
    }
    g->idx = 0;
  }
  uint64_t y = g->mt[g->idx++];
  y ^= (y >> 29) & 0x5555555555555555ULL;
  y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
  y ^= (y << 37) & 0xFFF7EEE000000000ULL;
  y ^= (y >> 43);
  return y;
}

/* ------------------------------------------------------------------------------
 * Zobrist table: src/zobrist.cc:49-80.  Each entry = four successive draws,
 * v = r0; v = (v<<16)^r1; v = (v<<16)^r2; v = (v<<16)^r3.  Entry index = 4*pos+base.
 * ------------------------------------------------------------------------------ */
void orc_zobrist_table(uint32_t zobrist_len, uint64_t * tab) {
  orc_mt64 g;
  orc_mt64_seed(&g, 1);
  for (uint64_t i = 0; i < 4ULL * zobrist_len; ++i) {
    uint64_t v = orc_mt64_next(&g);
    v <<= 16; v ^= orc_mt64_next(&g);
    v <<= 16; v ^= orc_mt64_next(&g);
    v <<= 16; v ^= orc_mt64_next(&g);
    tab[i] = v;
  }
}

/* src/zobrist.cc:134-184: XOR over all positions of tab[4*pos + base(pos)] (the
   reference goes through a byte-combined table, which is the same XOR regrouped). */
uint64_t orc_zobrist_hash(const uint64_t * tab, const uint64_t * seq, uint32_t len) {
  uint64_t h = 0;
  for (uint32_t p = 0; p < len; ++p) h ^= tab[4ULL * p + orc_nt(seq, p)];
  return h;
}

/* src/zobrist.cc:189-212: hash of seq[1..len) placed at positions 0..len-2 */
uint64_t orc_zobrist_hash_delete_first(const uint64_t * tab, const uint64_t * seq, uint32_t len) {
  uint64_t h = 0;
  for (uint32_t p = 1; p < len; ++p) h ^= tab[4ULL * (p - 1) + orc_nt(seq, p)];
  return h;
}

/* src/zobrist.cc:215-240: hash of seq[0..len) placed at positions 1..len */
uint64_t orc_zobrist_hash_insert_first(const uint64_t * tab, const uint64_t * seq, uint32_t len) {
  uint64_t h = 0;
  for (uint32_t p = 0; p < len; ++p) h ^= tab[4ULL * (p + 1) + orc_nt(seq, p)];
  return h;
}

/* ------------------------------------------------------------------------------
 * generate_variants: src/variants.cc:184-249.  Order: substitutions by position
 * then base; deletions (one per homopolymer run, reported at the run's first
 * position); insertions: 4 at position 0, then after every position the 3 bases
 * that differ from that position's base.
 * ------------------------------------------------------------------------------ */
static inline uint64_t zv(const uint64_t * tab, uint32_t pos, unsigned base) { return tab[4ULL * pos + base]; }

uint32_t orc_generate_variants(const uint64_t * tab, const uint64_t * seq, uint32_t len,
                               uint64_t hash, orc_var * out) {
  uint32_t n = 0;
  /* substitutions, variants.cc:192-206 */
  for (uint32_t p = 0; p < len; ++p) {
    const unsigned cur = orc_nt(seq, p);
    const uint64_t h1 = hash ^ zv(tab, p, cur);
    for (unsigned b = 0; b < 4; ++b) {
      if (b == cur) continue;
      out[n].hash = h1 ^ zv(tab, p, b); out[n].pos = p; out[n].type = ORC_SUBSTITUTION;
      out[n].base = (uint8_t)b; out[n].pad = 0; ++n;
    }
  }
  /* deletions, variants.cc:210-222 */
  uint64_t h = orc_zobrist_hash_delete_first(tab, seq, len);
  out[n].hash = h; out[n].pos = 0; out[n].type = ORC_DELETION; out[n].base = 0; out[n].pad = 0; ++n;
  unsigned prev = orc_nt(seq, 0);
  for (uint32_t p = 1; p < len; ++p) {
    const unsigned cur = orc_nt(seq, p);
    if (cur == prev) continue;
    h ^= zv(tab, p - 1, prev) ^ zv(tab, p - 1, cur);
    out[n].hash = h; out[n].pos = p; out[n].type = ORC_DELETION; out[n].base = 0; out[n].pad = 0; ++n;
    prev = cur;
  }
  /* insertions, variants.cc:226-246 */
  h = orc_zobrist_hash_insert_first(tab, seq, len);
  for (unsigned b = 0; b < 4; ++b) {
    out[n].hash = h ^ zv(tab, 0, b); out[n].pos = 0; out[n].type = ORC_INSERTION;
    out[n].base = (uint8_t)b; out[n].pad = 0; ++n;
  }
  for (uint32_t p = 0; p < len; ++p) {
    const unsigned cur = orc_nt(seq, p);
    h ^= zv(tab, p, cur) ^ zv(tab, p + 1, cur);
    for (unsigned b = 0; b < 4; ++b) {
      if (b == cur) continue;
      out[n].hash = h ^ zv(tab, p + 1, b); out[n].pos = p + 1; out[n].type = ORC_INSERTION;
      out[n].base = (uint8_t)b; out[n].pad = 0; ++n;
    }
  }
  return n;
}

/* variants.cc:61-75 */
static int seq_identical(const uint64_t * a, uint32_t a0, const uint64_t * b, uint32_t b0, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) if (orc_nt(a, a0 + i) != orc_nt(b, b0 + i)) return 0;
  return 1;
}

/* variants.cc:118-165 */
int orc_check_variant(const uint64_t * seed, uint32_t seed_len, const orc_var * var,
                      const uint64_t * amp, uint32_t amp_len) {
  const uint32_t p = var->pos;
  switch (var->type) {
    case ORC_SUBSTITUTION:
      return seed_len == amp_len && seq_identical(seed, 0, amp, 0, p) &&
             orc_nt(amp, p) == var->base && seq_identical(seed, p + 1, amp, p + 1, seed_len - p - 1);
    case ORC_DELETION:
      return seed_len - 1 == amp_len && seq_identical(seed, 0, amp, 0, p) &&
             seq_identical(seed, p + 1, amp, p, seed_len - p - 1);
    case ORC_INSERTION:
      return seed_len + 1 == amp_len && seq_identical(seed, 0, amp, 0, p) &&
             orc_nt(amp, p) == var->base && seq_identical(seed, p, amp, p + 1, seed_len - p);
    default:
      return 0;
  }
}

static inline void nt_put(uint64_t * seq, uint32_t pos, unsigned base) {
  const unsigned sh = (pos & 31U) << 1;
  seq[pos >> 5] = (seq[pos >> 5] & ~(3ULL << sh)) | ((uint64_t)base << sh);
}

/* variants.cc:78-115 (the result is zero padded here; the reference's buffer may carry
   stale bits beyond the new length, which nothing reads) */
uint32_t orc_generate_variant_sequence(const uint64_t * seed, uint32_t seed_len,
                                       const orc_var * var, uint64_t * out) {
  const uint32_t p = var->pos;
  uint32_t n = 0;
  memset(out, 0, sizeof(uint64_t) * orc_nt_words(seed_len + 1));
  switch (var->type) {
    case ORC_SUBSTITUTION:
      for (uint32_t i = 0; i < seed_len; ++i) nt_put(out, i, i == p ? var->base : orc_nt(seed, i));
      n = seed_len;
      break;
    case ORC_DELETION:
      for (uint32_t i = 0; i < p; ++i) nt_put(out, i, orc_nt(seed, i));
      for (uint32_t i = p + 1; i < seed_len; ++i) nt_put(out, i - 1, orc_nt(seed, i));
      n = seed_len - 1;
      break;
    case ORC_INSERTION:
      for (uint32_t i = 0; i < p; ++i) nt_put(out, i, orc_nt(seed, i));
      nt_put(out, p, var->base);
      for (uint32_t i = p; i < seed_len; ++i) nt_put(out, i + 1, orc_nt(seed, i));
      n = seed_len + 1;
      break;
    default:
      break;
  }
  return n;
}

/* ------------------------------------------------------------------------------
 * Table sizing: src/utils/hashtable_size.cc:29-42 — the integer quotient
 * 10*(n+1)/7 is taken FIRST (integer division), then 2^ceil(log2(.)) in doubles.
 * ------------------------------------------------------------------------------ */
uint64_t orc_hashtable_size(uint64_t n) {
  const uint64_t q = 10ULL * (n + 1ULL) / 7ULL;
  return (uint64_t)pow(2.0, ceil(log((double)q) / log(2.0)));
}

/* src/bloompat.cc:74-90: 1024 patterns with exactly 8 distinct bits, generator private
   to bloompat.cc, so the draws start from a fresh mt19937_64(1). */
static void patterns_generate(uint64_t * out, uint32_t count, uint32_t k) {
  orc_mt64 g;
  orc_mt64_seed(&g, 1);
  for (uint32_t i = 0; i < count; ++i) {
    uint64_t pat = 0;
    for (uint32_t j = 0; j < k; ++j) {
      uint64_t bit = 1ULL << (orc_mt64_next(&g) & 63U);
      while (pat & bit) bit = 1ULL << (orc_mt64_next(&g) & 63U);
      pat |= bit;
    }
    out[i] = pat;
  }
}
void orc_bloom_patterns(uint64_t * out1024) { patterns_generate(out1024, 1024, 8); }
/* src/bloomflex.cc:72-88: 65536 patterns with k bits, generator private to bloomflex.cc */
void orc_bloomflex_patterns(uint32_t k, uint64_t * out65536) { patterns_generate(out65536, 65536, k); }

/* ------------------------------------------------------------------------------
 * B1 index: src/hashtable.cc:47-146, src/bloompat.cc:46-120, src/algod1.cc:188-208,
 * 1122-1150.
 * ------------------------------------------------------------------------------ */
static inline int occ_get(const uint8_t * bm, uint64_t i) { return (bm[i >> 3] >> (i & 7U)) & 1; }
static inline void occ_set(uint8_t * bm, uint64_t i) { bm[i >> 3] |= (uint8_t)(1U << (i & 7U)); }

static inline void bloom_set(orc_d1_index * ix, uint64_t h) {      /* bloompat.cc:62-65 */
  ix->bloom[(h >> 10) & ix->bloom_mask] &= ~ix->patterns[h & 1023U];
}
static inline int bloom_get(const orc_d1_index * ix, uint64_t h) { /* bloompat.cc:68-71 */
  return (ix->bloom[(h >> 10) & ix->bloom_mask] & ix->patterns[h & 1023U]) == 0;
}

static int amp_identical(const orc_db * db, uint32_t a, uint32_t b) {  /* algod1.cc:174-185 */
  if (db->seqlen[a] != db->seqlen[b]) return 0;
  return memcmp(db->seqs + db->seq_off[a], db->seqs + db->seq_off[b],
                sizeof(uint64_t) * orc_nt_words(db->seqlen[a])) == 0;
}

/* algod1.cc:188-208 */
static int table_insert(const orc_db * db, orc_d1_index * ix, uint32_t amp) {
  const uint64_t h = ix->seqhash[amp];
  const uint64_t mask = ix->table_size - 1;
  uint64_t i = (h >> 32) & mask;                                     /* hashtable.cc:47-53 */
  int dup = 0;
  while (occ_get(ix->hash_occupied, i)) {
    if (ix->hash_values[i] == h && amp_identical(db, amp, ix->hash_data[i])) dup = 1;
    i = (i + 1) & mask;
  }
  occ_set(ix->hash_occupied, i);
  ix->hash_values[i] = h;
  ix->hash_data[i] = amp;
  bloom_set(ix, h);
  return dup;
}

orc_d1_index * orc_d1_index_build(const orc_db * db, int * has_duplicate) {
  orc_d1_index * ix = (orc_d1_index *)calloc(1, sizeof(orc_d1_index));
  if (!ix) return NULL;
  ix->table_size = orc_hashtable_size(db->n);
  ix->hash_values = (uint64_t *)calloc(ix->table_size, sizeof(uint64_t));
  ix->hash_data = (uint32_t *)calloc(ix->table_size, sizeof(uint32_t));
  ix->hash_occupied = (uint8_t *)calloc((ix->table_size + 63) / 8, 1);
  /* bloom_init(hashtablesize): size in BYTES, at least 8 (bloompat.cc:100-120) */
  uint64_t bloom_bytes = ix->table_size < 8 ? 8 : ix->table_size;
  ix->bloom = (uint64_t *)malloc(bloom_bytes);
  ix->bloom_mask = (bloom_bytes >> 3) - 1;
  ix->seqhash = (uint64_t *)malloc(sizeof(uint64_t) * (db->n ? db->n : 1));
  ix->zobrist = (uint64_t *)malloc(sizeof(uint64_t) * 4 * ((size_t)db->longest + 2));
  if (!ix->hash_values || !ix->hash_data || !ix->hash_occupied || !ix->bloom || !ix->seqhash || !ix->zobrist) {
    orc_d1_index_free(ix);
    return NULL;
  }
  memset(ix->bloom, 0xFF, bloom_bytes);
  orc_bloom_patterns(ix->patterns);
  /* db.cc:652-653: zobrist_len = max(4*longest_header, longest+2); the table is a
     prefix-stable stream, so longest+2 entries are all this path ever reads */
  orc_zobrist_table(db->longest + 2, ix->zobrist);
  for (uint32_t i = 0; i < db->n; ++i) {                            /* db.cc:761 */
    ix->seqhash[i] = orc_zobrist_hash(ix->zobrist, db->seqs + db->seq_off[i], db->seqlen[i]);
  }
  int dup = 0;
  for (uint32_t i = 0; i < db->n; ++i) {                            /* algod1.cc:1131-1139 */
    if (table_insert(db, ix, i)) { dup = 1; break; }
  }
  if (has_duplicate) *has_duplicate = dup;
  return ix;
}

void orc_d1_index_free(orc_d1_index * ix) {
  if (!ix) return;
  free(ix->hash_values); free(ix->hash_data); free(ix->hash_occupied);
  free(ix->bloom); free(ix->seqhash); free(ix->zobrist);
  free(ix);
}

/* algod1.cc:558-627 */
uint32_t orc_d1_check_variants(const orc_db * db, orc_d1_index * ix, uint32_t seed,
                               int no_cluster_breaking, uint32_t * hits) {
  const uint64_t * sseq = db->seqs + db->seq_off[seed];
  const uint32_t slen = db->seqlen[seed];
  orc_var * vars = (orc_var *)malloc(sizeof(orc_var) * (7ULL * slen + 5));
  const uint32_t nv = orc_generate_variants(ix->zobrist, sseq, slen, ix->seqhash[seed], vars);
  const uint64_t mask = ix->table_size - 1;
  uint32_t nh = 0;
  ix->stat_variants += nv;
  for (uint32_t v = 0; v < nv; ++v) {
    const uint64_t h = vars[v].hash;
    if (!bloom_get(ix, h)) continue;
    ix->stat_bloom_pass++;
    uint64_t i = (h >> 32) & mask;
    while (occ_get(ix->hash_occupied, i)) {
      if (ix->hash_values[i] == h) {
        const uint32_t amp = ix->hash_data[i];
        ix->stat_hash_match++;
        if (amp != seed && (no_cluster_breaking || db->abundance[seed] >= db->abundance[amp])) {
          if (orc_check_variant(sseq, slen, &vars[v], db->seqs + db->seq_off[amp], db->seqlen[amp])) {
            ix->stat_verified++;
            hits[nh++] = amp;
            break;
          }
        }
      }
      i = (i + 1) & mask;
    }
  }
  free(vars);
  return nh;
}

/* algod1.cc:630-670, single thread => rows in amplicon order */
uint64_t orc_d1_network(const orc_db * db, orc_d1_index * ix, int no_cluster_breaking,
                        uint32_t first, uint32_t count,
                        uint64_t * offsets, uint32_t * neighbours, uint64_t cap) {
  uint32_t * hits = (uint32_t *)malloc(sizeof(uint32_t) * (7ULL * db->longest + 5));
  uint64_t total = 0;
  for (uint32_t k = 0; k < count; ++k) {
    offsets[k] = total;
    const uint32_t nh = orc_d1_check_variants(db, ix, first + k, no_cluster_breaking, hits);
    for (uint32_t j = 0; j < nh; ++j) {
      if (total < cap) neighbours[total] = hits[j];
      ++total;
    }
  }
  offsets[count] = total;
  free(hits);
  return total;
}

/* ------------------------------------------------------------------------------
 * B2 fastidious: src/algod1.cc:1337-1467 (sizing + the two passes), 495-518
 * (mark_light_var), 374-450 (check_heavy_var / _2), 339-371 (hash_check_attach),
 * 244-258 (add_graft_candidate).  Single-threaded, so bloomflex_set's
 * unsynchronised &= (bloomflex.cc:61-64) is exact here (== reference with -t 1).
 * ------------------------------------------------------------------------------ */
int orc_d1_fastidious(const orc_db * db, const uint8_t * is_light, uint64_t light_nt,
                      uint32_t bloom_bits, uint32_t * graft_cand, uint64_t * counters) {
  int dupdummy = 0;
  orc_d1_index * ix = orc_d1_index_build(db, &dupdummy);
  if (!ix) return 1;
  const uint64_t mask = ix->table_size - 1;
  for (uint32_t i = 0; i < db->n; ++i) graft_cand[i] = 0xFFFFFFFFU;

  /* sizing, algod1.cc:1337-1357, 1383-1403 */
  uint32_t k = (uint32_t)(0.4 * (double)bloom_bits);
  if (k < 1) k = 1;
  uint64_t m = light_nt * 7ULL * bloom_bits;
  if (m < 64) m = 64;
  const uint64_t n_bytes = ((m - 1) / 8) + 1;
  const uint64_t fsize = n_bytes >> 3;                              /* bloomflex.cc:101 */
  uint64_t * fbits = (uint64_t *)malloc(sizeof(uint64_t) * (fsize ? fsize : 1));
  uint64_t * fpat = (uint64_t *)malloc(sizeof(uint64_t) * 65536);
  const uint32_t maxv = 7 * (db->longest + 1) + 5;
  orc_var * v1 = (orc_var *)malloc(sizeof(orc_var) * maxv);
  orc_var * v2 = (orc_var *)malloc(sizeof(orc_var) * maxv);
  uint64_t * vseq = (uint64_t *)malloc(sizeof(uint64_t) * (orc_nt_words(db->longest + 2) + 1));
  if (!fbits || !fpat || !v1 || !v2 || !vseq) return 1;
  memset(fbits, 0xFF, sizeof(uint64_t) * fsize);
  orc_bloomflex_patterns(k, fpat);

  /* empty the table and the amplicon Bloom, algod1.cc:1411-1412 */
  memset(ix->hash_occupied, 0, (ix->table_size + 63) / 8);
  memset(ix->bloom, 0xFF, (ix->bloom_mask + 1) * 8);

  /* pass A, least to most abundant: algod1.cc:521-552, 495-518 */
  uint64_t light_variants = 0;
  for (uint32_t a = db->n; a-- > 0;) {
    if (!is_light[a]) continue;
    table_insert(db, ix, a);
    const uint32_t nv = orc_generate_variants(ix->zobrist, db->seqs + db->seq_off[a], db->seqlen[a],
                                              ix->seqhash[a], v1);
    for (uint32_t v = 0; v < nv; ++v) {
      const uint64_t h = v1[v].hash;
      fbits[(h >> 16) % fsize] &= ~fpat[h & 0xFFFFU];               /* bloomflex.cc:43-64 */
    }
    light_variants += nv;
  }

  /* pass B, most to least abundant: algod1.cc:453-492, 398-450, 374-395, 339-371 */
  uint64_t heavy_variants = 0, candidates = 0;
  for (uint32_t s = 0; s < db->n; ++s) {
    if (is_light[s]) continue;
    const uint64_t * sseq = db->seqs + db->seq_off[s];
    const uint32_t slen = db->seqlen[s];
    const uint32_t nv = orc_generate_variants(ix->zobrist, sseq, slen, ix->seqhash[s], v1);
    heavy_variants += nv;
    for (uint32_t v = 0; v < nv; ++v) {
      const uint64_t h = v1[v].hash;
      if ((fbits[(h >> 16) % fsize] & fpat[h & 0xFFFFU]) != 0) continue;   /* bloomflex_get */
      const uint32_t vlen = orc_generate_variant_sequence(sseq, slen, &v1[v], vseq);
      const uint64_t vh = orc_zobrist_hash(ix->zobrist, vseq, vlen);
      const uint32_t nv2 = orc_generate_variants(ix->zobrist, vseq, vlen, vh, v2);
      for (uint32_t w = 0; w < nv2; ++w) {
        const uint64_t h2 = v2[w].hash;
        if (!bloom_get(ix, h2)) continue;
        uint64_t i = (h2 >> 32) & mask;
        while (occ_get(ix->hash_occupied, i)) {
          if (ix->hash_values[i] == h2) {
            const uint32_t amp = ix->hash_data[i];
            if (orc_check_variant(vseq, vlen, &v2[w], db->seqs + db->seq_off[amp], db->seqlen[amp])) {
              ++candidates;                                         /* algod1.cc:244-258 */
              if (graft_cand[amp] == 0xFFFFFFFFU || graft_cand[amp] > s) graft_cand[amp] = s;
              break;
            }
          }
          i = (i + 1) & mask;
        }
      }
    }
  }
  counters[0] = light_variants;
  counters[1] = heavy_variants;
  counters[2] = candidates;
  counters[3] = m;
  counters[4] = k;
  free(vseq); free(v2); free(v1); free(fpat); free(fbits);
  orc_d1_index_free(ix);
  return 0;
}

/* ------------------------------------------------------------------------------
 * B3: src/qgram.cc:68-96 (signature), 247-252 (bound), popcount per popcnt.cc:45-62
 * ------------------------------------------------------------------------------ */
void orc_findqgrams(const uint64_t * seq, uint32_t len, uint8_t * out128) {
  memset(out128, 0, 128);
  uint64_t q = 0;
  uint32_t p = 0;
  while (p < 4 && p < len) { q = (q << 2) | orc_nt(seq, p); ++p; }
  while (p < len) {
    q = (q << 2) | orc_nt(seq, p);
    out128[(q >> 3) & 127U] ^= (uint8_t)(1U << (q & 7U));
    ++p;
  }
}

uint64_t orc_qgram_diff(const uint8_t * a128, const uint8_t * b128) {
  uint64_t pop = 0;
  for (int i = 0; i < 128; ++i) pop += (uint64_t)__builtin_popcount((unsigned)(a128[i] ^ b128[i]));
  return (pop + 9) / 10;
}

/* ------------------------------------------------------------------------------
 * B4: src/nw.cc:40-191 — minimum-cost global alignment with affine gaps, tie-broken
 * backtrack; returns aligned columns - identical columns.  SURVEY §7/§8c: this
 * scalar routine is the specification of search8/search16 + backtrack<> for every
 * pair whose score does not saturate.  dseq = target/database (rows), qseq = query
 * (columns).  Match cost 0, mismatch cost `mismatch` (score_matrix.h:36-64 for the
 * symbols 1..4 that nucleotides map to).
 * ------------------------------------------------------------------------------ */
uint64_t orc_nw_diff(const uint64_t * dseq, uint32_t dlen, const uint64_t * qseq, uint32_t qlen,
                     uint64_t mismatch, uint64_t gapopen, uint64_t gapextend,
                     uint64_t * alnlen, uint64_t * score) {
  uint8_t * dir = (uint8_t *)calloc((size_t)dlen * qlen + 1, 1);
  uint64_t * he = (uint64_t *)malloc(sizeof(uint64_t) * (2ULL * qlen + 2));
  for (uint64_t c = 0; c < qlen; ++c) {                             /* nw.cc:66-70 */
    he[2 * c] = gapopen + (c + 1) * gapextend;
    he[2 * c + 1] = 2 * gapopen + (c + 2) * gapextend;
  }
  uint64_t last_h = 0;
  for (uint64_t r = 0; r < dlen; ++r) {                             /* nw.cc:75-113 */
    uint64_t top = 2 * gapopen + (r + 2) * gapextend;
    uint64_t diag = (r == 0) ? 0 : gapopen + r * gapextend;
    const unsigned dnt = orc_nt(dseq, (uint32_t)r);
    for (uint64_t c = 0; c < qlen; ++c) {
      uint8_t * cell = &dir[qlen * r + c];
      const uint64_t prev_diag = he[2 * c];
      uint64_t left = he[2 * c + 1];
      diag += (dnt == orc_nt(qseq, (uint32_t)c)) ? 0 : mismatch;
      if (top < diag) *cell |= 1;                                   /* maskup */
      if (top < diag) diag = top;
      if (left < diag) diag = left;
      if (left == diag) *cell |= 2;                                 /* maskleft */
      he[2 * c] = diag;
      last_h = diag;
      diag += gapopen + gapextend;
      left += gapextend;
      top += gapextend;
      if (top < diag) *cell |= 4;                                   /* maskextup */
      if (left < diag) *cell |= 8;                                  /* maskextleft */
      if (diag < top) top = diag;
      if (diag < left) left = diag;
      he[2 * c + 1] = left;
      diag = prev_diag;
    }
  }
  /* backtrack, nw.cc:116-191 */
  uint64_t alength = 0, matches = 0;
  uint64_t col = qlen, row = dlen;
  char op = 0;
  while (col > 0 && row > 0) {
    const uint8_t cell = dir[qlen * (row - 1) + (col - 1)];
    ++alength;
    if (op == 'I' && (cell & 8)) { --row; op = 'I'; }
    else if (op == 'D' && (cell & 4)) { --col; op = 'D'; }
    else if (cell & 2) { --row; op = 'I'; }
    else if (cell & 1) { --col; op = 'D'; }
    else {
      if (orc_nt(qseq, (uint32_t)(col - 1)) == orc_nt(dseq, (uint32_t)(row - 1))) ++matches;
      --col; --row; op = 'M';
    }
  }
  alength += col + row;
  if (alnlen) *alnlen = alength;
  if (score) *score = last_h;
  free(he); free(dir);
  return alength - matches;
}

/* ------------------------------------------------------------------------------
 * d = 0: dereplication, src/derep.cc:276-354 (dereplicating).  Buckets are found by
 * hash & (tablesize - 1) with linear probing and wrap-around; a bucket matches when the hash,
 * the length and the packed sequence are all equal.  Output: per amplicon the first amplicon
 * (in db order) with the identical sequence — the reference's bucket.seqno_first.
 * ------------------------------------------------------------------------------ */
int orc_derep(const orc_db * db, uint32_t * first_identical) {
  const uint32_t n = db->n;
  if (n == 0) return 0;
  const uint64_t tsize = orc_hashtable_size(n);                     /* derep.cc:387 */
  const uint64_t mask = tsize - 1;
  uint64_t * bhash = (uint64_t *)calloc(tsize, sizeof(uint64_t));
  uint32_t * bfirst = (uint32_t *)malloc(tsize * sizeof(uint32_t));
  uint8_t * used = (uint8_t *)calloc(tsize, 1);                     /* the reference tests mass != 0 */
  const uint32_t zlen = db->longest + 2;
  uint64_t * tab = (uint64_t *)malloc(4ULL * zlen * sizeof(uint64_t));
  if (!bhash || !bfirst || !used || !tab) { free(bhash); free(bfirst); free(used); free(tab); return 1; }
  orc_zobrist_table(zlen, tab);
  for (uint32_t i = 0; i < n; ++i) {
    const uint64_t * seq = db->seqs + db->seq_off[i];
    const uint32_t len = db->seqlen[i];
    const uint64_t h = orc_zobrist_hash(tab, seq, len);
    uint64_t b = h & mask;                                          /* derep.cc:299 */
    while (used[b]) {
      const uint32_t f = bfirst[b];
      if (bhash[b] == h && db->seqlen[f] == len &&
          memcmp(seq, db->seqs + db->seq_off[f], 8ULL * ((len + 31U) / 32U)) == 0) break;
      b = (b + 1) & mask;                                           /* derep.cc:308-315 */
    }
    if (!used[b]) { used[b] = 1; bhash[b] = h; bfirst[b] = i; }     /* derep.cc:325-332 */
    first_identical[i] = bfirst[b];
  }
  free(bhash); free(bfirst); free(used); free(tab);
  return 0;
}
