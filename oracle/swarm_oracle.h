/*
 * oracle/swarm_oracle.h — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by
 * or called from the product (swarm_amd/…): only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * A plain-C, single-threaded restatement of the algorithms on swarm 3.1.6's
 * amplicon neighbour-finding path.  Every function cites the reference
 * file:line (relative to /root/reference/) it restates.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks every function
 * here against the unmodified reference (oracle/_ref/libswarmref.so and
 * oracle/_ref/swarm, built by oracle/Makefile from /root/reference) and
 * tests/test_oracle_golden.py checks it against the committed fixtures under
 * tests/golden/ that were produced by that reference binary
 * (tests/golden/make_golden.py).
 *
 * Data model (same as the reference after db_read): amplicons are numbered in
 * "db order" = abundance descending, then header bytes ascending
 * (src/db.cc:388-413).  Sequences are 2-bit packed, A0 C1 G2 T3, 32 nt per
 * little-endian u64, LSB first, zero padded (src/db.cc:541-628,
 * src/utils/nt_codec.cc:35-75).  Here each amplicon's words start at
 * seq_off[i] (in u64 words) inside `seqs`.
 */
#ifndef SWARM_ORACLE_H
#define SWARM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint32_t n;                 /* number of amplicons */
  uint32_t longest;           /* longest sequence, nt */
  const uint64_t * seqs;      /* packed words of all amplicons */
  const uint64_t * seq_off;   /* n+1 word offsets */
  const uint32_t * seqlen;    /* n lengths (nt) */
  const uint64_t * abundance; /* n abundances */
} orc_db;

/* ---- std::mt19937_64 (src/utils/pseudo_rng.h:30-31: seed 1) ------------------ */
typedef struct { uint64_t mt[312]; int idx; } orc_mt64;
void     orc_mt64_seed(orc_mt64 * g, uint64_t seed);
uint64_t orc_mt64_next(orc_mt64 * g);

/* ---- 2-bit codec (src/utils/nt_codec.cc:35-75) -------------------------------- */
static inline unsigned orc_nt(const uint64_t * seq, uint32_t pos) {
  return (unsigned)((seq[pos >> 5] >> ((pos & 31U) << 1)) & 3U);
}
static inline uint32_t orc_nt_words(uint32_t len) { return (len + 31U) >> 5; }

/* ---- Zobrist hashing (src/zobrist.cc:49-80, 127-240) -------------------------- */
/* fills tab[4*zobrist_len] exactly as the first zobrist_init() call of a process does */
void     orc_zobrist_table(uint32_t zobrist_len, uint64_t * tab);
uint64_t orc_zobrist_hash(const uint64_t * tab, const uint64_t * seq, uint32_t len);
uint64_t orc_zobrist_hash_delete_first(const uint64_t * tab, const uint64_t * seq, uint32_t len);
uint64_t orc_zobrist_hash_insert_first(const uint64_t * tab, const uint64_t * seq, uint32_t len);

/* ---- microvariants (src/variants.h:31-38, src/variants.cc:78-249) ------------- */
enum { ORC_SUBSTITUTION = 0, ORC_DELETION = 1, ORC_INSERTION = 2 };
typedef struct { uint64_t hash; uint32_t pos; uint8_t type; uint8_t base; uint16_t pad; } orc_var;
/* out needs 7*len+4 entries; returns the count; order identical to the reference */
uint32_t orc_generate_variants(const uint64_t * tab, const uint64_t * seq, uint32_t len,
                               uint64_t hash, orc_var * out);
int      orc_check_variant(const uint64_t * seed, uint32_t seed_len, const orc_var * var,
                           const uint64_t * amp, uint32_t amp_len);
/* out needs orc_nt_words(seed_len+1) words, zeroed by the callee; returns new length */
uint32_t orc_generate_variant_sequence(const uint64_t * seed, uint32_t seed_len,
                                       const orc_var * var, uint64_t * out);

/* ---- hash table + Bloom filters ------------------------------------------------ */
uint64_t orc_hashtable_size(uint64_t n);                 /* src/utils/hashtable_size.cc:29-42 */
void     orc_bloom_patterns(uint64_t * out1024);         /* src/bloompat.cc:74-90 */
void     orc_bloomflex_patterns(uint32_t k, uint64_t * out65536); /* src/bloomflex.cc:72-88 */

/* ---- B1: d=1 network (src/algod1.cc:188-208, 558-670, 1122-1171) -------------- */
typedef struct {
  uint64_t   table_size;     /* slots (power of two) */
  uint64_t * hash_values;    /* [table_size] */
  uint32_t * hash_data;      /* [table_size] */
  uint8_t  * hash_occupied;  /* bitmap */
  uint64_t * bloom;          /* table_size/8 words, inverted polarity */
  uint64_t   bloom_mask;
  uint64_t   patterns[1024];
  uint64_t * seqhash;        /* [n] */
  uint64_t * zobrist;        /* [4*(longest+2)] */
  /* statistics of the last orc_d1_network call (SURVEY §2 probe counts) */
  uint64_t   stat_variants, stat_bloom_pass, stat_hash_match, stat_verified;
} orc_d1_index;

/* builds zobrist table, seqhash[], hash table and Bloom exactly as algo_d1_run does;
   returns NULL on allocation failure; *has_duplicate set like src/algod1.cc:1131-1150 */
orc_d1_index * orc_d1_index_build(const orc_db * db, int * has_duplicate);
void           orc_d1_index_free(orc_d1_index * ix);

/* For amplicon `seed`: neighbours in the reference's hit order (variant order).
   hits needs 7*len+5 entries.  src/algod1.cc:558-627 */
uint32_t orc_d1_check_variants(const orc_db * db, orc_d1_index * ix, uint32_t seed,
                               int no_cluster_breaking, uint32_t * hits);

/* Whole network as CSR over amplicons [first, first+count): offsets has count+1
   entries, neighbours has capacity `cap`; rows in hit order.  Returns total hits
   (may exceed cap: then only the first cap are written). */
uint64_t orc_d1_network(const orc_db * db, orc_d1_index * ix, int no_cluster_breaking,
                        uint32_t first, uint32_t count,
                        uint64_t * offsets, uint32_t * neighbours, uint64_t cap);

/* ---- B2: fastidious (src/algod1.cc:244-258, 339-552, 1337-1467) --------------- */
/* is_light[i] != 0 <=> amplicon i belongs to a swarm with mass < boundary.
   light_nt = total length of amplicons in light swarms; bloom_bits = --bloom-bits.
   graft_cand[n] (initialised by the callee to 0xFFFFFFFF) receives, per light
   amplicon, the smallest heavy amplicon id two microvariant steps away.
   counters[0] = light variants, [1] = heavy variants, [2] = graft candidates,
   [3] = bloom m (bits), [4] = k.  Returns 0. */
int orc_d1_fastidious(const orc_db * db, const uint8_t * is_light, uint64_t light_nt,
                      uint32_t bloom_bits, uint32_t * graft_cand, uint64_t * counters);

/* ---- d = 0: dereplication (src/derep.cc:276-354).  first_identical[i] = first amplicon in db
        order with the identical sequence (== i for a first occurrence).  0 on success. */
int orc_derep(const orc_db * db, uint32_t * first_identical);

/* ---- B3: q-gram prefilter (src/qgram.cc:68-96, 247-252) ----------------------- */
void     orc_findqgrams(const uint64_t * seq, uint32_t len, uint8_t * out128);
uint64_t orc_qgram_diff(const uint8_t * a128, const uint8_t * b128);

/* ---- B4: alignment diff (src/nw.cc:40-191 == the scalar specification of
        search8/search16 + backtrack, SURVEY §7 "Hard parts" 1) ------------------- */
/* returns the number of non-identical alignment columns; *alnlen (optional) = columns;
   *score (optional) = optimal cost */
uint64_t orc_nw_diff(const uint64_t * dseq, uint32_t dlen, const uint64_t * qseq, uint32_t qlen,
                     uint64_t mismatch, uint64_t gapopen, uint64_t gapextend,
                     uint64_t * alnlen, uint64_t * score);

#ifdef __cplusplus
}
#endif
#endif
