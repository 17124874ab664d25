// oracle/ref_shim.cc — TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrappers around the UNMODIFIED reference's hot-path functions so
// that tests (and only tests / fixture generators) can call them through
// ctypes.  This file is this repo's own code; it #includes the reference's
// headers from /root/reference/src at build time (oracle/Makefile, target
// `ref`) and is linked with the reference's own objects into
// oracle/_ref/libswarmref.so.  Nothing here is compiled or shipped when
// /root/reference is absent.
//
// NOTE: the reference keeps one `static std::mt19937_64 rand_64(1)` PER
// TRANSLATION UNIT (src/utils/pseudo_rng.h:30-31), so zobrist_init(),
// bloom_init() and bloomflex_init() each give the reference's run-time tables
// only the FIRST time they are called in a process.  The wrappers below
// refuse a second call.

#include "bloomflex.h"
#include "bloompat.h"
#include "nw.h"
#include "qgram.h"
#include "utils/hashtable_size.h"
#include "utils/nt_codec.h"
#include "utils/score_matrix.h"
#include "variants.h"
#include "zobrist.h"

#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
std::vector<uint64_t> g_zob_base;
std::vector<uint64_t> g_zob_byte;
bool g_zob_ready = false;
bool g_bloom_ready = false;
bool g_bloomflex_ready = false;
struct bloom_s g_bloom;
struct bloomflex_s g_bloomflex;
}  // namespace

extern "C" {

// zobrist.cc:111-117
int ref_zobrist_init(unsigned int zobrist_len) {
  if (g_zob_ready) { return -1; }
  zobrist_init(zobrist_len, g_zob_base, g_zob_byte);
  g_zob_ready = true;
  return 0;
}

// copy of the 4*zobrist_len table produced by zobrist.cc:49-80
uint64_t ref_zobrist_table(uint64_t * out, uint64_t capacity) {
  const uint64_t n = g_zob_base.size() < capacity ? g_zob_base.size() : capacity;
  std::memcpy(out, g_zob_base.data(), n * sizeof(uint64_t));
  return g_zob_base.size();
}

uint64_t ref_zobrist_hash(char const * seq, unsigned int len) { return zobrist_hash(seq, len); }
uint64_t ref_zobrist_hash_delete_first(char const * seq, unsigned int len) { return zobrist_hash_delete_first(seq, len); }
uint64_t ref_zobrist_hash_insert_first(char const * seq, unsigned int len) { return zobrist_hash_insert_first(seq, len); }
uint64_t ref_zobrist_value(unsigned int pos, unsigned char base) { return zobrist_value(pos, base); }

// variants.cc:184-249; out arrays need 7*len+4 entries
unsigned int ref_generate_variants(char const * seq, unsigned int len, uint64_t hash,
                                   uint64_t * out_hash, uint32_t * out_pos,
                                   uint8_t * out_type, uint8_t * out_base) {
  std::vector<struct var_s> list(7ULL * len + 5);
  const unsigned int n = generate_variants(seq, len, hash, list);
  for (unsigned int i = 0; i < n; ++i) {
    out_hash[i] = list[i].hash;
    out_pos[i] = list[i].pos;
    out_type[i] = static_cast<uint8_t>(list[i].type);
    out_base[i] = list[i].base;
  }
  return n;
}

// variants.cc:118-165
int ref_check_variant(char const * seed, unsigned int seed_len, unsigned int pos,
                      uint8_t type, uint8_t base, char const * amp, unsigned int amp_len) {
  struct var_s v;
  v.hash = 0; v.pos = pos; v.type = static_cast<Variant_type>(type); v.base = base; v.dummy = 0;
  return check_variant(seed, seed_len, v, amp, amp_len) ? 1 : 0;
}

// variants.cc:78-115; out needs nt_bytelength(len+1) bytes
unsigned int ref_generate_variant_sequence(char const * seed, unsigned int seed_len, unsigned int pos,
                                           uint8_t type, uint8_t base, char * out, uint64_t out_bytes) {
  struct var_s v;
  v.hash = 0; v.pos = pos; v.type = static_cast<Variant_type>(type); v.base = base; v.dummy = 0;
  std::vector<char> buf(out_bytes, 0);
  unsigned int len = 0;
  generate_variant_sequence(seed, seed_len, v, buf, len);
  std::memcpy(out, buf.data(), out_bytes);
  return len;
}

// utils/hashtable_size.cc:29-42
uint64_t ref_hashtable_size(uint64_t n) { return compute_hashtable_size(n); }

// bloompat.cc:74-120 — the 1024 patterns of the amplicon Bloom filter
int ref_bloom_patterns(uint64_t * out1024) {
  if (not g_bloom_ready) {
    bloom_init(64, g_bloom);
    g_bloom_ready = true;
  }
  std::memcpy(out1024, g_bloom.patterns.data(), 1024 * sizeof(uint64_t));
  return 0;
}

// bloomflex.cc:72-115 — the 65536 patterns with k bits (first call only)
int ref_bloomflex_patterns(unsigned int k, uint64_t * out65536) {
  if (g_bloomflex_ready) { return -1; }
  bloomflex_init(64, k, g_bloomflex);
  g_bloomflex_ready = true;
  std::memcpy(out65536, g_bloomflex.patterns_v.data(), 65536 * sizeof(uint64_t));
  return 0;
}

// qgram.cc:68-96 — 128-byte q-gram parity vector
void ref_findqgrams(char const * seq, uint64_t len, unsigned char * out128) {
  std::memset(out128, 0, 128);
  findqgrams(seq, len, out128);
}

// nw.cc:237-255 with the int64 score matrix of utils/score_matrix.h:36-64;
// returns nwdiff, writes the raw alignment ops (one char per column) and its length
uint64_t ref_nw(char const * dseq, uint64_t dlen, char const * qseq, uint64_t qlen,
                int64_t mismatch, uint64_t gapopen, uint64_t gapextend,
                char * out_alignment, uint64_t * out_alignment_len) {
  auto const matrix = create_score_matrix<int64_t>(mismatch);
  std::vector<unsigned char> directions(dlen * qlen + 1);
  std::vector<uint64_t> hearray(2 * qlen + 2);
  std::vector<char> raw;
  uint64_t nwdiff = 0;
  nw(dseq, dlen, qseq, qlen, matrix, gapopen, gapextend, nwdiff, directions, hearray, raw);
  if (out_alignment_len != nullptr) { *out_alignment_len = raw.size(); }
  if (out_alignment != nullptr) { std::memcpy(out_alignment, raw.data(), raw.size()); }
  return nwdiff;
}

}  // extern "C"
