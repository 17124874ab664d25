// host_tables.cpp — the small constant tables of the d=1 path, generated on the host
// and uploaded once.  They are made bit-identical to the reference's so that every
// intermediate of the GPU path (sequence hashes, variant hashes, Bloom bitmap) can be
// compared word for word with the oracle; no output of the program depends on them.
#include "swa_internal.h"

#include <cmath>
#include <random>

// The reference draws from `static std::mt19937_64 rand_64(1)`, one private instance
// per translation unit (src/utils/pseudo_rng.h:30-31), so each table below starts from
// a fresh generator seeded with 1.

// src/zobrist.cc:49-80 — 4 values per position; each value folds four draws with
// 16-bit left shifts in between.
void swa_zobrist_table(uint32_t zobrist_len, std::vector<uint64_t> & tab) {
  std::mt19937_64 gen(1);
  tab.resize(4ull * zobrist_len);
  for (auto & v : tab) {
    uint64_t x = gen();
    for (int k = 0; k < 3; ++k) { x = (x << 16) ^ gen(); }
    v = x;
  }
}

// src/bloompat.cc:74-90 (count 1024, k 8) and src/bloomflex.cc:72-88 (count 65536,
// k = number of hash functions): k distinct bit positions per pattern, redrawn on
// collision.
void swa_bloom_patterns(uint32_t count, uint32_t k, std::vector<uint64_t> & pat) {
  std::mt19937_64 gen(1);
  pat.assign(count, 0);
  for (auto & p : pat) {
    for (uint32_t j = 0; j < k; ++j) {
      uint64_t bit;
      do { bit = 1ull << (gen() & 63u); } while ((p & bit) != 0);
      p |= bit;
    }
  }
}

// src/utils/hashtable_size.cc:29-42 — smallest power of two >= the INTEGER quotient
// 10*(n+1)/7, evaluated through log/ceil/pow in doubles like the reference.
uint64_t swa_hashtable_size(uint64_t n) {
  const uint64_t quotient = 10ull * (n + 1ull) / 7ull;
  return static_cast<uint64_t>(std::pow(2.0, std::ceil(std::log(static_cast<double>(quotient)) / std::log(2.0))));
}
