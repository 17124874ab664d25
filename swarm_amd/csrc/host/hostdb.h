// hostdb.h — the host-resident amplicon database (what the reference keeps in the
// global seqindex[] / data_v blob, src/db.cc:65-67, src/utils/seqinfo.h:27-41).
#pragma once

#include <sys/mman.h>

#include <cstdint>
#include <cstdlib>
#include <memory>
#include <new>
#include <functional>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../../include/swarm_amd.h"
#include "pool.h"
#include "../../../include/swarm_amd_host.h"

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the big
// arrays below are filled by parallel loops right after being sized, and a value-initialising
// resize would first zero (and page-fault) hundreds of MB on one thread.  Large blocks are
// populated up front by a few threads (MADV_POPULATE_WRITE) when many workers will fill them: 64
// threads first-touching 4 KiB pages of one fresh mapping contend in the kernel (the reader's
// gather phase took 270 ms at 10 M amplicons that way, 177 ms populated; transparent huge pages
// do as well but may stall in compaction).  SWARM_AMD_HOST_ALLOC: 0 plain, 1 transparent huge
// pages, n >= 2 populate with n threads (default: 0 up to 32 usable CPUs, else 8).
template <class T>
struct swa_default_init_allocator {
  using value_type = T;
  static constexpr size_t kHugeFrom = size_t(8) << 20, kHugePage = size_t(2) << 20;
  swa_default_init_allocator() = default;
  template <class U> swa_default_init_allocator(const swa_default_init_allocator<U> &) noexcept {}
  template <class U> struct rebind { using other = swa_default_init_allocator<U>; };
  T * allocate(size_t n) {
    const size_t bytes = n * sizeof(T);
    void * p = nullptr;
    if (bytes >= kHugeFrom) {
      // (default: plain first touch by the workers that fill the block when there are few of them — 16 under the bench
      // box's CPU quota: the reader finishes 30 ms earlier than with the blocks populated up front, lease r5r —, populated by
      // 8 threads when there are many: 64 workers first-touching one fresh mapping contend in the kernel, round 4)
      static const int mode = [] { const char * e = std::getenv("SWARM_AMD_HOST_ALLOC"); return e == nullptr ? (swa_host_cpus() > 32u ? 8 : 0) : std::atoi(e); }();
      p = std::aligned_alloc(kHugePage, (bytes + kHugePage - 1) & ~(kHugePage - 1));
      if (p != nullptr && mode == 1) { (void)::madvise(p, bytes, MADV_HUGEPAGE); }
      if (p != nullptr && mode >= 2) {                         // populate with `mode` threads
        std::vector<std::thread> pool;
        const size_t pages = (bytes + 4095) / 4096;
        for (int t = 0; t < mode; ++t) {
          pool.emplace_back([=] {
            const size_t lo = pages * t / mode * 4096, hi = std::min(bytes, pages * (t + 1) / mode * 4096);
            if (hi > lo) { (void)::madvise(static_cast<char *>(p) + lo, hi - lo, 23 /* MADV_POPULATE_WRITE */); }
          });
        }
        for (auto & th : pool) { th.join(); }
      }
    } else {
      p = std::malloc(bytes != 0 ? bytes : 1);
    }
    if (p == nullptr) { throw std::bad_alloc(); }
    return static_cast<T *>(p);
  }
  void deallocate(T * p, size_t) noexcept { std::free(p); }
  template <class U> void construct(U * p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U * p, A &&... a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const swa_default_init_allocator<U> &) const noexcept { return true; }
  template <class U> bool operator!=(const swa_default_init_allocator<U> &) const noexcept { return false; }
};
template <class T> using swa_vec = std::vector<T, swa_default_init_allocator<T>>;


// One FASTA record as parsed.  The parser works on pieces of the file (one per thread); what it produces — the entries,
// their headers, their packed words — STAYS where the parser put it, in file order, and is the database's storage: db
// order is the array `ent` of pointers into it (and, for the GPU, the offsets of the amplicons' words: the device gathers
// them, swa_db_upload_unordered).  Round 4 copied everything into db-order arrays on the host: 0.9 GB more to fault in
// and to take apart at exit, and the largest phase of the reader after the sort (profiles/r05/NOTES.md).
struct swa_entry {
  uint64_t hdr_off;          // into the header pool of its piece
  uint64_t word_off;         // into the word pool of its piece
  uint64_t abundance;
  uint32_t hdr_len_piece;    // header length (24 bits) | piece number << 24
  uint32_t seqlen;
  int32_t ab_start, ab_end;  // abundance annotation [start, end) inside the header
  uint32_t hdr_len() const { return hdr_len_piece & 0xFFFFFFu; }
  uint32_t piece() const { return hdr_len_piece >> 24; }
};
struct swa_piece {
  std::vector<swa_entry> entries;
  std::vector<char> hdr_pool;          // NUL-terminated headers
  std::vector<uint64_t> words;         // packed sequences, every one from a word boundary
};

struct swa_hostdb {
  uint32_t n = 0;
  uint32_t longest = 0;
  uint32_t longest_header = 0;
  uint64_t nucleotides = 0;
  std::vector<swa_piece> pieces;            // the parsed file (at most 255 pieces)
  std::vector<uint64_t> piece_word_first;   // words in the pools of the pieces before piece p (pieces + 1 entries)
  // db order (abundance descending, then header ascending — src/db.cc:388-413)
  swa_vec<const swa_entry *> ent;           // amplicon k
  swa_vec<uint32_t> seqlen;                 // what the GPU upload and the per-swarm sums read as plain arrays
  swa_vec<uint64_t> abundance;
  swa_vec<uint64_t> src_off;                // first word of amplicon k in the concatenation of the pieces' word pools
  // the packed sequences in db order, contiguous (the layout of swa_db_view): made on demand, swa_hostdb_view
  swa_vec<uint64_t> seqs, seq_off;
  bool ordered = false;
  swa_vec<char> scratch;                    // the reader's scratch block (checks, sort), kept until the handle is freed
  char * input_map = nullptr;               // the mapping of the FASTA file, kept until the handle is freed (fasta_db.cpp)
  size_t input_size = 0;
  ~swa_hostdb() { if (input_map != nullptr) { ::munmap(input_map, input_size); } }
  std::vector<const uint64_t *> piece_ptrs;   // what swa_hostdb_unordered_view points at
  std::vector<uint64_t> piece_counts;
  std::string error;

  const char * hdr(uint32_t k) const { const swa_entry * e = ent[k]; return pieces[e->piece()].hdr_pool.data() + e->hdr_off; }
  const uint64_t * words(uint32_t k) const { const swa_entry * e = ent[k]; return pieces[e->piece()].words.data() + e->word_off; }
};
