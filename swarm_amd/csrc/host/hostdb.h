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
#include "../../../include/swarm_amd_host.h"

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the big
// arrays below are filled by parallel loops right after being sized, and a value-initialising
// resize would first zero (and page-fault) hundreds of MB on one thread.  Large blocks are
// populated up front by a few threads (MADV_POPULATE_WRITE): 64 threads first-touching 4 KiB
// pages of one fresh mapping contend in the kernel (the reader's gather phase took 270 ms at
// 10 M amplicons that way, 177 ms populated; transparent huge pages do as well but may stall in
// compaction).  SWARM_AMD_HOST_ALLOC: 0 plain, 1 transparent huge pages, n >= 2 populate with n
// threads (default 8).
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
      static const int mode = [] { const char * e = std::getenv("SWARM_AMD_HOST_ALLOC"); return e == nullptr ? 8 : std::atoi(e); }();
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

// The pages of a big block given back to the kernel in slices by the worker threads (MADV_DONTNEED works under the
// shared memory-map lock, so the slices go in parallel); the block itself stays mapped for its owner to free.  Freeing
// a gigabyte costs one thread ~75 ms on the bench host — at exit, inside the caller's wall time.
void swa_release_pages(void * p, size_t bytes);

struct swa_hostdb {
  uint32_t n = 0;
  uint32_t longest = 0;
  uint32_t longest_header = 0;
  uint64_t nucleotides = 0;
  swa_vec<uint64_t> seqs;        // packed words, db order, contiguous
  swa_vec<uint64_t> seq_off;     // n + 1
  swa_vec<uint32_t> seqlen;
  swa_vec<uint64_t> abundance;
  swa_vec<char> headers;         // NUL-terminated headers, db order
  swa_vec<uint64_t> hdr_off;     // n + 1
  swa_vec<int32_t> ab_start;     // abundance annotation span inside each header
  swa_vec<int32_t> ab_end;
  std::string error;
  ~swa_hostdb() {
    swa_release_pages(seqs.data(), seqs.size() * sizeof(uint64_t));
    swa_release_pages(headers.data(), headers.size());
    swa_release_pages(seq_off.data(), seq_off.size() * sizeof(uint64_t));
    swa_release_pages(hdr_off.data(), hdr_off.size() * sizeof(uint64_t));
    swa_release_pages(abundance.data(), abundance.size() * sizeof(uint64_t));
  }
};
