// hostdb.h — the host-resident amplicon database (what the reference keeps in the
// global seqindex[] / data_v blob, src/db.cc:65-67, src/utils/seqinfo.h:27-41).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/swarm_amd.h"
#include "../../../include/swarm_amd_host.h"

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the big
// arrays below are filled by parallel loops right after being sized, and a value-initialising
// resize would first zero (and page-fault) hundreds of MB on one thread.
template <class T>
struct swa_default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = swa_default_init_allocator<U>; };
  using std::allocator<T>::allocator;
  template <class U> void construct(U * p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U * p, A &&... a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using swa_vec = std::vector<T, swa_default_init_allocator<T>>;

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
};
