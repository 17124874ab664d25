// out.h — buffered output for the writers (the reference prints every identifier with its own
// fprintf; at 10 M amplicons that alone costs a second) and the three identifier formats of
// src/db.cc:946-1026.
#pragma once

#include <algorithm>
#include <atomic>
#include <thread>
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <omp.h>

#include "hostdb.h"
#include "pool.h"

class BufOut {
 public:
  // memory-only sink: formatted bytes pile up in the buffer until take() (used by writers that
  // format ranges of clusters on several threads and then emit the pieces in order)
  BufOut() : memory_only_(true) {}
  std::string take() { std::string out; out.swap(buf_); return out; }
  void reserve(size_t bytes) { buf_.reserve(bytes); }
  explicit BufOut(const char * path) {
    if (path == nullptr) { return; }
    if (std::strcmp(path, "-") == 0) { fp_ = stdout; owned_ = false; }
    else { fp_ = std::fopen(path, "w"); owned_ = true; }
    buf_.reserve(kFlush + 4096);
  }
  ~BufOut() { close(); }
  BufOut(const BufOut &) = delete;
  BufOut & operator=(const BufOut &) = delete;
  bool ok() const { return fp_ != nullptr; }
  void put(char c) { buf_.push_back(c); maybe_flush(); }
  void write(const char * p, size_t n) {
    if (!memory_only_ && n >= kDirect) { flush(); std::fwrite(p, 1, n, fp_); return; }   // big pieces go straight out
    buf_.append(p, n);
    maybe_flush();
  }
  void str(const char * s) { write(s, std::strlen(s)); }
  void u64(uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v != 0);
    while (n > 0) { buf_.push_back(tmp[--n]); }
    maybe_flush();
  }
  void fixed1(double v) {                 // "%.1f"
    char tmp[64];
    const int n = std::snprintf(tmp, sizeof(tmp), "%.1f", v);
    write(tmp, (size_t)n);
  }
  void close() {
    if (fp_ == nullptr) { return; }
    flush();
    if (owned_) { std::fclose(fp_); } else { std::fflush(fp_); }
    fp_ = nullptr;
  }

 private:
  static constexpr size_t kFlush = 1 << 20;
  static constexpr size_t kDirect = 64 << 10;
  void maybe_flush() { if (!memory_only_ && buf_.size() >= kFlush) { flush(); } }
  void flush() { if (!buf_.empty()) { std::fwrite(buf_.data(), 1, buf_.size(), fp_); buf_.clear(); } }
  FILE * fp_ = nullptr;
  bool owned_ = false;
  bool memory_only_ = false;
  std::string buf_;
};

// Big outputs are formatted by several threads — `format(sink, begin, end)` writes items [begin, end) into a memory
// sink — while the calling thread emits the finished pieces in order, so the bytes are those of the single-threaded
// loop and the copy into the page cache (one thread's work: writes to one file serialise in the kernel) runs beside
// the formatting instead of after it.  More pieces than threads, claimed from a counter: items differ in cost (swarm
// sizes; an alignment per member for -u).
inline int swa_host_team() { return std::max(1, std::min<int>(omp_get_max_threads(), (int)std::min(swa_host_cpus(), 32u))); }

// pieces [bounds[p], bounds[p + 1]) of the items, formatted by the team, written in order by the calling thread
template <class F>
void swa_run_pieces(BufOut & o, const std::vector<size_t> & bounds, int threads, F && format) {
  const size_t npieces = bounds.size() - 1;
  std::vector<std::string> pieces(npieces);
  std::vector<std::atomic<int>> ready(npieces);
  for (auto & r : ready) { r.store(0, std::memory_order_relaxed); }
  std::atomic<size_t> next{0};
#pragma omp parallel num_threads(threads)
  {
    if (omp_get_thread_num() == 0) {
      // The team the runtime GRANTED may be one thread (OMP_THREAD_LIMIT, a nested region, omp dynamic) whatever was asked
      // for: then nobody else formats, and a writer that only waited would wait for ever — it claims pieces
      // itself.
      const bool alone = omp_get_num_threads() < 2;
      for (size_t p = 0; p < npieces; ++p) {
        while (ready[p].load(std::memory_order_acquire) == 0) {
          if (!alone) { std::this_thread::yield(); continue; }
          const size_t q = next.fetch_add(1, std::memory_order_relaxed);
          if (q >= npieces) { continue; }
          BufOut sink;
          format(sink, bounds[q], bounds[q + 1]);
          pieces[q] = sink.take();
          ready[q].store(1, std::memory_order_release);
        }
        o.write(pieces[p].data(), pieces[p].size());
        std::string().swap(pieces[p]);
      }
    } else {
      for (;;) {
        const size_t p = next.fetch_add(1, std::memory_order_relaxed);
        if (p >= npieces) { break; }
        BufOut sink;
        format(sink, bounds[p], bounds[p + 1]);
        pieces[p] = sink.take();
        ready[p].store(1, std::memory_order_release);
      }
    }
  }
}

template <class F>
void swa_format_in_pieces(BufOut & o, size_t items, bool parallel, F && format) {
  const int threads = swa_host_team();
  if (!parallel || threads < 3 || items < 2) { format(o, (size_t)0, items); return; }
  const size_t npieces = std::min<size_t>(items, (size_t)threads * 8);
  std::vector<size_t> bounds(npieces + 1);
  for (size_t p = 0; p <= npieces; ++p) { bounds[p] = items * p / npieces; }
  swa_run_pieces(o, bounds, threads, format);
}

// The same with pieces of equal WEIGHT — weight(k) = what item k costs to format, e.g. the members a swarm prints.  Swarms
// come largest first: pieces of equal item counts put a tenth of a plain run's members — and, with --fastidious, where the
// heavy swarms in front print the millions of light ones grafted onto them, most of the file — into the first piece, which
// one thread then formats while the writer and the other threads wait for it.
template <class W, class F>
void swa_format_in_weighted_pieces(BufOut & o, size_t items, bool parallel, W && weight, F && format) {
  const int threads = swa_host_team();
  if (!parallel || threads < 3 || items < 2) { format(o, (size_t)0, items); return; }
  const size_t npieces = std::min<size_t>(items, (size_t)threads * 8);
  // sums over chunks of the items by all threads, then every boundary by a search over the chunks and a walk inside one
  const size_t nchunks = std::min<size_t>(items, 4096);
  std::vector<uint64_t> upto(nchunks + 1, 0);
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int64_t c = 0; c < (int64_t)nchunks; ++c) {
    uint64_t sum = 0;
    for (size_t k = items * (size_t)c / nchunks; k < items * ((size_t)c + 1) / nchunks; ++k) { sum += weight(k); }
    upto[(size_t)c + 1] = sum;
  }
  for (size_t c = 0; c < nchunks; ++c) { upto[c + 1] += upto[c]; }
  const uint64_t total = upto[nchunks];
  std::vector<size_t> bounds(npieces + 1, items);
  bounds[0] = 0;
  for (size_t p = 1; p < npieces; ++p) {
    if (total == 0) { bounds[p] = items * p / npieces; continue; }
    const uint64_t target = total / npieces * p + total % npieces * p / npieces;      // total * p / npieces without the overflow
    const size_t c = (size_t)(std::upper_bound(upto.begin(), upto.end(), target) - upto.begin()) - 1;   // upto[c] <= target < upto[c + 1]
    if (c >= nchunks) { bounds[p] = items; continue; }
    uint64_t acc = upto[c];
    size_t k = items * c / nchunks;
    const size_t chunk_end = items * (c + 1) / nchunks;
    while (k < chunk_end && acc + weight(k) <= target) { acc += weight(k); ++k; }
    bounds[p] = std::max(k, bounds[p - 1]);
  }
  swa_run_pieces(o, bounds, threads, format);
}

namespace swa_out {

inline const char * hdr(const swa_hostdb * db, uint32_t i) { return db->hdr(i); }
inline uint32_t hdrlen(const swa_hostdb * db, uint32_t i) { return db->ent[i]->hdr_len(); }

// fprint_id (src/db.cc:946-968)
inline void id(BufOut & o, const swa_hostdb * db, uint32_t i, bool usearch, int64_t append_abundance) {
  o.write(hdr(db, i), hdrlen(db, i));
  if (append_abundance != 0 && db->ent[i]->ab_start == db->ent[i]->ab_end) {
    if (usearch) { o.str(";size="); o.u64(db->abundance[i]); o.put(';'); }
    else { o.put('_'); o.u64(db->abundance[i]); }
  }
}

// fprint_id_noabundance (src/db.cc:971-999)
inline void id_noabundance(BufOut & o, const swa_hostdb * db, uint32_t i, bool usearch) {
  const int s = db->ent[i]->ab_start, e = db->ent[i]->ab_end, len = (int)hdrlen(db, i);
  if (s < e) {
    o.write(hdr(db, i), (size_t)s);
    if (usearch) {
      if (s > 0 && e < len) { o.put(';'); }
      o.write(hdr(db, i) + e, (size_t)(len - e));
    }
  } else {
    o.write(hdr(db, i), (size_t)len);
  }
}

// fprint_id_with_new_abundance (src/db.cc:1002-1026)
inline void id_new_abundance(BufOut & o, const swa_hostdb * db, uint32_t i, uint64_t abundance, bool usearch) {
  o.write(hdr(db, i), (size_t)db->ent[i]->ab_start);
  if (usearch) {
    if (db->ent[i]->ab_start > 0) { o.put(';'); }
    o.str("size=");
    o.u64(abundance);
    o.put(';');
    o.write(hdr(db, i) + db->ent[i]->ab_end, (size_t)((int)hdrlen(db, i) - db->ent[i]->ab_end));
  } else {
    o.put('_');
    o.u64(abundance);
  }
}

// db_fprintseq (src/db.cc:925-943)
inline void sequence(BufOut & o, const swa_hostdb * db, uint32_t i, std::string & scratch) {
  const uint64_t * w = db->words(i);
  const uint32_t len = db->seqlen[i];
  scratch.resize(len);
  for (uint32_t p = 0; p < len; ++p) { scratch[p] = "ACGT"[(w[p >> 5] >> ((p & 31u) << 1)) & 3u]; }
  o.write(scratch.data(), len);
  o.put('\n');
}

}  // namespace swa_out
