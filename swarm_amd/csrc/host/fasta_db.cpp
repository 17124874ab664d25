// fasta_db.cpp — host side of seam L2: FASTA -> packed amplicon database in the
// reference's db order.  Behavioural mirror of db_read (src/db.cc:432-803): same
// accepted alphabet, same header / abundance rules, same error texts, same sort
// (abundance descending, then header bytes ascending, src/db.cc:388-413) — with a
// different shape: the input is memory-mapped, cut at record boundaries and parsed by
// all host cores in parallel into per-thread SoA pieces; abundance parsing and the
// identifier-uniqueness check run in the same parallel pass (lock-free open addressing);
// the sort is a parallel merge sort on (abundance, 8-byte header prefix) keys that falls
// back to strcmp only on ties; and what the parser wrote STAYS where it is, in file order —
// db order is an array of pointers into it, and the packed words are put in db order by the
// GPU as they arrive (swa_db_upload_unordered) —, instead of the reference's single-threaded
// getline loop over an interleaved header/sequence blob.
#include "hostdb.h"
#include "pool.h"

#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>


#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Piece {            // what one thread parsed: the data (it becomes the database's storage) and the parse's own notes
  swa_piece data;
  uint64_t lines = 0;     // newline-terminated lines consumed
  uint64_t nucleotides = 0;
  uint32_t longest = 0, longest_header = 0;
  std::string error;      // first error of the piece (line numbers still local)
  uint32_t error_line = 0;
  int error_kind = 0;     // 0 none, 1 message carries a \x01 where the absolute line number goes, 2 final
  // second-phase errors (abundance value, empty identifier): the reference finds them only after it has read the whole
  // file (db.cc:676-694), so an illegal character anywhere comes first; the first one of the piece is kept
  std::string late_error;
  uint32_t late_line = 0;
  int late_kind = 0;      // as error_kind
  uint64_t late_entry = 0;   // index of the entry inside the piece
  uint64_t missing = 0;   // entries without abundance annotation
  uint32_t missing_line = 0;
  std::string missing_hdr;
};

struct Input {
  const char * data = nullptr;
  size_t size = 0;
  bool mapped = false;
  std::vector<char> owned;
  // A regular file is not mapped: the parser threads pread() it through buffers of a few megabytes each.  Mapped, the
  // 1.6 GB of a 10 M-amplicon file were 400 000 page-table entries this process had to take apart again at exit (or
  // whenever it unmapped them): a quarter of a second in the runs where the kernel does not do that on its own time
  // (profiles/r05/NOTES.md 8-9).  SWARM_AMD_INPUT=mmap: the mapping, for comparison.
  int fd = -1;
  ~Input() { if (fd >= 0) { ::close(fd); } }
};

bool load_input(const char * path, Input & in, std::string & err) {
  const bool is_stdin = (std::strcmp(path, "-") == 0);
  if (!is_stdin) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) {
      err = std::string("\nError: Unable to open input data file (") + path + ").\n";   // db.cc:458-461
      return false;
    }
    struct stat st{};
    const char * input_env = std::getenv("SWARM_AMD_INPUT");
    const bool want_map = input_env != nullptr && std::strcmp(input_env, "mmap") == 0;
    if (!want_map && ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      in.fd = fd;
      in.size = (size_t)st.st_size;
      (void)::posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
      return true;
    }
    if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      void * p = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (p != MAP_FAILED) {
        (void)::madvise(p, (size_t)st.st_size, MADV_WILLNEED);
        in.data = static_cast<const char *>(p);
        in.size = (size_t)st.st_size;
        in.mapped = true;
        ::close(fd);
        return true;
      }
    }
    // pipes, empty files, mmap failure: plain read
    size_t used = 0;
    in.owned.resize(1 << 20);
    for (;;) {
      if (used == in.owned.size()) { in.owned.resize(in.owned.size() * 2); }
      const ssize_t got = ::read(fd, in.owned.data() + used, in.owned.size() - used);
      if (got <= 0) { break; }
      used += (size_t)got;
    }
    ::close(fd);
    in.owned.resize(used);
  } else {
    size_t used = 0;
    in.owned.resize(1 << 20);
    for (;;) {
      if (used == in.owned.size()) { in.owned.resize(in.owned.size() * 2); }
      const size_t got = std::fread(in.owned.data() + used, 1, in.owned.size() - used, stdin);
      if (got == 0) { break; }
      used += got;
    }
    in.owned.resize(used);
  }
  in.data = in.owned.data();
  in.size = in.owned.size();
  return true;
}

// (_)([0-9]+)$  — db.cc:161-211
bool swarm_abundance(const char * h, uint32_t len, int32_t & start, int32_t & end, int64_t & number) {
  int64_t us = -1;
  for (int64_t i = (int64_t)len - 1; i >= 0; --i) { if (h[i] == '_') { us = i; break; } }
  if (us < 0) { return false; }
  const uint32_t digits = len - (uint32_t)us - 1;
  if (digits > 20) { return false; }
  for (uint32_t i = 0; i < digits; ++i) { if (h[us + 1 + i] < '0' || h[us + 1 + i] > '9') { return false; } }
  start = (int32_t)us;
  end = (int32_t)len;
  if (digits <= 18) {                // (no overflow possible; "_" alone gives 0 like atol(""))
    int64_t v = 0;
    for (uint32_t i = 0; i < digits; ++i) { v = v * 10 + (h[us + 1 + i] - '0'); }
    number = v;
    return true;
  }
  char tmp[24];
  std::memcpy(tmp, h + us + 1, digits);
  tmp[digits] = 0;
  number = std::atol(tmp);
  return true;
}

// (^|;)size=([0-9]+)(;|$)  — db.cc:214-283
bool usearch_abundance(const char * h, uint32_t len, int32_t & start, int32_t & end, int64_t & number) {
  const int64_t hlen = len;
  const int64_t alen = 5;
  int64_t pos = 0;
  while (pos + alen < hlen) {
    int64_t found = -1;
    for (int64_t i = pos; i + alen <= hlen; ++i) { if (std::memcmp(h + i, "size=", 5) == 0) { found = i; break; } }
    if (found < 0) { break; }
    pos = found;
    if (pos > 0 && h[pos - 1] != ';') { pos += alen + 1; continue; }
    int64_t digits = 0;
    while (pos + alen + digits < hlen && h[pos + alen + digits] >= '0' && h[pos + alen + digits] <= '9') { ++digits; }
    if (digits == 0) { pos += alen + 1; continue; }
    if (pos + alen + digits < hlen && h[pos + alen + digits] != ';') { pos += alen + digits + 2; continue; }
    start = pos > 0 ? (int32_t)(pos - 1) : 0;
    end = (int32_t)std::min(pos + alen + digits + 1, hlen);
    std::string num(h + pos + alen, (size_t)digits);
    number = std::atol(num.c_str());
    return true;
  }
  return false;
}

inline uint64_t bytes_hash(const char * s, uint32_t len) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint32_t i = 0; i < len; ++i) { h = (h ^ (unsigned char)s[i]) * 0x100000001b3ull; }
  return h ^ (h >> 29);
}

// 32 nucleotides at q -> 64 bits (2 bits each, first nucleotide lowest: the layout of src/db.cc:541-628), with AVX2;
// false when any of the 32 bytes is not one of ACGTU / acgtu (the caller's byte loop then says what it is).  Upper-cased
// by clearing bit 5; the low nibble selects the code (A 1 -> 0, C 3 -> 1, G 7 -> 2, T 4 -> 3, U 5 -> 3) and the high nibble
// the letter must have (4 for A C G, 5 for T U); pairs, quads and octets of codes are folded with multiply-adds.
__attribute__((target("avx2"))) bool pack32_avx2(const char * q, uint64_t * out) {
  const __m256i raw = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q));
  const __m256i up = _mm256_and_si256(raw, _mm256_set1_epi8((char)0xDF));
  const __m256i lo = _mm256_and_si256(up, _mm256_set1_epi8(0x0F));
  const __m256i hi = _mm256_and_si256(_mm256_srli_epi16(up, 4), _mm256_set1_epi8(0x0F));
  const __m256i code_of = _mm256_setr_epi8(0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i high_of = _mm256_setr_epi8(-1, 4, -1, 4, 5, 5, -1, 4, -1, -1, -1, -1, -1, -1, -1, -1, -1, 4, -1, 4, 5, 5, -1, 4, -1, -1, -1, -1, -1, -1, -1, -1);
  const __m256i ok = _mm256_cmpeq_epi8(_mm256_shuffle_epi8(high_of, lo), hi);
  if (_mm256_movemask_epi8(ok) != -1) { return false; }
  const __m256i codes = _mm256_shuffle_epi8(code_of, lo);                                        // 32 x 2 bits, one per byte
  const __m256i pairs = _mm256_maddubs_epi16(codes, _mm256_set1_epi16(0x0401));                  // 16 x 4 bits in 16-bit lanes
  const __m256i quads = _mm256_madd_epi16(pairs, _mm256_set1_epi32(0x00100001));                 // 8 x 8 bits in 32-bit lanes
  const __m256i bytes = _mm256_shuffle_epi8(quads, _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                                                   0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
  const uint64_t low = (uint32_t)_mm256_extract_epi32(bytes, 0), high = (uint32_t)_mm256_extract_epi32(bytes, 4);
  *out = low | (high << 32);
  return true;
}

// the same for the last r < 32 nucleotides of a line: 32 bytes are read at q (the caller knows they exist), the first r
// must be nucleotides, the rest — the line's end and whatever follows — is ignored; *out holds 2 r bits
__attribute__((target("avx2"))) bool pack_tail_avx2(const char * q, uint32_t r, uint64_t * out) {
  const __m256i raw = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q));
  const __m256i up = _mm256_and_si256(raw, _mm256_set1_epi8((char)0xDF));
  const __m256i lo = _mm256_and_si256(up, _mm256_set1_epi8(0x0F));
  const __m256i hi = _mm256_and_si256(_mm256_srli_epi16(up, 4), _mm256_set1_epi8(0x0F));
  const __m256i code_of = _mm256_setr_epi8(0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i high_of = _mm256_setr_epi8(-1, 4, -1, 4, 5, 5, -1, 4, -1, -1, -1, -1, -1, -1, -1, -1, -1, 4, -1, 4, 5, 5, -1, 4, -1, -1, -1, -1, -1, -1, -1, -1);
  const __m256i iota = _mm256_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
  const __m256i inside = _mm256_cmpgt_epi8(_mm256_set1_epi8((char)r), iota);                     // bytes [0, r)
  const __m256i ok = _mm256_cmpeq_epi8(_mm256_shuffle_epi8(high_of, lo), hi);
  if (_mm256_movemask_epi8(_mm256_or_si256(ok, _mm256_andnot_si256(inside, _mm256_set1_epi8(-1)))) != -1) { return false; }
  const __m256i codes = _mm256_and_si256(_mm256_shuffle_epi8(code_of, lo), inside);
  const __m256i pairs = _mm256_maddubs_epi16(codes, _mm256_set1_epi16(0x0401));
  const __m256i quads = _mm256_madd_epi16(pairs, _mm256_set1_epi32(0x00100001));
  const __m256i bytes = _mm256_shuffle_epi8(quads, _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                                                   0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
  const uint64_t low = (uint32_t)_mm256_extract_epi32(bytes, 0), high = (uint32_t)_mm256_extract_epi32(bytes, 4);
  *out = low | (high << 32);
  return true;
}

// parse records of [begin, end) — begin points at a '>' that starts a line (or at the file start)
// (may be called again for the next stretch of the same piece — the streamed reader hands it whole records a buffer at a
// time: line numbers, entries and pools continue; reserve_span = bytes of text the piece will hold in all, 0 = sized already)
void parse_piece(const char * begin, const char * end, const int8_t * map, bool usearch, int64_t append_abundance,
                 uint32_t piece_no, Piece & out, size_t reserve_span) {
  const char * p = begin;
  uint32_t lineno = (uint32_t)out.lines + 1u;
  static const bool have_avx2 = __builtin_cpu_supports("avx2") && std::getenv("SWARM_AMD_NO_AVX2") == nullptr;
  auto line_end = [&](const char * q) { const void * nl = std::memchr(q, '\n', (size_t)(end - q)); return nl ? (const char *)nl : end; };
  auto & entries = out.data.entries;
  auto & words = out.data.words;
  auto & hdr_pool = out.data.hdr_pool;
  if (reserve_span != 0) {
    entries.reserve(reserve_span / 160 + 16);
    words.reserve(reserve_span / 28 + 16);
    hdr_pool.reserve(reserve_span / 12 + 16);
  }
  auto fail = [&](const std::string & msg, uint32_t line, int kind) { out.error = msg; out.error_line = line; out.error_kind = kind; };
  while (p < end) {
    if (*p != '>') { fail("\nError: Illegal header line in fasta file.\n", lineno, 2); break; }   // db.cc:492-494
    const char * le = line_end(p);
    const char * h = p + 1;
    uint32_t hlen = 0;
    while (h + hlen < le && h[hlen] != ' ' && h[hlen] != '\r' && h[hlen] != 0) { ++hlen; }      // db.cc:498-499
    if (hlen > 16777215u) { fail("\nError: Headers longer than 16,777,215 symbols are not supported.\n", lineno, 2); break; }
    swa_entry e{};
    e.hdr_off = hdr_pool.size();
    e.hdr_len_piece = hlen | (piece_no << 24);
    const uint32_t entry_line = lineno;
    hdr_pool.insert(hdr_pool.end(), h, h + hlen);
    hdr_pool.push_back('\0');
    p = (le < end) ? le + 1 : end;
    ++lineno;

    e.word_off = words.size();
    uint64_t acc = 0;
    uint32_t fill = 0, len = 0;
    bool bad = false;
    while (p < end && *p != '>') {
      le = line_end(p);
      const char * q = p;
      // eight nucleotides per turn while the line holds nothing else (the eight table lookups are
      // independent; 1.8x the byte-at-a-time loop); anything unusual drops to the loop below
      if (have_avx2) {                                      // 32 nucleotides per turn: a whole word
        uint64_t chunk;
        while (q + 32 <= le && pack32_avx2(q, &chunk)) {
          if (fill == 0u) { words.push_back(chunk); }
          else { words.push_back(acc | (chunk << (2u * fill))); acc = chunk >> (64u - 2u * fill); }
          len += 32u;
          q += 32;
        }
        const uint32_t rest = (uint32_t)(le - q);           // the line's last nucleotides in one turn, too
        if (rest != 0u && rest < 32u && q + 32 <= end && pack_tail_avx2(q, rest, &chunk)) {
          acc |= chunk << (2u * fill);
          if (fill + rest >= 32u) {
            words.push_back(acc);
            acc = chunk >> (2u * (32u - fill));             // (fill >= 1 here: rest < 32)
          }
          fill = (fill + rest) & 31u;
          len += rest;
          q = le;
        }
      }
      while (q + 8 <= le) {
        const uint32_t c0 = (uint8_t)map[(uint8_t)q[0]], c1 = (uint8_t)map[(uint8_t)q[1]], c2 = (uint8_t)map[(uint8_t)q[2]],
                       c3 = (uint8_t)map[(uint8_t)q[3]], c4 = (uint8_t)map[(uint8_t)q[4]], c5 = (uint8_t)map[(uint8_t)q[5]],
                       c6 = (uint8_t)map[(uint8_t)q[6]], c7 = (uint8_t)map[(uint8_t)q[7]];
        if (((c0 | c1 | c2 | c3 | c4 | c5 | c6 | c7) & 0x80u) != 0u) { break; }
        const uint64_t bits = c0 | (c1 << 2) | (c2 << 4) | (c3 << 6) | (c4 << 8) | (c5 << 10) | (c6 << 12) | (c7 << 14);
        acc |= bits << (2u * fill);
        if (fill + 8u >= 32u) {
          words.push_back(acc);
          const uint32_t used = 32u - fill;                    // nucleotides of `bits` that went into the full word
          acc = used < 8u ? bits >> (2u * used) : 0;
          fill = fill + 8u - 32u;
        } else {
          fill += 8u;
        }
        len += 8u;
        q += 8;
      }
      for (; q < le; ++q) {
        const unsigned char ch = (unsigned char)*q;
        const int8_t code = map[ch];
        if (code >= 0) {
          acc |= (uint64_t)code << (2u * fill);
          ++len;
          if (++fill == 32u) { words.push_back(acc); acc = 0; fill = 0; }
        } else if (ch == 0) {
          break;                                 // the reference's line scan stops at a NUL
        } else if (ch != '\n' && ch != '\r') {
          char msg[160];
          if (ch >= 32 && ch <= 126) { std::snprintf(msg, sizeof(msg), "\nError: Illegal character '%c' in sequence on line \x01.\n", ch); }
          else { std::snprintf(msg, sizeof(msg), "\nError: Illegal character (ascii no %d) in sequence on line \x01.\n", (int)ch); }
          fail(msg, lineno, 1);                                                                  // db.cc:575-590
          bad = true;
          break;
        }
      }
      if (bad) { break; }
      if (len > 67108861u) { fail("\nError: Sequences longer than 67,108,861 symbols are not supported.\n", lineno, 2); bad = true; break; }
      p = (le < end) ? le + 1 : end;
      ++lineno;
    }
    if (bad) { break; }
    if (len == 0) { fail("\nError: Empty sequence found on line \x01.\n", lineno - 1, 1); break; }   // db.cc:608-611
    if (fill > 0) { words.push_back(acc); }
    e.seqlen = len;
    out.nucleotides += len;
    out.longest = std::max(out.longest, len);
    out.longest_header = std::max(out.longest_header, hlen);

    // abundance annotation (db.cc:286-343)
    const char * hdr = hdr_pool.data() + e.hdr_off;
    int32_t s = 0, t = 0;
    int64_t number = 0, abundance = 0;
    const bool found = usearch ? usearch_abundance(hdr, hlen, s, t, number) : swarm_abundance(hdr, hlen, s, t, number);
    auto fail_late = [&](const std::string & msg, int kind) {
      if (out.late_kind == 0) { out.late_error = msg; out.late_line = entry_line; out.late_kind = kind; out.late_entry = entries.size(); }
    };
    bool entry_failed = false;
    if (found) {
      if (number <= 0) {
        fail_late(std::string("\nError: Illegal abundance value on line \x01:\n") + hdr + "\nAbundance values should be positive integers.\n", 1);
        entry_failed = true;
        number = 1;                              // (reading goes on: a first-phase error further down outranks this one)
      }
      abundance = number;
    }
    if (abundance == 0) {
      s = (int32_t)hlen;
      t = s;
      if (append_abundance != 0) { abundance = append_abundance; }
      else if (++out.missing == 1) { out.missing_line = entry_line; out.missing_hdr = hdr; }
    }
    e.abundance = (uint64_t)abundance;
    e.ab_start = s;
    e.ab_end = t;
    if (!entry_failed && e.ab_start == 0 && e.ab_end == (int32_t)hlen) { fail_late("\nError: Empty sequence identifier.\n", 2); }
    entries.push_back(e);
  }
  out.lines = lineno - 1;
}

// bytes [pos, pos + len) of the file into dst; false when the file ends (or fails) before that
bool pread_all(int fd, char * dst, size_t len, size_t pos) {
  while (len != 0) {
    const ssize_t got = ::pread(fd, dst, len, (off_t)pos);
    if (got <= 0) { return false; }
    dst += got; pos += (size_t)got; len -= (size_t)got;
  }
  return true;
}

// the first record start behind `from`: the smallest p > from with text[p - 1] == '\n' and text[p] == '>' (or the file's size)
size_t next_record_start(int fd, size_t from, size_t size) {
  std::vector<char> window(64 << 10);
  bool after_newline = false;                                  // (text[from - 1] does not count: the cut lies BEHIND from)
  for (size_t pos = from; pos < size;) {
    const size_t len = std::min(window.size(), size - pos);
    if (!pread_all(fd, window.data(), len, pos)) { return size; }
    size_t k = 0;
    if (after_newline && window[0] == '>' && pos > from) { return pos; }
    while (k < len) {
      const void * nl = std::memchr(window.data() + k, '\n', len - k);
      if (nl == nullptr) { k = len; after_newline = false; break; }
      k = (size_t)((const char *)nl - window.data()) + 1;
      if (k < len) { if (window[k] == '>') { return pos + k; } }
      else { after_newline = true; }
    }
    pos += len;
  }
  return size;
}

// bytes [lo, hi) of the file — whole records: lo and hi are record starts (or the file's ends) — through one buffer
void parse_piece_streamed(int fd, size_t lo, size_t hi, const int8_t * map, bool usearch, int64_t append_abundance, uint32_t piece_no, Piece & out) {
  const char * chunk_env = std::getenv("SWARM_AMD_READ_CHUNK_KB");     // (tests: buffers smaller than a record)
  const size_t chunk = chunk_env != nullptr ? std::max<size_t>(1, (size_t)std::atol(chunk_env)) << 10 : size_t(4) << 20;
  std::vector<char> buf(std::min(chunk, hi - lo) + 64);        // (+ 64: the packers may look 32 bytes ahead, never behind `end`)
  size_t have = 0, off = lo;
  bool first = true;
  while (off < hi || have != 0) {
    const size_t want = std::min(buf.size() - 64 - have, hi - off);
    if (want != 0) {
      if (!pread_all(fd, buf.data() + have, want, off)) { hi = off; continue; }     // (a file that shrank under the reader ends here)
      have += want; off += want;
    }
    size_t b = have;                                           // records [0, b) of the buffer are whole
    if (off < hi) {
      b = 0;
      for (size_t k = have; k > 1;) {                          // the last "\n>" of the buffer
        const void * nl = ::memrchr(buf.data(), '\n', k - 1);  // a newline with a byte behind it inside [0, have)
        if (nl == nullptr) { break; }
        k = (size_t)((const char *)nl - buf.data()) + 1;
        if (buf[k] == '>') { b = k; break; }
        --k;
      }
      if (b == 0) { buf.resize((buf.size() - 64) * 2 + 64); continue; }   // one record fills the buffer: a larger one
    }
    parse_piece(buf.data(), buf.data() + b, map, usearch, append_abundance, piece_no, out, first ? hi - lo : 0);
    first = false;
    if (out.error_kind != 0) { return; }
    std::memmove(buf.data(), buf.data() + b, have - b);
    have -= b;
  }
}

unsigned worker_count(size_t bytes) {
  const char * env = std::getenv("SWARM_AMD_HOST_THREADS");
  unsigned t = env != nullptr ? (unsigned)std::atoi(env) : swa_host_cpus();
  if (t < 1) { t = 1; }
  if (t > 64) { t = 64; }
  const size_t by_size = bytes / (4u << 20) + 1;            // at least 4 MB of text per thread
  return (unsigned)std::min<size_t>(t, by_size);
}

// fn(t) for t in [0, tasks) on the process's worker threads (pool.h)
template <typename F>
void run_parallel(unsigned tasks, F && fn) { swa_pool::get().run(tasks, fn); }


struct PhaseTimer {                       // SWARM_AMD_DB_TIMING=1 prints the phase times to stderr: wall, and CPU seconds of the whole process
  bool on = std::getenv("SWARM_AMD_DB_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  double cpu = cpu_now();
  static double cpu_now() {
    struct rusage u{};
    getrusage(RUSAGE_SELF, &u);
    return (double)u.ru_utime.tv_sec + (double)u.ru_stime.tv_sec + 1e-6 * ((double)u.ru_utime.tv_usec + (double)u.ru_stime.tv_usec);
  }
  void lap(const char * what) {
    if (!on) { return; }
    const auto now = std::chrono::steady_clock::now();
    const double c = cpu_now();
    std::fprintf(stderr, "[hostdb] %-28s %8.3f ms   cpu %6.3f s\n", what, std::chrono::duration<double, std::milli>(now - t).count(), c - cpu);
    t = now;
    cpu = c;
  }
};

// sorts a[0, n) by `less` (a strict weak order) on `threads` threads; tmp[0, n) is scratch; the result is in a
template <class Rec, class Less>
void parallel_sample_sort(Rec * a, Rec * tmp, uint16_t * where, uint64_t n, unsigned threads, Less less, PhaseTimer * timer = nullptr) {
  if (threads <= 1 || n < 100000) { std::sort(a, a + n, less); return; }
  static const unsigned env_buckets = [] { const char * e = std::getenv("SWARM_AMD_SORT_BUCKETS"); return e != nullptr ? (unsigned)std::atoi(e) : 0u; }();
  const unsigned buckets = env_buckets >= 2u ? std::min(env_buckets, 4096u) : std::min<unsigned>(threads * 8u, 1024u);
  constexpr uint64_t kOver = 64;                            // sampled records per bucket
  const uint64_t nsample = (uint64_t)buckets * kOver;
  std::vector<Rec> sample(nsample);
  for (uint64_t i = 0; i < nsample; ++i) { sample[i] = a[(n - 1) * i / (nsample - 1)]; }
  std::sort(sample.begin(), sample.end(), less);
  std::vector<Rec> split(buckets - 1);
  for (unsigned b = 1; b < buckets; ++b) { split[b - 1] = sample[(uint64_t)b * kOver]; }
  std::vector<uint64_t> place((size_t)threads * buckets, 0);
  if (timer != nullptr) { timer->lap("  sort: sample + splitters"); }
  run_parallel(threads, [&](unsigned t) {
    uint64_t * c = &place[(size_t)t * buckets];
    for (uint64_t i = n * t / threads; i < n * (t + 1) / threads; ++i) {
      unsigned lo = 0, hi = buckets - 1;                    // bucket = number of splitters not above the record
      while (lo < hi) {
        const unsigned mid = (lo + hi) / 2;
        if (less(a[i], split[mid])) { hi = mid; } else { lo = mid + 1; }
      }
      where[i] = (uint16_t)lo;
      ++c[lo];
    }
  });
  std::vector<uint64_t> start(buckets + 1, 0);
  uint64_t at = 0;
  for (unsigned b = 0; b < buckets; ++b) {
    start[b] = at;
    for (unsigned t = 0; t < threads; ++t) { const uint64_t c = place[(size_t)t * buckets + b]; place[(size_t)t * buckets + b] = at; at += c; }
  }
  start[buckets] = at;
  if (timer != nullptr) { timer->lap("  sort: buckets found"); }
  run_parallel(threads, [&](unsigned t) {
    uint64_t * c = &place[(size_t)t * buckets];
    for (uint64_t i = n * t / threads; i < n * (t + 1) / threads; ++i) { tmp[c[where[i]]++] = a[i]; }
  });
  if (timer != nullptr) { timer->lap("  sort: records filed"); }
  std::atomic<unsigned> next{0};
  run_parallel(threads, [&](unsigned) {
    for (;;) {
      const unsigned b = next.fetch_add(1);
      if (b >= buckets) { break; }
      std::sort(tmp + start[b], tmp + start[b + 1], less);
      std::copy(tmp + start[b], tmp + start[b + 1], a + start[b]);
    }
  });
  if (timer != nullptr) { timer->lap("  sort: buckets sorted"); }
}

// ---- the db-order sort by radix (the usual case)
// A sort record: the abundance (saturating at 32 bits), the first 8 header bytes big endian, the entry's number.  Its
// integer key K = (abundance descending, key8 ascending) decides the db order wherever it differs; records with equal K
// (same abundance, identifiers that share their first 8 bytes) are put right afterwards by the full comparison.
struct SortRec { uint64_t key8; uint32_t abundance, entry; };
static_assert(sizeof(SortRec) == 16, "the scratch block is laid out for 16-byte sort records");

inline bool k_before(const SortRec & a, const SortRec & b) { return a.abundance != b.abundance ? a.abundance > b.abundance : a.key8 < b.key8; }
inline bool k_equal(const SortRec & a, const SortRec & b) { return a.abundance == b.abundance && a.key8 == b.key8; }

// r[0, n) by K with byte-wise LSD passes between r and s (meant for pieces that fit the core's cache); digits on which
// all records agree — the abundance's upper bytes, a constant first letter — cost no pass.  The result is in r.
void lsd_sort_by_k(SortRec * r, SortRec * s, size_t n) {
  if (n < 48) {
    for (size_t i = 1; i < n; ++i) {                          // insertion sort
      const SortRec x = r[i];
      size_t j = i;
      while (j > 0 && k_before(x, r[j - 1])) { r[j] = r[j - 1]; --j; }
      r[j] = x;
    }
    return;
  }
  uint32_t hist[12][256];
  std::memset(hist, 0, sizeof(hist));
  for (size_t i = 0; i < n; ++i) {
    const uint64_t k = r[i].key8;
    const uint32_t a = ~r[i].abundance;
    for (int d = 0; d < 8; ++d) { ++hist[d][(k >> (8 * d)) & 255u]; }
    for (int d = 0; d < 4; ++d) { ++hist[8 + d][(a >> (8 * d)) & 255u]; }
  }
  SortRec * src = r, * dst = s;
  for (int d = 0; d < 12; ++d) {
    uint32_t * h = hist[d];
    bool one_bin = false;
    uint32_t at = 0;
    for (int b = 0; b < 256; ++b) { const uint32_t c = h[b]; if (c == n) { one_bin = true; } h[b] = at; at += c; }
    if (one_bin) { continue; }
    if (d < 8) {
      const int shift = 8 * d;
      for (size_t i = 0; i < n; ++i) { dst[h[(src[i].key8 >> shift) & 255u]++] = src[i]; }
    } else {
      const int shift = 8 * (d - 8);
      for (size_t i = 0; i < n; ++i) { dst[h[((~src[i].abundance) >> shift) & 255u]++] = src[i]; }
    }
    std::swap(src, dst);
  }
  if (src != r) { std::memcpy(r, src, n * sizeof(SortRec)); }
}

// a[0, n) into db order on `threads` threads: splitters from a regular sample, every thread files its share of the
// records under them (K alone: records with equal K share a bucket), every bucket — about 8 K records up to 16 M amplicons,
// the two copies fit a core's L2 — is sorted by LSD passes and its runs of equal K by the identifiers' next bytes.  tmp[0, n) and where[0, n) are
// scratch; the result is in a.  false (nothing moved): the sample says that K decides too little — identifiers with a
// long common prefix — and the caller sorts by comparisons.  (No record's abundance may be saturated.)
// `deeper`: what settles records with equal K without a comparison sort over cold identifiers — .ask_entry(rec) / .ask_text(rec)
// prefetch a record's entry and, once that is there, its identifier; .key(rec) = identifier bytes [8, 16) big endian.
template <class Less, class Deeper>
bool parallel_radix_sort(SortRec * a, SortRec * tmp, uint16_t * where, uint64_t n, unsigned threads, Less full_less, Deeper deeper, PhaseTimer * timer) {
  static const unsigned env_buckets = [] { const char * e = std::getenv("SWARM_AMD_SORT_BUCKETS"); return e != nullptr ? (unsigned)std::atoi(e) : 0u; }();
  const unsigned buckets = env_buckets >= 2u ? std::min(env_buckets, 16384u)
                                             : (unsigned)std::min<uint64_t>(std::max<uint64_t>(n / 8192u, (uint64_t)threads * 8u), 2048u);
  // (at most 2048: every thread files into all buckets at once — write streams — and searches log2(buckets) splitters a
  // record.  10^8 amplicons, build container: 12 207 buckets found / filed / sorted in 0.39 / 0.22 / 1.08 s, 2048 in
  // 0.30 / 0.16-0.26 / 1.14 s — no difference worth more streams)
  constexpr uint64_t kOver = 32;                            // sampled records per bucket
  const uint64_t nsample = (uint64_t)buckets * kOver;
  std::vector<SortRec> sample(2 * nsample);
  for (uint64_t i = 0; i < nsample; ++i) { sample[i] = a[(n - 1) * i / (nsample - 1)]; }
  lsd_sort_by_k(sample.data(), sample.data() + nsample, nsample);
  uint64_t ties = 0;
  for (uint64_t i = 1; i < nsample; ++i) { ties += k_equal(sample[i], sample[i - 1]) ? 1u : 0u; }
  if (ties * 4 > nsample) {
    if (timer != nullptr && timer->on) { std::fprintf(stderr, "[hostdb]   sort: %llu of %llu sampled records share their key with a neighbour: by comparisons\n", (unsigned long long)ties, (unsigned long long)nsample); }
    return false;
  }
  // the splitters, padded with "after everything" to one less than a power of two: every search takes the same steps, so
  // four records are searched at once (the chain of dependent loads of one search leaves the core idle otherwise)
  unsigned padded = 1;
  while (padded < buckets) { padded <<= 1; }
  std::vector<SortRec> split(padded, SortRec{~0ull, 0u, 0u});
  for (unsigned b = 1; b < buckets; ++b) { split[b - 1] = sample[(uint64_t)b * kOver]; }
  std::vector<uint64_t> place((size_t)threads * buckets, 0);
  if (timer != nullptr) { timer->lap("  sort: sample + splitters"); }
  run_parallel(threads, [&](unsigned t) {
    uint64_t * c = &place[(size_t)t * buckets];
    const SortRec * sp = split.data();
    // 1 if x comes before s by K, without a branch
    auto before = [](const SortRec & x, const SortRec & s) -> unsigned {
      return (unsigned)(x.abundance > s.abundance) | ((unsigned)(x.abundance == s.abundance) & (unsigned)(x.key8 < s.key8));
    };
    const uint64_t lo_i = n * t / threads, hi_i = n * (t + 1) / threads;
    uint64_t i = lo_i;
    for (; i + 4 <= hi_i; i += 4) {                         // bucket = number of splitters not above the record
      const SortRec x0 = a[i], x1 = a[i + 1], x2 = a[i + 2], x3 = a[i + 3];
      unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0;
      for (unsigned step = padded >> 1; step > 0; step >>= 1) {
        b0 += step & (before(x0, sp[b0 + step - 1]) - 1u);
        b1 += step & (before(x1, sp[b1 + step - 1]) - 1u);
        b2 += step & (before(x2, sp[b2 + step - 1]) - 1u);
        b3 += step & (before(x3, sp[b3 + step - 1]) - 1u);
      }
      b0 = std::min(b0, buckets - 1); b1 = std::min(b1, buckets - 1); b2 = std::min(b2, buckets - 1); b3 = std::min(b3, buckets - 1);
      where[i] = (uint16_t)b0; where[i + 1] = (uint16_t)b1; where[i + 2] = (uint16_t)b2; where[i + 3] = (uint16_t)b3;
      ++c[b0]; ++c[b1]; ++c[b2]; ++c[b3];
    }
    for (; i < hi_i; ++i) {
      const SortRec x = a[i];
      unsigned b = 0;
      for (unsigned step = padded >> 1; step > 0; step >>= 1) { b += step & (before(x, sp[b + step - 1]) - 1u); }
      b = std::min(b, buckets - 1);
      where[i] = (uint16_t)b;
      ++c[b];
    }
  });
  std::vector<uint64_t> start(buckets + 1, 0);
  uint64_t at = 0;
  for (unsigned b = 0; b < buckets; ++b) {
    start[b] = at;
    for (unsigned t = 0; t < threads; ++t) { const uint64_t c = place[(size_t)t * buckets + b]; place[(size_t)t * buckets + b] = at; at += c; }
  }
  start[buckets] = at;
  if (timer != nullptr) { timer->lap("  sort: buckets found"); }
  run_parallel(threads, [&](unsigned t) {
    uint64_t * c = &place[(size_t)t * buckets];
    for (uint64_t i = n * t / threads; i < n * (t + 1) / threads; ++i) { tmp[c[where[i]]++] = a[i]; }
  });
  if (timer != nullptr) { timer->lap("  sort: records filed"); }
  std::atomic<unsigned> next{0};
  run_parallel(threads, [&](unsigned) {
    std::vector<uint32_t> tied, run_begins, run_ends;       // (positions inside a bucket: a bucket holds far fewer than 2^32 records)
    for (;;) {
      const unsigned b = next.fetch_add(1);
      if (b >= buckets) { break; }
      SortRec * const r = tmp + start[b];
      const uint64_t m = start[b + 1] - start[b];
      lsd_sort_by_k(r, a + start[b], m);                    // (a's own stretch of the same size is the second buffer)
      // Runs of equal K: the identifiers decide.  Numbered identifiers of 8 and more characters share their first 8 bytes in
      // runs of ten, a hundred, ...: a comparison sort of every run reads its identifiers cold, two dependent cache misses
      // (entry, text) a record, one after the other (20 M such amplicons: 3.9 CPU-s).  Instead the records of all runs get the
      // NEXT 8 identifier bytes as their key — asked for 16 and 8 records ahead, so the misses overlap — and a run is then
      // sorted on keys that are there; the text itself is compared only where 16 bytes do not decide.
      tied.clear();
      for (uint64_t i = 0; i < m;) {
        uint64_t j = i + 1;
        while (j < m && k_equal(r[j], r[i])) { ++j; }
        if (j - i > 1) { for (uint64_t k = i; k < j; ++k) { tied.push_back((uint32_t)k); } run_ends.push_back((uint32_t)j); run_begins.push_back((uint32_t)i); }
        i = j;
      }
      const size_t nt = tied.size();
      for (size_t t = 0; t < nt; ++t) {
        if (t + 16 < nt) { deeper.ask_entry(r[tied[t + 16]]); }
        if (t + 8 < nt) { deeper.ask_text(r[tied[t + 8]]); }
        r[tied[t]].key8 = deeper.key(r[tied[t]]);             // (K has done its work inside a run: all its records share it)
      }
      for (size_t q = 0; q < run_begins.size(); ++q) {
        SortRec * const lo = r + run_begins[q], * const hi = r + run_ends[q];
        if (hi - lo <= 24) {                                  // insertion sort: the usual run is ten records
          for (SortRec * x = lo + 1; x < hi; ++x) {
            const SortRec v = *x;
            SortRec * y = x;
            while (y > lo && full_less(v, y[-1])) { *y = y[-1]; --y; }
            *y = v;
          }
        } else {
          std::sort(lo, hi, full_less);
        }
      }
      run_begins.clear(); run_ends.clear();
      std::memcpy(a + start[b], r, m * sizeof(SortRec));
    }
  });
  if (timer != nullptr) { timer->lap("  sort: buckets sorted by radix"); }
  return true;
}

}  // namespace


extern "C" int swa_hostdb_read_fasta_staged(const char * path, int usearch, int64_t append_abundance, int check_dup_seqs,
                                            swa_words_ready_fn on_words, void * user, swa_hostdb ** out) {
  if (out == nullptr || path == nullptr) { return SWA_E_ARG; }
  PhaseTimer timer;
  auto * db = new swa_hostdb();
  *out = db;
  Input in;
  if (!load_input(path, in, db->error)) { return SWA_E_ARG; }
  // The mapping of the input stays with the handle, like the scratch block below: unmapping 1.6 GB of text costs this
  // thread ~25 ms; with the handle (or the process) it costs the same or — when the kernel takes the address space apart
  // on its own time — nothing, and the pages are the page cache's either way.  (On a failed read it goes at once.)
  struct Unmap {
    Input & i; swa_hostdb * db; bool keep = false;
    ~Unmap() {
      if (!i.mapped) { return; }
      if (keep) { db->input_map = const_cast<char *>(i.data); db->input_size = i.size; }
      else { ::munmap(const_cast<char *>(i.data), i.size); }
      i.mapped = false;
    }
  } unmap{in, db};

  int8_t map[256];
  std::memset(map, -1, sizeof(map));
  map['A'] = map['a'] = 0; map['C'] = map['c'] = 1; map['G'] = map['g'] = 2;
  map['T'] = map['t'] = 3; map['U'] = map['u'] = 3;                               // db.cc:100-114

  // ---- cut the text at record boundaries ("\n>") and parse the pieces in parallel
  const unsigned threads = worker_count(in.size);
  std::vector<size_t> cuts(threads + 1, in.size);
  cuts[0] = 0;
  for (unsigned t = 1; t < threads; ++t) {
    size_t pos = in.size / threads * t;
    if (pos < cuts[t - 1]) { pos = cuts[t - 1]; }
    if (in.fd >= 0) { cuts[t] = next_record_start(in.fd, pos, in.size); continue; }
    while (pos < in.size) {
      const void * nl = std::memchr(in.data + pos, '\n', in.size - pos);
      if (nl == nullptr) { pos = in.size; break; }
      pos = (size_t)((const char *)nl - in.data) + 1;
      if (pos < in.size && in.data[pos] == '>') { break; }
    }
    cuts[t] = pos;
  }
  std::vector<Piece> parsed(threads);
  run_parallel(threads, [&](unsigned t) {
    if (cuts[t] < cuts[t + 1]) {
      if (in.fd >= 0) { parse_piece_streamed(in.fd, cuts[t], cuts[t + 1], map, usearch != 0, append_abundance, t, parsed[t]); }
      else { parse_piece(in.data + cuts[t], in.data + cuts[t + 1], map, usearch != 0, append_abundance, t, parsed[t], cuts[t + 1] - cuts[t]); }
    }
  });
  timer.lap("map + parallel parse");

  // ---- first error in file order, with absolute line numbers
  uint64_t lines_before = 0;
  for (unsigned t = 0; t < threads; ++t) {
    Piece & pc = parsed[t];
    if (pc.error_kind != 0) {
      if (pc.error_kind == 1) {                     // the first \x01 stands for the absolute line number
        const size_t at = pc.error.find('\x01');
        db->error = pc.error;
        if (at != std::string::npos) { db->error.replace(at, 1, std::to_string(pc.error_line + lines_before)); }
      } else {
        db->error = pc.error;
      }
      return SWA_E_ARG;
    }
    if (pc.missing != 0) { pc.missing_line += (uint32_t)lines_before; }
    lines_before += pc.lines;
  }

  // ---- what the pieces hold becomes the database's storage, as it lies
  db->pieces.resize(threads);
  std::vector<uint64_t> piece_first(threads + 1, 0);        // entries before piece p
  db->piece_word_first.assign(threads + 1, 0);
  for (unsigned t = 0; t < threads; ++t) {
    db->pieces[t] = std::move(parsed[t].data);
    piece_first[t + 1] = piece_first[t] + db->pieces[t].entries.size();
    db->piece_word_first[t + 1] = db->piece_word_first[t] + db->pieces[t].words.size();
    db->nucleotides += parsed[t].nucleotides;
    db->longest = std::max(db->longest, parsed[t].longest);
    db->longest_header = std::max(db->longest_header, parsed[t].longest_header);
  }
  const auto & pieces = db->pieces;
  // the first second-phase error in file order (abundance value / empty identifier), as entry index + text
  uint64_t late_at = ~0ull;
  std::string late_error;
  {
    uint64_t lines = 0;
    for (unsigned t = 0; t < threads && late_at == ~0ull; ++t) {
      const Piece & pc = parsed[t];
      if (pc.late_kind != 0) {
        late_at = piece_first[t] + pc.late_entry;
        late_error = pc.late_error;
        const size_t at = late_error.find('\x01');
        if (pc.late_kind == 1 && at != std::string::npos) { late_error.replace(at, 1, std::to_string(pc.late_line + lines)); }
      }
      lines += pc.lines;
    }
  }
  const uint64_t n64 = piece_first[threads];
  if (n64 > 0xFFFFFFFEull) { db->error = "\nError: too many sequences.\n"; return SWA_E_ARG; }
  const uint32_t n = (uint32_t)n64;
  db->n = n;
  // The packed words are final: a caller with a GPU coming up beside this thread can start copying them (they are
  // put in db order on the device, once the order is known: swa_db_stage_words / swa_db_upload_unordered)
  if (on_words != nullptr && n != 0 && late_at == ~0ull) {
    std::vector<const uint64_t *> ptrs(threads);
    std::vector<uint64_t> counts(threads);
    for (unsigned t = 0; t < threads; ++t) { ptrs[t] = pieces[t].words.data(); counts[t] = pieces[t].words.size(); }
    on_words(user, ptrs.data(), counts.data(), threads);
  }
  // entry g of the file (pieces hold almost equal shares of the entries: a guess — by a multiplication, this runs
  // several times per amplicon — and a step or two)
  const uint64_t pieces_per_entry_q32 = ((uint64_t)threads << 32) / std::max<uint64_t>(n64, 1);
  auto entry_at = [&](uint64_t g) -> const swa_entry & {
    unsigned p = (unsigned)std::min<uint64_t>(threads - 1, (g * pieces_per_entry_q32) >> 32);
    while (g < piece_first[p]) { --p; }
    while (g >= piece_first[p + 1]) { ++p; }
    return pieces[p].entries[g - piece_first[p]];
  };
  auto hdr_of = [&](const swa_entry & e) { return pieces[e.piece()].hdr_pool.data() + e.hdr_off; };
  auto words_of = [&](const swa_entry & e) { return pieces[e.piece()].words.data() + e.word_off; };
  timer.lap("pieces taken over");

  // ---- identifier uniqueness (db.cc:680-758) without a table in DRAM.  The reference's hash table over the
  // identifiers is one cache miss per amplicon (round 4's lock-free version of it: 1.3 CPU seconds at 10 M amplicons,
  // 40 % of the reader's — and CPU seconds are what a container with a CPU quota runs out of, lease r5f).  Here: a 64-bit
  // hash per identifier (the headers stream, piece by piece), the (hash, entry) pairs partitioned by the hash's top bits
  // into buckets of ~10 000 (counting sort: sequential writes), and per bucket a table that lives in L2.  Two equal
  // 42-bit hash parts are compared as strings; the LATER entry of the earliest repetition is reported, like the
  // sequential scan of the reference does.
  std::string dup_id_error, dup_seq_error;
  uint64_t dup_id_at = ~0ull, dup_seq_at = ~0ull;           // entry indices (the later entry of the earliest repetition)
  // ONE block of scratch memory for the checks and the sort, one after the other (34 bytes per amplicon: the sort's two
  // record arrays and bucket numbers; the checks take 18 of them): fresh memory is paid for twice — populated when it
  // is mapped, taken apart when it is freed, ~12 ms each way per 160 MB here — and the five arrays this replaces were
  // each mapped and freed on this thread.  It stays with the handle (unmapping 340 MB here would cost this thread 25 ms; at process exit it costs the same or — when the kernel takes the address space apart on its own time — nothing).
  db->scratch.resize((size_t)n * 34 + 64);
  char * const scratch = db->scratch.data();
  timer.lap("scratch block");
  auto id_span = [&](const swa_entry & e, const char *& str, uint32_t & l) {
    const char * hdr = hdr_of(e);
    if (e.ab_start > 0) { str = hdr; l = (uint32_t)e.ab_start; }
    else { str = hdr + e.ab_end; l = e.hdr_len() - (uint32_t)e.ab_end; }
  };
  // the same scheme for both checks: hash_of(entry) -> 64 bits, same(entry, entry) -> bool
  auto earliest_repetition = [&](auto && hash_of, auto && same) -> uint64_t {
    if (n < 2) { return ~0ull; }
    unsigned bucket_bits = 1;
    while (bucket_bits < 16 && (n64 >> bucket_bits) > 8192) { ++bucket_bits; }
    const uint64_t buckets = 1ull << bucket_bits;
    uint64_t * const packed = reinterpret_cast<uint64_t *>(scratch);          // key32 << 32 | entry; `bucket` = top bits of the hash, kept apart
    uint64_t * const parted = packed + n;
    uint16_t * const bucket_of = reinterpret_cast<uint16_t *>(parted + n);
    std::vector<uint64_t> place((size_t)threads * buckets, 0);
    run_parallel(threads, [&](unsigned t) {
      uint64_t * c = &place[(size_t)t * buckets];
      uint64_t g = piece_first[t];
      for (const swa_entry & e : pieces[t].entries) {
        const uint64_t h = hash_of(e);
        const uint64_t b = h >> (64u - bucket_bits);
        packed[g] = (((h >> 22) & 0xFFFFFFFFull) << 32) | g;
        bucket_of[g] = (uint16_t)b;
        ++c[b];
        ++g;
      }
    });
    std::vector<uint64_t> start(buckets + 1, 0);
    uint64_t at = 0;
    for (uint64_t b = 0; b < buckets; ++b) {
      start[b] = at;
      for (unsigned t = 0; t < threads; ++t) { const uint64_t c = place[(size_t)t * buckets + b]; place[(size_t)t * buckets + b] = at; at += c; }
    }
    start[buckets] = at;
    run_parallel(threads, [&](unsigned t) {
      uint64_t * c = &place[(size_t)t * buckets];
      for (uint64_t g = piece_first[t]; g < piece_first[t + 1]; ++g) { parted[c[bucket_of[g]]++] = packed[g]; }
    });
    std::atomic<uint64_t> found{~0ull};
    std::atomic<uint64_t> next{0};
    run_parallel(threads, [&](unsigned) {
      std::vector<uint64_t> table;
      for (;;) {
        const uint64_t b = next.fetch_add(1);
        if (b >= buckets) { break; }
        const uint64_t m = start[b + 1] - start[b];
        uint64_t tsize = 64;
        while (tsize < 3 * m) { tsize <<= 1; }
        table.assign(tsize, ~0ull);
        for (uint64_t i = start[b]; i < start[b + 1]; ++i) {
          const uint64_t rec = parted[i];
          uint64_t slot = ((rec >> 32) * 0x9E3779B1ull) & (tsize - 1);
          for (;;) {
            const uint64_t cur = table[slot];
            if (cur == ~0ull) { table[slot] = rec; break; }
            if ((cur >> 32) == (rec >> 32) && same(entry_at((uint32_t)cur), entry_at((uint32_t)rec))) {
              const uint64_t later = std::max<uint64_t>((uint32_t)cur, (uint32_t)rec);
              uint64_t seen = found.load();
              while (later < seen && !found.compare_exchange_weak(seen, later)) { }
              // (the EARLIER of the two stays in the table: a third copy is then compared with it)
              if ((uint32_t)rec < (uint32_t)cur) { table[slot] = rec; }
              break;
            }
            slot = (slot + 1) & (tsize - 1);
          }
        }
      }
    });
    return found.load();
  };
  {
    dup_id_at = earliest_repetition(
        [&](const swa_entry & e) { const char * ids; uint32_t idl; id_span(e, ids, idl); const uint64_t h = bytes_hash(ids, idl); return h * 0x9E3779B97F4A7C15ull; },
        [&](const swa_entry & x, const swa_entry & y) {
          const char * xs; const char * ys; uint32_t xl, yl;
          id_span(x, xs, xl); id_span(y, ys, yl);
          return xl == yl && std::memcmp(xs, ys, xl) == 0;
        });
    if (dup_id_at != ~0ull) {
      const char * ids; uint32_t idl;
      id_span(entry_at(dup_id_at), ids, idl);
      dup_id_error = "\nError: Duplicated sequence identifier: " + std::string(ids, idl) + "\n";
    }
  }
  timer.lap("identifiers checked");
  // duplicated sequences are checked here only for d > 1 (db.cc:763-790); d = 1 finds
  // them while building the amplicon table (algod1.cc:1131-1150 / swa_d1_index_build)
  if (check_dup_seqs && n > 1) {
    dup_seq_at = earliest_repetition(
        [&](const swa_entry & e) {
          const uint64_t * w = words_of(e);
          const uint32_t nw = (e.seqlen + 31u) >> 5;
          uint64_t hsh = e.seqlen * 0x9E3779B97F4A7C15ull;
          for (uint32_t k = 0; k < nw; ++k) { hsh = (hsh ^ w[k]) * 0xff51afd7ed558ccdull; hsh ^= hsh >> 32; }
          return hsh * 0x9E3779B97F4A7C15ull;
        },
        [&](const swa_entry & x, const swa_entry & y) {
          return x.seqlen == y.seqlen && std::memcmp(words_of(x), words_of(y), ((x.seqlen + 31u) >> 5) * 8ull) == 0;
        });
    if (dup_seq_at != ~0ull) {
      dup_seq_error = "\nError: some fasta entries have identical sequences.\n"
                  "Swarm expects dereplicated fasta files.\n"
                  "Such files can be produced with swarm or vsearch:\n"
                  " swarm -d 0 -w derep.fasta -o /dev/null input.fasta\n"
                  "or\n"
                  " vsearch --derep_fulllength input.fasta --sizein --sizeout --output derep.fasta\n";
    }
    timer.lap("sequences checked");
  }
  // The reference walks the entries once (db.cc:676-795): abundance value, empty identifier, repeated identifier are
  // fatal at the entry where they occur, a repeated sequence (d > 1) ends the walk there and is reported after it.
  // Hence: the earliest entry with any of them decides; at the same entry in that order.
  auto checks_verdict = [&]() {
    if (late_at != ~0ull && late_at <= dup_id_at && late_at <= dup_seq_at) { db->error = late_error; return (int)SWA_E_ARG; }
    if (dup_id_at != ~0ull && dup_id_at <= dup_seq_at) { db->error = dup_id_error; return (int)SWA_E_ARG; }
    if (dup_seq_at != ~0ull) { db->error = dup_seq_error; return (int)SWA_E_DUPLICATES; }
    return (int)SWA_OK;
  };

  if (late_at != ~0ull) {                                   // (no point in sorting: some error will be reported)
    const int rc_checks = checks_verdict();
    return rc_checks;
  }
  {                                                                                  // db.cc:369-385
    uint64_t missing = 0;
    uint32_t missing_line = 0;
    std::string missing_hdr;
    for (const auto & pc : parsed) {
      if (pc.missing != 0 && missing == 0) { missing_line = pc.missing_line; missing_hdr = pc.missing_hdr; }
      missing += pc.missing;
    }
    if (missing != 0) {
      if (const int rc_checks = checks_verdict()) { return rc_checks; }     // (an identifier / sequence error comes first)
      db->error = "\nError: Abundance annotations not found for " + std::to_string(missing) +
                  " sequences, starting on line " + std::to_string(missing_line) + ".\n>" + missing_hdr + "\n" +
                  "Fasta headers must end with abundance annotations (_INT or ;size=INT).\n"
                  "The -z option must be used if the abundance annotation is in the latter format.\n"
                  "Abundance annotations can be produced by dereplicating the sequences.\n"
                  "The header is defined as the string comprised between the \">\" symbol\n"
                  "and the first space or the end of the line, whichever comes first.\n";
      return SWA_E_ARG;
    }
  }

  if (const int rc_checks = checks_verdict()) { return rc_checks; }

  // ---- db order: abundance descending, then header (strcmp) ascending — db.cc:388-413.
  // The sort moves compact records: the abundance (saturating at 32 bits: two saturated ones are told apart through
  // their entries), the first 8 header bytes big endian (they decide most ties without touching the header text; equal
  // prefixes fall through to strcmp: same order) and the entry's number.  Inputs that are already in db order (swarm's
  // own -w output, vsearch output) skip it.
  SortRec * const recs = reinterpret_cast<SortRec *>(scratch);
  std::atomic<bool> saturated{false};                       // an abundance of 2^32 - 1 or more somewhere: K does not order those
  run_parallel(threads, [&](unsigned t) {
    uint64_t g = piece_first[t];
    const char * pool = pieces[t].hdr_pool.data();
    for (const swa_entry & e : pieces[t].entries) {
      const unsigned char * h = reinterpret_cast<const unsigned char *>(pool + e.hdr_off);
      const uint32_t hlen = e.hdr_len();
      uint64_t key = 0;
      for (uint32_t i = 0; i < 8; ++i) { key = (key << 8) | (i < hlen ? h[i] : 0u); }
      recs[g] = SortRec{key, (uint32_t)std::min<uint64_t>(e.abundance, 0xFFFFFFFFull), (uint32_t)g};
      if (e.abundance >= 0xFFFFFFFFull) { saturated.store(true, std::memory_order_relaxed); }
      ++g;
    }
  });
  auto less = [&](const SortRec & a, const SortRec & b) {
    if (a.abundance != b.abundance) { return a.abundance > b.abundance; }
    if (a.abundance == 0xFFFFFFFFu) {
      const uint64_t x = entry_at(a.entry).abundance, y = entry_at(b.entry).abundance;
      if (x != y) { return x > y; }
    }
    if (a.key8 != b.key8) { return a.key8 < b.key8; }
    return std::strcmp(hdr_of(entry_at(a.entry)), hdr_of(entry_at(b.entry))) < 0;
  };
  // (for the radix sort's runs of equal keys: identifier bytes [8, 16) of a record, and how to ask for them ahead of time)
  struct Deeper {
    decltype(entry_at) & entry;
    decltype(hdr_of) & text;
    void ask_entry(const SortRec & r) const { __builtin_prefetch(&entry(r.entry)); }
    void ask_text(const SortRec & r) const { __builtin_prefetch(text(entry(r.entry)) + 8); }
    uint64_t key(const SortRec & r) const {
      const swa_entry & e = entry(r.entry);
      const unsigned char * h = reinterpret_cast<const unsigned char *>(text(e));
      const uint32_t hlen = e.hdr_len();
      uint64_t k = 0;
      for (uint32_t i = 8; i < 16; ++i) { k = (k << 8) | (i < hlen ? h[i] : 0u); }
      return k;
    }
  };
  std::atomic<bool> sorted{true};
  run_parallel(threads, [&](unsigned t) {
    const uint64_t lo = n64 * t / threads, hi = n64 * (t + 1) / threads;
    for (uint64_t i = std::max<uint64_t>(lo, 1); i < hi && sorted.load(std::memory_order_relaxed); ++i) {
      if (less(recs[i], recs[i - 1])) { sorted.store(false, std::memory_order_relaxed); }
    }
  });
  timer.lap("sort records made, order checked");
  if (!sorted.load()) {
    // Sample sort over the host cores: splitters from a regular sample, every thread files its share of the records
    // under them (a binary search per record; the counts per thread and bucket give every record its place without
    // atomics), the buckets — 8 per thread, so that an unlucky splitter costs little — are sorted independently.  The
    // order is a strict total order (identifiers are unique), so the result equals std::sort's.  (r04 used libstdc++'s
    // parallel multiway merge sort on 32-byte records: 151 ms at 10 M amplicons on 64 threads.)  32 threads, not 64: the
    // bucket sorts took 17-20 ms on 32 and 56-93 ms on 64 threads beside the checks' 32 (lease r5b).
    SortRec * const other = recs + n;
    uint16_t * const where = reinterpret_cast<uint16_t *>(other + n);
    const char * env_sort = std::getenv("SWARM_AMD_SORT_THREADS");
    const unsigned sort_threads = std::max(1u, std::min(threads, env_sort != nullptr ? (unsigned)std::atoi(env_sort) : 32u));
    // The records' integer key (abundance, first 8 identifier bytes) sorts by radix — a third of the comparison sort's CPU
    // time at 10 M amplicons —; identifiers that mostly share their first 8 bytes and small inputs take the comparison sort.
    // Both leave the one db order (identifiers are unique: a strict total order).  Records whose abundance saturates the
    // key's 32 bits — a heavy-tailed set of 10^8 amplicons has a few — come before all others whatever their identifiers: they
    // are moved to the front and sorted among themselves through their entries; the radix sort gets the rest.
    static const bool no_radix = std::getenv("SWARM_AMD_NO_RADIX_SORT") != nullptr;
    uint64_t nsat = 0;
    if (!no_radix && saturated.load()) {
      std::vector<std::vector<uint64_t>> found(threads);
      run_parallel(threads, [&](unsigned t) {
        for (uint64_t i = n64 * t / threads; i < n64 * (t + 1) / threads; ++i) { if (recs[i].abundance == 0xFFFFFFFFu) { found[t].push_back(i); } }
      });
      std::vector<uint64_t> at;                             // where they lie, ascending
      for (const auto & f : found) { at.insert(at.end(), f.begin(), f.end()); }
      nsat = at.size();
      // the places among the first nsat that hold another record, against the saturated records behind them
      std::vector<uint64_t> holes;
      {
        size_t k = 0;
        for (uint64_t i = 0; i < nsat; ++i) {
          while (k < at.size() && at[k] < i) { ++k; }
          if (k < at.size() && at[k] == i) { continue; }
          holes.push_back(i);
        }
      }
      size_t h = 0;
      for (const uint64_t p : at) { if (p >= nsat) { std::swap(recs[holes[h]], recs[p]); ++h; } }
      std::sort(recs, recs + nsat, less);
    }
    const bool by_radix = !no_radix && sort_threads > 1 && n64 - nsat >= 100000 &&
                          parallel_radix_sort(recs + nsat, other + nsat, where + nsat, n64 - nsat, sort_threads, less, Deeper{entry_at, hdr_of}, &timer);
    if (!by_radix) { parallel_sample_sort(recs, other, where, n64, sort_threads, less, &timer); }
  }
  timer.lap("sort");

  // ---- db order as pointers into the pieces; the plain arrays the GPU upload and the per-swarm sums read
  db->ent.resize(n);
  db->seqlen.resize(n);
  db->abundance.resize(n);
  db->src_off.resize(n);
  timer.lap("db-order arrays sized");
  run_parallel(threads, [&](unsigned t) {
    const uint64_t lo = n64 * t / threads, hi = n64 * (t + 1) / threads;
    for (uint64_t k = lo; k < hi; ++k) {
      if (k + 32 < hi) {                                    // (an entry is 40 bytes: it may lie across two cache lines)
        const char * ahead = reinterpret_cast<const char *>(&entry_at(recs[k + 32].entry));
        __builtin_prefetch(ahead + 8); __builtin_prefetch(ahead + 31);
      }
      const swa_entry & e = entry_at(recs[k].entry);
      db->ent[k] = &e;
      db->seqlen[k] = e.seqlen;
      db->abundance[k] = e.abundance;
      db->src_off[k] = db->piece_word_first[e.piece()] + e.word_off;
    }
  });
  timer.lap("db order");
  unmap.keep = true;
  return SWA_OK;
}

extern "C" int swa_hostdb_read_fasta(const char * path, int usearch, int64_t append_abundance, int check_dup_seqs,
                                     swa_hostdb ** out) {
  return swa_hostdb_read_fasta_staged(path, usearch, append_abundance, check_dup_seqs, nullptr, nullptr, out);
}

extern "C" void swa_hostdb_free(swa_hostdb * db) { delete db; }

extern "C" const char * swa_hostdb_error(const swa_hostdb * db) { return db != nullptr ? db->error.c_str() : ""; }

// the packed sequences contiguous in db order (swa_db_view): gathered from the pieces the first time somebody asks
static void order_on_host(swa_hostdb * db) {
  if (db->ordered) { return; }
  const uint64_t n = db->n;
  db->seq_off.resize(n + 1);
  const unsigned threads = swa_pool::get().size();
  std::vector<uint64_t> sum(threads + 1, 0);
  run_parallel(threads, [&](unsigned t) {
    uint64_t w = 0;
    for (uint64_t k = n * t / threads; k < n * (t + 1) / threads; ++k) { w += (db->seqlen[k] + 31u) >> 5; }
    sum[t + 1] = w;
  });
  for (unsigned t = 0; t < threads; ++t) { sum[t + 1] += sum[t]; }
  db->seqs.resize(sum[threads] + 1);
  run_parallel(threads, [&](unsigned t) {
    uint64_t w = sum[t];
    const uint64_t lo = n * t / threads, hi = n * (t + 1) / threads;
    for (uint64_t k = lo; k < hi; ++k) {
      if (k + 8 < hi) { __builtin_prefetch(db->words((uint32_t)(k + 8))); }
      const uint64_t nw = (db->seqlen[k] + 31u) >> 5;
      db->seq_off[k] = w;
      std::memcpy(&db->seqs[w], db->words((uint32_t)k), nw * 8ull);
      w += nw;
    }
  });
  db->seq_off[n] = sum[threads];
  db->seqs[sum[threads]] = 0;
  db->ordered = true;
}

extern "C" void swa_hostdb_view(const swa_hostdb * db, swa_db_view * v) {
  order_on_host(const_cast<swa_hostdb *>(db));
  v->n = db->n;
  v->longest = db->longest;
  v->seqs = db->seqs.data();
  v->seq_off = db->seq_off.data();
  v->seqlen = db->seqlen.data();
  v->abundance = db->abundance.data();
}

extern "C" void swa_hostdb_unordered_view(const swa_hostdb * db, swa_db_unordered_view * v) {
  auto * mdb = const_cast<swa_hostdb *>(db);
  mdb->piece_ptrs.resize(db->pieces.size());
  mdb->piece_counts.resize(db->pieces.size());
  for (size_t p = 0; p < db->pieces.size(); ++p) { mdb->piece_ptrs[p] = db->pieces[p].words.data(); mdb->piece_counts[p] = db->pieces[p].words.size(); }
  v->n = db->n;
  v->longest = db->longest;
  v->pieces = (uint32_t)db->pieces.size();
  v->piece_words = mdb->piece_ptrs.data();
  v->piece_word_count = mdb->piece_counts.data();
  v->src_off = db->src_off.data();
  v->seqlen = db->seqlen.data();
  v->abundance = db->abundance.data();
}

extern "C" uint64_t swa_hostdb_nucleotides(const swa_hostdb * db) { return db->nucleotides; }

extern "C" const char * swa_hostdb_header(const swa_hostdb * db, uint32_t i, uint32_t * len) {
  if (len != nullptr) { *len = db->ent[i]->hdr_len(); }
  return db->hdr(i);
}
