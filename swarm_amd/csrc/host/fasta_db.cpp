// fasta_db.cpp — host side of seam L2: FASTA -> packed amplicon database in the
// reference's db order.  Behavioural mirror of db_read (src/db.cc:432-803): same
// accepted alphabet, same header / abundance rules, same error texts, same sort
// (abundance descending, then header bytes ascending, src/db.cc:388-413) — with a
// different shape: the whole input is scanned once from memory into SoA arrays whose
// sequence words are 8-byte aligned and contiguous in SORTED order (what the GPU
// wants to stream), instead of the reference's interleaved header/sequence blob.
#include "hostdb.h"

#include <algorithm>
#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

namespace {

struct RawEntry {
  uint64_t hdr_off;      // into raw header pool
  uint32_t hdr_len;
  uint32_t lineno;       // line of the '>' header
  uint64_t word_off;     // into raw word pool
  uint32_t seqlen;
  int32_t ab_start;      // abundance annotation [start, end) inside the header
  int32_t ab_end;
  uint64_t abundance;
};

bool read_all(const char * path, std::vector<char> & buf, std::string & err) {
  FILE * fp = nullptr;
  const bool is_stdin = (std::strcmp(path, "-") == 0);
  fp = is_stdin ? stdin : std::fopen(path, "rb");
  if (fp == nullptr) {
    err = std::string("\nError: Unable to open input data file (") + path + ").\n";   // db.cc:458-461
    return false;
  }
  size_t used = 0;
  buf.resize(1 << 20);
  for (;;) {
    if (used == buf.size()) { buf.resize(buf.size() * 2); }
    const size_t got = std::fread(buf.data() + used, 1, buf.size() - used, fp);
    used += got;
    if (got == 0) { break; }
  }
  if (!is_stdin) { std::fclose(fp); }
  buf.resize(used);
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
  char tmp[24];
  std::memcpy(tmp, h + us + 1, digits);
  tmp[digits] = 0;
  number = std::atol(tmp);          // "_" alone gives 0 like atol("")
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

inline uint64_t id_hash(const char * s, uint32_t len) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint32_t i = 0; i < len; ++i) { h = (h ^ (unsigned char)s[i]) * 0x100000001b3ull; }
  return h ^ (h >> 29);
}

}  // namespace

extern "C" int swa_hostdb_read_fasta(const char * path, int usearch, int64_t append_abundance, int check_dup_seqs,
                                     swa_hostdb ** out) {
  if (out == nullptr || path == nullptr) { return SWA_E_ARG; }
  auto * db = new swa_hostdb();
  *out = db;
  std::vector<char> buf;
  if (!read_all(path, buf, db->error)) { return SWA_E_ARG; }

  int8_t map[256];
  std::memset(map, -1, sizeof(map));
  map['A'] = map['a'] = 0; map['C'] = map['c'] = 1; map['G'] = map['g'] = 2;
  map['T'] = map['t'] = 3; map['U'] = map['u'] = 3;                               // db.cc:100-114

  std::vector<RawEntry> raw;
  std::vector<char> hdr_pool;
  std::vector<uint64_t> words;
  raw.reserve(buf.size() / 160 + 16);
  words.reserve(buf.size() / 28 + 16);
  hdr_pool.reserve(buf.size() / 12 + 16);

  const char * p = buf.data();
  const char * const end = p + buf.size();
  uint32_t lineno = 1;
  auto line_end = [&](const char * q) { const void * nl = std::memchr(q, '\n', (size_t)(end - q)); return nl ? (const char *)nl : end; };

  while (p < end) {
    if (*p != '>') {
      db->error = "\nError: Illegal header line in fasta file.\n";                 // db.cc:492-494
      return SWA_E_ARG;
    }
    const char * le = line_end(p);
    const char * h = p + 1;
    uint32_t hlen = 0;
    while (h + hlen < le && h[hlen] != ' ' && h[hlen] != '\r' && h[hlen] != 0) { ++hlen; }   // db.cc:498-499
    if (hlen > 16777215u) {
      db->error = "\nError: Headers longer than 16,777,215 symbols are not supported.\n";
      return SWA_E_ARG;
    }
    RawEntry e{};
    e.hdr_off = hdr_pool.size();
    e.hdr_len = hlen;
    e.lineno = lineno;
    hdr_pool.insert(hdr_pool.end(), h, h + hlen);
    hdr_pool.push_back('\0');
    p = (le < end) ? le + 1 : end;
    ++lineno;

    e.word_off = words.size();
    uint64_t acc = 0;
    uint32_t fill = 0;
    uint32_t len = 0;
    while (p < end && *p != '>') {
      le = line_end(p);
      for (const char * q = p; q < le; ++q) {
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
          if (ch >= 32 && ch <= 126) {
            std::snprintf(msg, sizeof(msg), "\nError: Illegal character '%c' in sequence on line %u.\n", ch, lineno);
          } else {
            std::snprintf(msg, sizeof(msg), "\nError: Illegal character (ascii no %d) in sequence on line %u.\n", (int)ch, lineno);
          }
          db->error = msg;                                                          // db.cc:575-590
          return SWA_E_ARG;
        }
      }
      if (len > 67108861u) {
        db->error = "\nError: Sequences longer than 67,108,861 symbols are not supported.\n";
        return SWA_E_ARG;
      }
      p = (le < end) ? le + 1 : end;
      ++lineno;
    }
    if (len == 0) {
      db->error = "\nError: Empty sequence found on line " + std::to_string(lineno - 1) + ".\n";   // db.cc:608-611
      return SWA_E_ARG;
    }
    if (fill > 0) { words.push_back(acc); }
    e.seqlen = len;
    db->nucleotides += len;
    db->longest = std::max(db->longest, len);
    db->longest_header = std::max(db->longest_header, hlen);
    raw.push_back(e);
  }
  const uint64_t n64 = raw.size();
  if (n64 > 0xFFFFFFFEull) { db->error = "\nError: too many sequences.\n"; return SWA_E_ARG; }
  const uint32_t n = (uint32_t)n64;
  db->n = n;

  // abundances, identifier uniqueness (db.cc:286-343, 680-758)
  uint64_t missing = 0;
  uint32_t missing_line = 0;
  const char * missing_hdr = nullptr;
  std::vector<uint32_t> idtab(n ? 2ull * n : 1, 0xFFFFFFFFu);
  auto id_span = [&](const RawEntry & e, const char *& s, uint32_t & l) {
    const char * hdr = hdr_pool.data() + e.hdr_off;
    if (e.ab_start > 0) { s = hdr; l = (uint32_t)e.ab_start; }
    else { s = hdr + e.ab_end; l = e.hdr_len - (uint32_t)e.ab_end; }
  };
  for (uint32_t i = 0; i < n; ++i) {
    RawEntry & e = raw[i];
    const char * hdr = hdr_pool.data() + e.hdr_off;
    int32_t s = 0, t = 0;
    int64_t number = 0;
    int64_t abundance = 0;
    const bool found = usearch ? usearch_abundance(hdr, e.hdr_len, s, t, number) : swarm_abundance(hdr, e.hdr_len, s, t, number);
    if (found) {
      if (number <= 0) {
        db->error = "\nError: Illegal abundance value on line " + std::to_string(e.lineno) + ":\n" + hdr +
                    "\nAbundance values should be positive integers.\n";
        return SWA_E_ARG;
      }
      abundance = number;
    }
    if (abundance == 0) {
      s = (int32_t)e.hdr_len;
      t = s;
      if (append_abundance != 0) { abundance = append_abundance; }
      else {
        if (++missing == 1) { missing_line = e.lineno; missing_hdr = hdr; }
      }
    }
    e.abundance = (uint64_t)abundance;
    e.ab_start = s;
    e.ab_end = t;
    if (e.ab_start == 0 && e.ab_end == (int32_t)e.hdr_len) {
      db->error = "\nError: Empty sequence identifier.\n";
      return SWA_E_ARG;
    }
    const char * ids; uint32_t idl;
    id_span(e, ids, idl);
    uint64_t slot = id_hash(ids, idl) % idtab.size();
    while (idtab[slot] != 0xFFFFFFFFu) {
      const char * os; uint32_t ol;
      id_span(raw[idtab[slot]], os, ol);
      if (ol == idl && std::memcmp(os, ids, idl) == 0) {
        db->error = "\nError: Duplicated sequence identifier: " + std::string(ids, idl) + "\n";
        return SWA_E_ARG;
      }
      slot = (slot + 1) % idtab.size();
    }
    idtab[slot] = i;
  }
  idtab.clear(); idtab.shrink_to_fit();

  // duplicated sequences are checked here only for d > 1 (db.cc:763-790); d = 1 finds
  // them while building the amplicon table (algod1.cc:1131-1150 / swa_d1_index_build)
  if (check_dup_seqs && n > 1) {
    std::vector<uint32_t> tab(2ull * n, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < n; ++i) {
      const RawEntry & e = raw[i];
      const uint32_t nw = (e.seqlen + 31u) >> 5;
      uint64_t hsh = e.seqlen * 0x9E3779B97F4A7C15ull;
      for (uint32_t w = 0; w < nw; ++w) { hsh = (hsh ^ words[e.word_off + w]) * 0xff51afd7ed558ccdull; hsh ^= hsh >> 32; }
      uint64_t slot = hsh % tab.size();
      while (tab[slot] != 0xFFFFFFFFu) {
        const RawEntry & o = raw[tab[slot]];
        if (o.seqlen == e.seqlen && std::memcmp(&words[o.word_off], &words[e.word_off], nw * 8ull) == 0) {
          db->error = "\nError: some fasta entries have identical sequences.\n"
                      "Swarm expects dereplicated fasta files.\n"
                      "Such files can be produced with swarm or vsearch:\n"
                      " swarm -d 0 -w derep.fasta -o /dev/null input.fasta\n"
                      "or\n"
                      " vsearch --derep_fulllength input.fasta --sizein --sizeout --output derep.fasta\n";
          return SWA_E_DUPLICATES;
        }
        slot = (slot + 1) % tab.size();
      }
      tab[slot] = i;
    }
  }
  if (missing != 0) {                                                                // db.cc:369-385
    db->error = "\nError: Abundance annotations not found for " + std::to_string(missing) +
                " sequences, starting on line " + std::to_string(missing_line) + ".\n>" + missing_hdr + "\n" +
                "Fasta headers must end with abundance annotations (_INT or ;size=INT).\n"
                "The -z option must be used if the abundance annotation is in the latter format.\n"
                "Abundance annotations can be produced by dereplicating the sequences.\n"
                "The header is defined as the string comprised between the \">\" symbol\n"
                "and the first space or the end of the line, whichever comes first.\n";
    return SWA_E_ARG;
  }

  // db order: abundance descending, then header (strcmp) ascending — db.cc:388-413
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  auto less = [&](uint32_t a, uint32_t b) {
    const RawEntry & x = raw[a];
    const RawEntry & y = raw[b];
    if (x.abundance != y.abundance) { return x.abundance > y.abundance; }
    return std::strcmp(hdr_pool.data() + x.hdr_off, hdr_pool.data() + y.hdr_off) < 0;
  };
  if (!std::is_sorted(order.begin(), order.end(), less)) { std::sort(order.begin(), order.end(), less); }

  db->seq_off.resize((size_t)n + 1);
  db->seqlen.resize(n);
  db->abundance.resize(n);
  db->hdr_off.resize((size_t)n + 1);
  db->ab_start.resize(n);
  db->ab_end.resize(n);
  db->seqs.resize(words.size() + 1);
  db->headers.resize(hdr_pool.size() + 1);
  uint64_t woff = 0, hoff = 0;
  for (uint32_t k = 0; k < n; ++k) {
    const RawEntry & e = raw[order[k]];
    const uint32_t nw = (e.seqlen + 31u) >> 5;
    db->seq_off[k] = woff;
    std::memcpy(&db->seqs[woff], &words[e.word_off], nw * 8ull);
    woff += nw;
    db->seqlen[k] = e.seqlen;
    db->abundance[k] = e.abundance;
    db->hdr_off[k] = hoff;
    std::memcpy(&db->headers[hoff], hdr_pool.data() + e.hdr_off, (size_t)e.hdr_len + 1);
    hoff += (uint64_t)e.hdr_len + 1;
    db->ab_start[k] = e.ab_start;
    db->ab_end[k] = e.ab_end;
  }
  db->seq_off[n] = woff;
  db->hdr_off[n] = hoff;
  db->seqs[woff] = 0;
  return SWA_OK;
}

extern "C" void swa_hostdb_free(swa_hostdb * db) { delete db; }

extern "C" const char * swa_hostdb_error(const swa_hostdb * db) { return db != nullptr ? db->error.c_str() : ""; }

extern "C" void swa_hostdb_view(const swa_hostdb * db, swa_db_view * v) {
  v->n = db->n;
  v->longest = db->longest;
  v->seqs = db->seqs.data();
  v->seq_off = db->seq_off.data();
  v->seqlen = db->seqlen.data();
  v->abundance = db->abundance.data();
}

extern "C" uint64_t swa_hostdb_nucleotides(const swa_hostdb * db) { return db->nucleotides; }

extern "C" const char * swa_hostdb_header(const swa_hostdb * db, uint32_t i, uint32_t * len) {
  if (len != nullptr) { *len = (uint32_t)(db->hdr_off[i + 1] - db->hdr_off[i] - 1); }
  return db->headers.data() + db->hdr_off[i];
}
