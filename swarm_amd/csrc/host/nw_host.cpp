// nw_host.cpp — see nw_host.h.  Minimum-cost global alignment with affine gaps and the
// reference's tie-breaking (comparison semantics of src/nw.cc:91-103, walk-back priority of
// src/nw.cc:139-172); only the uclust writer uses it.
#include "nw_host.h"

#include <algorithm>

namespace {
inline unsigned nt(const uint64_t * s, uint64_t p) { return (unsigned)((s[p >> 5] >> ((p & 31u) << 1)) & 3u); }
constexpr uint8_t kUp = 1, kLeft = 2, kExtUp = 4, kExtLeft = 8;
}  // namespace

uint64_t swa_nw_align(const uint64_t * dseq, uint32_t dlen, const uint64_t * qseq, uint32_t qlen, uint64_t mismatch,
                      uint64_t gapopen, uint64_t gapextend, swa_nw_scratch & sc) {
  sc.dir.assign((size_t)dlen * qlen + 1, 0);
  sc.he.resize(2 * (size_t)qlen + 2);
  for (uint64_t c = 0; c < qlen; ++c) {
    sc.he[2 * c] = gapopen + (c + 1) * gapextend;          // H above the first row
    sc.he[2 * c + 1] = 2 * gapopen + (c + 2) * gapextend;  // vertical gap state above the first row
  }
  for (uint64_t r = 0; r < dlen; ++r) {
    uint64_t across = 2 * gapopen + (r + 2) * gapextend;   // horizontal gap state left of the row
    uint64_t diag = r == 0 ? 0 : gapopen + r * gapextend;
    const unsigned dn = nt(dseq, r);
    for (uint64_t c = 0; c < qlen; ++c) {
      uint8_t bits = 0;
      const uint64_t next_diag = sc.he[2 * c];
      uint64_t down = sc.he[2 * c + 1];
      uint64_t h = diag + (dn == nt(qseq, c) ? 0 : mismatch);
      if (across < h) { bits |= kUp; h = across; }
      if (down < h) { h = down; }
      if (down == h) { bits |= kLeft; }
      sc.he[2 * c] = h;
      const uint64_t opened = h + gapopen + gapextend;
      down += gapextend;
      across += gapextend;
      if (across < opened) { bits |= kExtUp; }
      if (down < opened) { bits |= kExtLeft; }
      across = std::min(across, opened);
      down = std::min(down, opened);
      sc.he[2 * c + 1] = down;
      sc.dir[qlen * r + c] = bits;
      diag = next_diag;
    }
  }
  // walk back from the last cell
  sc.ops.clear();
  uint64_t matches = 0;
  uint64_t col = qlen, row = dlen;
  char op = 0;
  while (col > 0 && row > 0) {
    const uint8_t bits = sc.dir[qlen * (row - 1) + (col - 1)];
    if (op == 'I' && (bits & kExtLeft)) { --row; op = 'I'; }
    else if (op == 'D' && (bits & kExtUp)) { --col; op = 'D'; }
    else if (bits & kLeft) { --row; op = 'I'; }
    else if (bits & kUp) { --col; op = 'D'; }
    else {
      if (nt(qseq, col - 1) == nt(dseq, row - 1)) { ++matches; }
      --col; --row; op = 'M';
    }
    sc.ops.push_back(op);
  }
  sc.ops.append(col, 'D');
  sc.ops.append(row, 'I');
  const uint64_t columns = sc.ops.size();
  std::reverse(sc.ops.begin(), sc.ops.end());
  return columns - matches;
}

std::string swa_cigar(const std::string & ops) {
  std::string out;
  size_t i = 0;
  while (i < ops.size()) {
    size_t j = i;
    while (j < ops.size() && ops[j] == ops[i]) { ++j; }
    if (j - i > 1) { out += std::to_string(j - i); }
    out.push_back(ops[i]);
    i = j;
  }
  return out;
}
