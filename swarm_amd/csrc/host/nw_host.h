// nw_host.h — scalar global aligner used only by the uclust writer (-u), once per cluster
// member (the reference's nw(), src/nw.cc:237-255; out of the GPU path by design).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

struct swa_nw_scratch {
  std::vector<uint8_t> dir;
  std::vector<uint64_t> he;
  std::string ops;          // alignment operations, first column first ('M', 'I', 'D')
};

// dseq/qseq: 2-bit packed words.  Returns the number of non-identical columns and leaves the
// operations in scratch.ops.
uint64_t swa_nw_align(const uint64_t * dseq, uint32_t dlen, const uint64_t * qseq, uint32_t qlen, uint64_t mismatch,
                      uint64_t gapopen, uint64_t gapextend, swa_nw_scratch & scratch);

// run-length encoding with counts of 1 omitted (the reference's CIGAR flavour,
// src/utils/cigar.cc:28-60)
std::string swa_cigar(const std::string & ops);
