// hostdb.h — the host-resident amplicon database (what the reference keeps in the
// global seqindex[] / data_v blob, src/db.cc:65-67, src/utils/seqinfo.h:27-41).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/swarm_amd.h"
#include "../../../include/swarm_amd_host.h"

struct swa_hostdb {
  uint32_t n = 0;
  uint32_t longest = 0;
  uint32_t longest_header = 0;
  uint64_t nucleotides = 0;
  std::vector<uint64_t> seqs;        // packed words, db order, contiguous
  std::vector<uint64_t> seq_off;     // n + 1
  std::vector<uint32_t> seqlen;
  std::vector<uint64_t> abundance;
  std::vector<char> headers;         // NUL-terminated headers, db order
  std::vector<uint64_t> hdr_off;     // n + 1
  std::vector<int32_t> ab_start;     // abundance annotation span inside each header
  std::vector<int32_t> ab_end;
  std::string error;
};
