// oracle/gpu_seams/gpu_bridge.cc — TEST INFRASTRUCTURE ONLY.
//
// The binding INTEGRATION.md describes, written out and compiled: the unmodified reference sources + this file + two
// include files spliced in at the d = 1 seams (b1.inc, b2.inc; `make -C oracle ref-gpu`) give oracle/_ref/swarm_gpu, the
// REFERENCE program with its four hot seams bound to libswarm_amd.so.  tests/test_ref_gpu.py runs it over the golden
// cases on the GPU box: the C ABI is bindable as claimed, by construction rather than by prose.
// Nothing here is part of the product; no reference source text is stored in this repository (the seams are spliced
// into a scratch copy under oracle/_ref/ by line number, patch_seams.py).
#include "swarm_amd.h"

#include "db.h"
#include "qgram.h"
#include "scan.h"
#include "utils/alignment_parameters.h"
#include "utils/nt_codec.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static swa_ctx * g_gpu = nullptr;
static int64_t g_differences = 1;
static bool g_qgrams_built = false, g_search_begun = false;

// seam L2: the packed database, realigned into the SoA form, once per run
swa_ctx * gpu_bridge_ctx() {
  if (g_gpu != nullptr) { return g_gpu; }
  const unsigned int n = db_getsequencecount();
  std::vector<uint64_t> words, off((size_t)n + 1), abund(n);
  std::vector<uint32_t> len(n);
  for (unsigned int i = 0; i < n; ++i) {
    off[i] = words.size();
    len[i] = db_getsequencelen(i);
    abund[i] = db_getabundance(i);
    const size_t nw = ((size_t)len[i] + 31) / 32;
    words.resize(words.size() + nw, 0);
    std::memcpy(&words[off[i]], db_getsequence(i), nt_bytelength(len[i]) < nw * 8 ? nt_bytelength(len[i]) : nw * 8);
  }
  off[n] = words.size();
  words.push_back(0); words.push_back(0);
  const char * dev = std::getenv("SWARM_AMD_DEVICE");
  if (swa_ctx_create(dev != nullptr ? std::atoi(dev) : 0, nullptr, &g_gpu) != SWA_OK) {
    std::fprintf(stderr, "\nError: no usable gfx950 GPU\n");
    std::exit(1);
  }
  swa_db_view v{n, db_getlongestsequence(), words.data(), off.data(), len.data(), abund.data()};
  if (swa_db_upload(g_gpu, &v) != SWA_OK) {
    std::fprintf(stderr, "\nError: %s\n", swa_last_error(g_gpu));
    std::exit(1);
  }
  return g_gpu;
}

void gpu_bridge_configure(int64_t differences) { g_differences = differences; }

// seam B3 (src/qgram.h:31-35; the reference's own definition is compiled under another name)
auto qgram_diff_fast(uint64_t seed, uint64_t listlen, uint64_t * amplist, uint64_t * difflist,
                     std::vector<struct thread_info_s> & thread_info_v) -> void {
  (void)thread_info_v;
  swa_ctx * gpu = gpu_bridge_ctx();
  if (!g_qgrams_built) {
    if (swa_qgram_build(gpu) != SWA_OK) { std::fprintf(stderr, "\nError: %s\n", swa_last_error(gpu)); std::exit(1); }
    g_qgrams_built = true;
  }
  if (listlen != 0 && swa_qgram_diff(gpu, seed, listlen, amplist, difflist) != SWA_OK) {
    std::fprintf(stderr, "\nError: %s\n", swa_last_error(gpu));
    std::exit(1);
  }
}

// seam B4 (src/scan.h:30-37)
auto search_do(uint64_t query_no, uint64_t listlength, uint64_t * targets, uint64_t * scores, uint64_t * diffs,
               uint64_t * alignlengths, int bits, ThreadRunner * search_threads) -> void {
  (void)bits; (void)search_threads;
  swa_ctx * gpu = gpu_bridge_ctx();
  if (!g_search_begun) {
    if (swa_search_begin(gpu, (uint64_t)penalty_mismatch, (uint64_t)penalty_gapopen, (uint64_t)penalty_gapextend,
                         (uint64_t)g_differences) != SWA_OK) {
      std::fprintf(stderr, "\nError: %s\n", swa_last_error(gpu));
      std::exit(1);
    }
    g_search_begun = true;
  }
  if (listlength != 0 && swa_search_do(gpu, query_no, listlength, targets, scores, diffs, alignlengths) != SWA_OK) {
    std::fprintf(stderr, "\nError: %s\n", swa_last_error(gpu));
    std::exit(1);
  }
}
