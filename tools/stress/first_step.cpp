// tools/stress/first_step.cpp — ONE fresh process, ONE first d = 1 step, checksums of every stage on one line.
//
// The anomaly this hunts (DESIGN.md "the anomaly"): twice in three rounds the FIRST run of an index
// build on a fresh box lost the links of a few wavefronts' worth of amplicons, silently, and never again in later
// processes.  tools/stress/run.sh starts this program a few hundred times per lease — fresh process, fresh HIP runtime,
// first launch of every kernel — under the runtime conditions only a first run has (context warm-up on a helper thread
// or not, AMD_SERIALIZE_KERNEL on or off, streaming or table build) and compares the lines: a difference names the stage.
//
//   first_step in.fa [expected_csr_hex]      exit 0: ran (and matched, if expected given); 3: differs; 4: library error
//
// Stages: lines (the amplicon lines, exact), members0/1 (multiset of ids per index), sizes0/1 (multiset of group sizes
// per index: listed groups), csr (offsets + sorted neighbours, exact), links (count).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/swarm_amd.h"
#include "../../include/swarm_amd_host.h"

static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
static uint64_t chain(const void * p, size_t bytes) {          // order-dependent
  const auto * w = static_cast<const uint64_t *>(p);
  uint64_t h = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < bytes / 8; ++i) { h = mix(h ^ w[i]); }
  const auto * b = static_cast<const uint8_t *>(p);
  for (size_t i = bytes & ~size_t(7); i < bytes; ++i) { h = mix(h ^ b[i]); }
  return h;
}

int main(int argc, char ** argv) {
  if (argc < 2) { fprintf(stderr, "usage: first_step in.fa [expected_csr_hex]\n"); return 2; }
  const char * env_warm = getenv("STRESS_WARMUP");
  const bool warm = !(env_warm != nullptr && env_warm[0] == '0');
  swa_ctx * ctx = nullptr;
  if (swa_ctx_create(0, nullptr, &ctx) != SWA_OK) { fprintf(stderr, "no device\n"); return 4; }
  std::thread helper;
  if (warm) { helper = std::thread([&]() { (void)swa_ctx_warmup(ctx); }); }   // (as the command line does it: beside the FASTA read)
  swa_hostdb * db = nullptr;
  if (swa_hostdb_read_fasta(argv[1], 0, 0, 0, &db) != SWA_OK) { fprintf(stderr, "fasta: %s\n", swa_hostdb_error(db)); return 4; }
  if (helper.joinable()) { helper.join(); }
  swa_db_view v{};
  swa_hostdb_view(db, &v);
  const uint32_t n = v.n;
  int dup = 0;
  int rc = swa_db_upload(ctx, &v);
  if (rc == SWA_OK) { rc = swa_d1_index_build(ctx, &dup); }
  std::vector<uint64_t> off((size_t)n + 1);
  std::vector<uint32_t> nb((size_t)8 * n + 1024);
  uint64_t total = 0;
  if (rc == SWA_OK) { rc = swa_d1_network(ctx, 0, 0, n, off.data(), nb.data(), nb.size(), &total); }
  if (rc != SWA_OK) { printf("ERROR rc=%d %s\n", rc, swa_last_error(ctx)); return 4; }
  uint64_t h_lines = 0, h_members[2] = {0, 0}, h_sizes[2] = {0, 0};
  uint32_t aw[2] = {0, 0};
  (void)swa_d1_anchor_windows(ctx, aw);
  {
    // where the lists lie and how wide a line is (selector 16): u64 [4 width classes][7 size kinds + end], items in all, quads per line
    uint64_t layout[4 * 8 + 2] = {};
    const bool have_layout = swa_d1_debug_read(ctx, 16, layout, sizeof(layout)) == SWA_OK;
    const size_t line_bytes = have_layout ? (size_t)layout[33] * 16 : 0;
    std::vector<uint8_t> lines((size_t)n * line_bytes + 16);
    if (have_layout && swa_d1_debug_read(ctx, 15, lines.data(), (size_t)n * line_bytes) == SWA_OK) { h_lines = chain(lines.data(), (size_t)n * line_bytes); }
    std::vector<uint32_t> counters(256);
    const bool have_counters = have_layout && swa_d1_debug_read(ctx, 14, counters.data(), counters.size() * 4) == SWA_OK;
    for (int which = 0; which < 2 && have_counters; ++which) {
      std::vector<uint32_t> members((size_t)n + 2);
      if (swa_d1_debug_read(ctx, 10 + which, members.data(), (size_t)n * 4) != SWA_OK) { continue; }
      for (uint32_t i = 0; i < n; ++i) { h_members[which] += mix(members[i] + 1ull); }
      std::vector<uint32_t> items((size_t)layout[32] * 3 + 4);
      if (swa_d1_debug_read(ctx, 12 + which, items.data(), (size_t)layout[32] * 12) != SWA_OK) { continue; }
      for (int cls = 0; cls < 4; ++cls) {
        for (int k = 0; k < 6; ++k) {
          const uint32_t cnt = counters[64 + (which * 4 + cls) * 8 + k];
          for (uint32_t j = 0; j < cnt; ++j) { h_sizes[which] += mix(items[(layout[cls * 8 + k] + j) * 3 + 1] + 77ull); }
        }
      }
    }
  }
  const uint64_t h_csr = mix(chain(off.data(), off.size() * 8) ^ chain(nb.data(), (size_t)total * 4));
  printf("n=%u w=%u win=%u,%u dup=%d lines=%016llx members0=%016llx members1=%016llx sizes0=%016llx sizes1=%016llx links=%llu csr=%016llx\n", n,
         swa_d1_anchor_width(ctx), aw[0], aw[1], dup, (unsigned long long)h_lines, (unsigned long long)h_members[0], (unsigned long long)h_members[1],
         (unsigned long long)h_sizes[0], (unsigned long long)h_sizes[1], (unsigned long long)total, (unsigned long long)h_csr);
  int status = 0;
  if (argc > 2) {
    const uint64_t want = strtoull(argv[2], nullptr, 16);
    if (want != h_csr) { printf("DIFFERENT csr: want %016llx\n", (unsigned long long)want); status = 3; }
  }
  fflush(stdout);
  _Exit(status);                                               // (like the command line: no teardown of the runtime)
}
