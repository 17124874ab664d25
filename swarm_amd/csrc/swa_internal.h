// swa_internal.h — shared by the HIP translation units of libswarm_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/swarm_amd.h"

// ---- device-side layout of one amplicon hash-table slot ----------------------------
// The reference keeps an occupied bitmap + u64 hash_values[] + u32 hash_data[]
// (src/hashtable.cc:41-44): three cache lines per probe step.  Here one 16-byte slot
// carries all three, so a probe step is ONE 16-B HBM/L2 transaction.
struct alignas(16) swa_slot {
  uint64_t hash;
  uint32_t amp;   // SWA_NO_AMPLICON = empty
  uint32_t pad;
};

// a growable device buffer (never shrinks; the db and tables stay resident in HBM)
struct swa_dbuf {
  void * ptr = nullptr;
  size_t bytes = 0;
};

struct swa_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cus = 256;
  std::string err;

  // optional per-kernel timing (HIP events on `stream`)
  bool timing = false;
  hipEvent_t ev[32] = {};
  bool ev_ready = false;
  bool ev_used[16] = {};

  // database (device pointers)
  swa_db_view db{};
  bool db_owned = false;
  swa_dbuf d_seqs, d_seq_off, d_seqlen, d_abund;

  // d=1 index
  bool d1_ready = false;
  uint64_t table_size = 0;       // slots, power of two
  uint64_t bloom_words = 0;      // u64 words in the amplicon Bloom
  uint32_t zobrist_len = 0;      // positions in the Zobrist table (longest + 2)
  uint32_t zobrist_resident = 0; // length of the table currently in d_zobrist (0 = none)
  bool patterns_resident = false;
  swa_dbuf d_zobrist, d_seqhash, d_table, d_bloom, d_patterns;
  // One 4 KB block holds every small status array of the context, so that the host reads them with ONE copy per phase (a
  // device-to-host copy of a few bytes is a kernel of its own, ~5 us with its gap: the step made nine):
  //   [0, 64) d_flags · [64, 192) d_stats · [192, 384) d_guard · [384, 512) d_extra: [0] last CSR offset [1] links the
  //   partition sorted · [1024, 2048) d_acounters.  d_flags / d_stats / d_guard / d_acounters are VIEWS into it (never freed
  //   on their own; their `bytes` is their room, so swa_reserve leaves them alone).
  swa_dbuf d_status;
  void * h_status = nullptr;     // 4 KB of pinned host memory: where the d = 1 step looks at the status block
  swa_dbuf d_flags;              // u32[16]: [0] duplicate flag
  swa_dbuf d_stats;              // u64[8] probe statistics + [8] edge counter
  swa_dbuf d_edges;              // u64 edge list (src << 32 | dst)
  swa_dbuf d_counts, d_cursor, d_scan_tmp, d_offsets_tmp, d_nb_tmp, d_long_rows;
  // anchored d=1 index (d1_anchor.inc, d1_stream.inc): [0] prefix groups, [1] suffix groups
  bool anchor_usable = false;    // decided by swa_d1_index_build: lengths fit, db order holds, not switched off
  bool anchor_ready = false;     // the streaming index below exists (for the owner in owner_rank / owner_world)
  bool full_index = false;       // d_seqhash / d_aux cover ALL amplicons and d_table / d_bloom are built (ensure_full_index)
  // routed build (swa_d1_index_build_routed): the members of this rank's groups arrive as id lists; and what is a fact
  // of the uploaded database rather than of one index build, kept until the next upload
  const uint32_t * route_ids[2] = {nullptr, nullptr};
  const unsigned long long * route_rec[2] = {nullptr, nullptr};   // routed by key records (swa_d1_index_build_records): no k_keys pass
  uint32_t route_m[2] = {0, 0};
  bool rank_ready = false;       // d_arank holds the abundance ranks of this database
  bool db_unordered = false;     // ... which is not in abundance order (the anchored passes are then not used)
  bool props_ready = false;      // db_shortest / class_pop below
  uint32_t db_shortest = 0;      // shortest sequence
  uint32_t class_pop[8] = {};    // sequences per width class (d1_anchor.inc: width_class; [kTooLong]: beyond the pair kernels)
  bool windows_ready = false;    // the anchor windows of this database are known (choose_anchor_windows + the safety net):
  uint32_t windows_chosen = 0;   // anchor_a = anchor_b = this
  uint32_t owner_rank = 0, owner_world = 1;   // swa_d1_set_ownership: this context serves the anchor groups of one rank
  uint32_t anchor_a = 0, anchor_b = 0;   // anchor windows moved inwards by this many nt ("window mode", chosen at index build)
  uint32_t anchor_w = 32;        // width of the anchor windows in nt: 32, 64 or 128 (wider: fewer pairs per group; needs 2 w + 1 nt)
  uint32_t windows_w = 32;       // ... as chosen for this database (with windows_chosen)
  swa_dbuf d_guard;              // u64[24] the guard's counters (d1.hip: guard_check)
  bool guard_index = false;      // [0..8) describe the index in place (made by the streaming build since the last clear)
  bool guard_keys_done = false;  // the key records of this upload have had their second opinion (k_guard_db / k_guard_records)
  bool guard_keys_pending = false;   // ... its sums, [16..22), wait for the next guard_check
  // d_akeys[0] / d_acounts[0]: scratch of the window sample; d_aitems: the work lists of the two indexes
  swa_dbuf d_aux, d_akeys[1], d_acounts[1], d_aitems[2];
  swa_dbuf d_acounters, d_afallback, d_arank, d_rank_tmp;
  swa_dbuf d_seg_fill;           // u32 fill of every per-wave edge segment
  swa_dbuf d_seg_base;           // u64 start of every segment in the compacted edge list (swa_d1_network_edges_device)
  uint64_t seg_cap = 0;          // entries per segment

  // q-gram / alignment state
  bool qgram_ready = false;
  swa_dbuf d_qgrams, d_list_a, d_list_b, d_list_c, d_list_d;
  uint64_t pen_mismatch = 18, pen_gapopen = 24, pen_gapextend = 13, resolution = 1;
  bool search_ready = false;
  uint32_t wfa_steps = 0;        // > 0: the wavefront alignment kernel is exact for the penalties / d in use
  uint32_t wfa_ring = 0;         // steps of history k_align_wfa keeps (the furthest a step looks back + 1)
  swa_dbuf d_wfa;

  // fused d >= 2 scan state (scan.hip)
  bool scan_ready = false;
  swa_dbuf d_scan_est, d_scan_swarmed, d_scan_targets, d_scan_diffs, d_scan_hits, d_scan_counters;
  swa_dbuf d_scan_cand;          // candidate list of the current swarm (scan.hip)
  swa_dbuf d_scan_seeds;         // seeds + limits of the current batch
  swa_dbuf d_scan_compares;      // q-gram comparison counts, one slot per workgroup
  uint32_t scan_cand_bound = 0;
  void * h_scan_pinned = nullptr;   // pinned, GPU-visible: seeds + limits in, hit mirror out
  std::vector<uint32_t> scan_host, scan_perm, scan_idx_tmp;
  std::vector<uint64_t> scan_sorted;
  uint64_t scan_pair_cap = 0;
  uint64_t scan_launches = 0;

  // fastidious state
  swa_dbuf d_light, d_graft, d_bloomflex, d_fpatterns, d_queue, d_fcounters;
  // pair route (d1_fast.inc): roles, group key table + counters, offsets, slot of every amplicon, member lists,
  // work items, the (heavy, light) pairs within two edits
  swa_dbuf d_frole, d_fkeys, d_fcnt, d_foff, d_fslot, d_fmembers, d_fitems, d_fpairs;
  uint64_t fast_pair_cap = 0;

  // d >= 2 in bulk (dn_graph.hip): the graph of all pairs within d differences, kept sorted on the device
  uint32_t dn_shortest = 0;      // shortest sequence of the database (0 = not measured yet)
  bool dn_graph_ready = false, dn_graph_ncb = false;
  uint64_t dn_pair_cap = 0, dn_comparisons = 0, dn_aligned = 0, dn_launches = 0, dn_edges = 0, dn_work = 0;
  swa_dbuf d_dn_keys, d_dn_vals;
  uint32_t dn_owner_rank = 0, dn_owner_world = 1;   // swa_dn_set_ownership: this context finds the pairs of the window groups it owns

  // streaming index build / CSR assembly (d1_stream.inc)
  bool lines_ready = false;      // d_lines holds this database's amplicon lines (made once per upload), lines_w words each
  uint32_t lines_quads = 0;      // ... of this many 16-byte quads each (4, 8 or 16)
  uint32_t list_counts[2 * 4 * 8] = {};   // items per work list of the index in place ([index][width class][8]: d_acounters + kCounterBase)
  bool list_counts_ready = false;
  uint64_t list_regions_items = 0;   // entries of an index's item buffer (d1.hip: list_regions)
  bool stream_index = false;     // the anchor indexes in place were made by the streaming build: members = ids in d_members
  uint32_t stream_extra_bits = 0;   // finer partition after a bucket held more distinct keys than the group kernel's table
  // [0] lines [1..4] records ping / pong per index [5, 6] fingerprints ping / pong [7] table slots of big buckets
  // [8, 9] flat counts [10, 11] tile tables [12, 13] chunk starts [14, 15] scan partials [16] scalars
  // [17, 18] members [19] oversized-group bits [20, 21] items per kind [22] link sort: records ping [23] pong
  // [24] buckets of the CSR stage left to whole workgroups [26] member table [27] member Bloom
  swa_dbuf d_stream[30];
  // member index: hash table + Bloom of the members of oversized groups only (what the plain kernel probes for them)
  bool member_index = false, only_oversized = false;
  uint32_t over_mass = 0;
  uint64_t mtable_size = 0, mbloom_words = 0;

  // the d = 1 network kept in d_offsets_tmp / d_nb_tmp (swa_d1_network_resident) and its clustering (cluster_gpu.hip)
  bool csr_ready = false;
  uint32_t index_first = 0, index_count = 0;   // the range of the last index build (network_run_guarded repeats it)
  bool index_routed = false;
  uint32_t guard_retries = 0;                  // steps repeated after the guard found counts that did not balance
  swa_dbuf d_words_stage;                      // swa_db_stage_words: the reader's word pools, file order
  const void * staged_first = nullptr; uint64_t staged_words = 0;
  bool cluster_ready = false;                  // swa_d1_cluster_device's arrays lie in d_cluster (swa_d1_cluster_fetch)
  uint32_t cluster_maxgen = 0;
  int pair_blocks[4] = {};                      // workgroups of k_d1_group_pairs a CU holds, per width class (0: not asked yet)
  bool g1_lds_opt_in = false;                  // k_group1's dynamic-LDS attribute has been set on this context's device
  uint32_t part_lds_opt_in = 0;    // ... and the wide-tile forms of k_part_scatter (one bit each)
  uint32_t pair_lds_opt_in = 0;    // ... and the W = 15 / 21 forms of k_d1_group_pairs (pass + 2 (W = 21) + 4 (NW = 2) + 8 (NW = 4))
  bool csr_has_diffs = false;                 // the resident network is a d >= 2 graph: one byte of differences per link behind the neighbours
  uint64_t csr_total = 0;
  swa_dbuf d_cluster, d_cluster_ctl;
};

int swa_fail(swa_ctx * ctx, int code, const char * what, hipError_t e);
int swa_fail_msg(swa_ctx * ctx, int code, const std::string & msg);
// SWARM_AMD_STEP_TIMING=1: wall-clock laps between the host-visible points of the d = 1 index build and network call, on
// stderr (what a first step in a fresh process is made of; the device is synchronised at each lap, so the numbers are
// not the pipelined step's)
inline void swa_lap(swa_ctx * ctx, const char * what);
int swa_reserve(swa_ctx * ctx, swa_dbuf & buf, size_t bytes);   // grow-only hipMalloc
void swa_release(swa_dbuf & buf);
int swa_hash_sequences(swa_ctx * ctx);                           // d1.hip: Zobrist table + d_seqhash + d_aux
// dn_graph.hip, for multi.hip: the (partial, under ownership) graph computed and left sorted in HBM — d_dn_keys / d_dn_vals
// from entry dn_work on, dn_edges entries —, and a sorted (query << 32 | target, diff) list written out as the CSR of swa_dn_graph
int swa_dn_graph_compute(swa_ctx * ctx, int no_cluster_breaking);
int swa_dn_graph_emit(swa_ctx * ctx, const unsigned long long * sorted, const uint32_t * svals, uint64_t nedges, uint64_t * offsets,
                      uint32_t * neighbours, uint8_t * diffs, uint64_t cap, uint64_t * total);

// d1.hip, for multi.hip: CSR of the whole database from link lists gathered on this context's device (partition + row kernels)
int swa_d1_csr_from_lists(swa_ctx * ctx, const unsigned long long * d_links, const uint64_t * starts, const uint64_t * counts, uint32_t lists,
                          uint64_t * d_offsets, uint32_t * d_neighbours, uint64_t cap);

// RAII-less timing brackets: SWA_T0(ctx, slot) ... SWA_T1(ctx, slot)
inline void swa_t0(swa_ctx * ctx, int slot) {
  if (ctx->timing && ctx->ev_ready) { (void)hipEventRecord(ctx->ev[2 * slot], ctx->stream); }
}
inline void swa_t1(swa_ctx * ctx, int slot) {
  if (ctx->timing && ctx->ev_ready) { (void)hipEventRecord(ctx->ev[2 * slot + 1], ctx->stream); ctx->ev_used[slot] = true; }
}

#define SWA_HIP(ctx, expr)                                                     \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) return swa_fail((ctx), SWA_E_DEVICE, #expr, e_);     \
  } while (0)

#define SWA_TRY(expr)                  \
  do {                                 \
    int rc_ = (expr);                  \
    if (rc_ != SWA_OK) return rc_;     \
  } while (0)

// host-side table generators (host_tables.cpp): bit-identical to the reference's
// first zobrist_init()/bloom_init()/bloomflex_init() of a process
void swa_zobrist_table(uint32_t zobrist_len, std::vector<uint64_t> & tab);    // src/zobrist.cc:49-80
void swa_bloom_patterns(uint32_t count, uint32_t k, std::vector<uint64_t> & pat); // src/bloompat.cc:74-90, src/bloomflex.cc:72-88
uint64_t swa_hashtable_size(uint64_t n);                                       // src/utils/hashtable_size.cc:29-42

// ---- device helpers -------------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ unsigned swa_nt(const uint64_t * seq, uint32_t pos) {
  return (unsigned)((seq[pos >> 5] >> ((pos & 31u) << 1)) & 3u);
}

__device__ __forceinline__ uint64_t swa_shfl_u64(uint64_t v, int src) {
  const int lo = __shfl((int)(uint32_t)v, src, 64);
  const int hi = __shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t swa_shfl_xor_u64(uint64_t v, int mask) {
  const int lo = __shfl_xor((int)(uint32_t)v, mask, 64);
  const int hi = __shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t swa_shfl_up_u64(uint64_t v, unsigned d) {
  const int lo = __shfl_up((int)(uint32_t)v, d, 64);
  const int hi = __shfl_up((int)(uint32_t)(v >> 32), d, 64);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t swa_shfl_down_u64(uint64_t v, unsigned d) {
  const int lo = __shfl_down((int)(uint32_t)v, d, 64);
  const int hi = __shfl_down((int)(uint32_t)(v >> 32), d, 64);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// Word `w` of the microvariant of `seed` (len nt, nw = ceil(len/32) valid words,
// zero padded) described by (type, pos, base): what generate_variant_sequence
// (src/variants.cc:78-115) materialises nucleotide by nucleotide, computed here
// with 64-bit shifts.  type: 0 substitution, 1 deletion, 2 insertion.
__device__ __forceinline__ uint64_t swa_variant_word(const uint64_t * seed, uint32_t nw, uint32_t type,
                                                     uint32_t pos, uint32_t base, uint32_t w) {
  const uint32_t wp = pos >> 5;
  const uint32_t sh = (pos & 31u) << 1;
  const uint64_t cur = (w < nw) ? seed[w] : 0ull;
  if (type == 0u) {
    if (w != wp) return cur;
    return (cur & ~(3ull << sh)) | ((uint64_t)base << sh);
  }
  const uint64_t lowmask = (1ull << sh) - 1ull;   // nts below `pos` inside word wp (sh < 64)
  if (type == 1u) {
    const uint64_t nxt = (w + 1 < nw) ? seed[w + 1] : 0ull;
    const uint64_t shifted = (cur >> 2) | (nxt << 62);      // every nt one position down
    if (w < wp) return cur;
    if (w > wp) return shifted;
    return (cur & lowmask) | (shifted & ~lowmask);
  }
  {
    const uint64_t prv = (w >= 1 && w - 1 < nw) ? seed[w - 1] : 0ull;
    const uint64_t shifted = (cur << 2) | (prv >> 62);      // every nt one position up
    if (w < wp) return cur;
    if (w > wp) return shifted;
    return (cur & lowmask) | ((uint64_t)base << sh) | (shifted & ~(lowmask | (3ull << sh)));
  }
}

#endif  // __HIPCC__

// A caller's result buffer about to receive a large download: its pages are faulted in by the library's worker threads first
// (one zero byte a page; the copy overwrites the whole range right after).  The runtime's copy into pageable pages that do not
// exist yet — a freshly allocated / calloc'ed array — faults them in one by one on its own thread: 8-13 GB/s instead of the
// 40-55 of touched pages (tools/experiments/d2h_cost.hip, profiles/r06).  Costs ~10 us per MB when the pages exist already.
#include "host/pool.h"
inline void swa_touch_pages(void * ptr, size_t bytes) {
  if (ptr == nullptr || bytes < (size_t(4) << 20)) { return; }
  char * p = static_cast<char *>(ptr);
  const size_t pages = (bytes + 4095) / 4096;
  const unsigned parts = std::min<unsigned>(swa_pool::get().size(), 16u);
  swa_pool::get().run(parts, [&](unsigned t) {
    for (size_t k = pages * t / parts; k < pages * (t + 1) / parts; ++k) {
      // (a plain write: the whole range is overwritten by the copy that follows.  Reading the byte first — "keep what is
      // there" — makes an untouched page a read fault onto the shared zero page and then a copy-on-write, i.e. an
      // invalidation every MMU notifier of the process hears: measured, 30 ms more for 160 MB under the runtime's notifiers)
      p[k * 4096] = 0;
    }
  });
}

#include <chrono>
inline void swa_lap(swa_ctx * ctx, const char * what) {
  static const bool on = std::getenv("SWARM_AMD_STEP_TIMING") != nullptr;
  if (!on) { return; }
  static auto last = std::chrono::steady_clock::now();
  (void)hipStreamSynchronize(ctx->stream);
  const auto now = std::chrono::steady_clock::now();
  std::fprintf(stderr, "[step] %-34s %8.3f ms\n", what, 1e3 * std::chrono::duration<double>(now - last).count());
  last = std::chrono::steady_clock::now();
}
