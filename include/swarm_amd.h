/*
 * swarm_amd.h — C ABI of the MI355X-native amplicon neighbour-finding library
 * (libswarm_amd.so, built from swarm_amd/csrc/ with hipcc --offload-arch=gfx950).
 *
 * The reference (torognes/swarm 3.1.6) has no plugin/FFI interface; its hot path is
 * reached through four internal C++ call seams.  This header is the
 * thin C ABI a maintainer would bind at exactly those seams: plain pointers and
 * sizes, caller-allocated result buffers (as in the reference, where algo_d1_run /
 * algo_run own every vector: src/algod1.cc:1104-1125, src/algo.cc:352-361), int
 * status instead of fatal()+exit(1) (src/utils/fatal.cc:27-31).
 *
 *   seam  reference interface replaced                               entry point
 *   ----  ---------------------------------------------------------  --------------------------
 *   L2    global seqindex[] read through db_get*() (src/db.h:35-53)  swa_db_upload / swa_db_attach
 *   B1    hash_insert loop + network_thread/check_variants           swa_d1_index_build
 *         (src/algod1.cc:188-208, 606-670, 1122-1171)                swa_d1_network[_device]
 *   B2    mark_light_thread / check_heavy_thread                     swa_d1_fastidious
 *         (src/algod1.cc:453-552, 1411-1467)
 *   B3    db_qgrams_init + qgram_diff_fast (src/db.cc:819-842,       swa_qgram_build
 *         src/qgram.h:31-35)                                         swa_qgram_diff
 *   B4    search_begin + search_do (src/scan.h:30-37, 259)           swa_search_begin / swa_search_do
 *
 * Amplicon numbering everywhere = the reference's db order after db_read():
 * abundance descending, then header ascending (src/db.cc:388-413).  Sequences are
 * 2-bit packed exactly as the reference packs them (A0 C1 G2 T/U3, 32 nt per
 * little-endian u64, LSB first, zero padded: src/db.cc:541-628,
 * src/utils/nt_codec.cc:35-75) but 8-byte aligned, one amplicon after the other.
 *
 * All functions are synchronous with respect to the caller unless the name ends in
 * _device (those only enqueue on the context's HIP stream and leave results in HBM).
 * One swa_ctx is bound to one GPU and one HIP stream; use one context per GPU (one
 * process per GPU, or one host thread per GPU).  The library never falls back to the
 * CPU: without a usable gfx950 device every call fails with SWA_E_DEVICE.
 */
#ifndef SWARM_AMD_H
#define SWARM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWA_ABI_VERSION 1

enum {
  SWA_OK = 0,
  SWA_E_DEVICE = 1,      /* no GPU / HIP runtime error (see swa_last_error) */
  SWA_E_ARG = 2,         /* invalid argument or call order */
  SWA_E_NOMEM = 3,       /* hipMalloc / host allocation failed */
  SWA_E_CAPACITY = 4,    /* caller's result buffer too small; *total tells the need */
  SWA_E_DUPLICATES = 5,  /* identical sequences present (reference: fatal, src/algod1.cc:1141-1150) */
  SWA_E_INTERNAL = 6     /* the d = 1 guard: index / network counts that must balance do not (never a short network instead) */
};

#define SWA_NO_AMPLICON 0xFFFFFFFFu   /* the reference's no_swarm (src/algod1.cc:80) */

typedef struct swa_ctx swa_ctx;

/* The packed amplicon database, n amplicons in db order.  Host or device pointers
   depending on the call that takes it. */
typedef struct swa_db_view {
  uint32_t n;                  /* amplicons */
  uint32_t longest;            /* longest sequence (nt) */
  const uint64_t * seqs;       /* packed words, amplicon i at words [seq_off[i], seq_off[i+1]) */
  const uint64_t * seq_off;    /* n+1 word offsets; seq_off[i+1]-seq_off[i] >= ceil(seqlen[i]/32) */
  const uint32_t * seqlen;     /* n lengths, nt (>= 1) */
  const uint64_t * abundance;  /* n abundances (>= 1) */
} swa_db_view;

/* ---- context ------------------------------------------------------------------ */
int  swa_abi_version(void);
/* device: HIP ordinal.  stream: a hipStream_t to run on (NULL -> the context creates
   its own non-blocking stream). */
int  swa_ctx_create(int device, void * stream, swa_ctx ** out);
void swa_ctx_destroy(swa_ctx * ctx);
/* message of the last failing call on this context ("" if none) */
const char * swa_last_error(const swa_ctx * ctx);
/* blocks until everything enqueued on the context's stream has finished */
int  swa_ctx_synchronize(swa_ctx * ctx);
/* Optional: pays the device's first-use costs (memory pools, copy queues, code-object loads) now — e.g. on a helper
   thread while the caller reads its input. */
int  swa_ctx_warmup(swa_ctx * ctx);
/* The same for a caller that knows its d: 1 loads the d = 1 step's and the clustering's code objects, >= 2 those of the q-gram,
   pair-graph and alignment kernels (and the clustering's), anything else all of them. */
int  swa_ctx_warmup_for(swa_ctx * ctx, int differences);
/* the copy engine's first download (~7.5 ms of runtime set-up whatever the size), paid ahead: any thread, beside the
   other warm-up calls */
int  swa_ctx_warmup_downloads(swa_ctx * ctx);

/* Per-kernel timing with HIP events on the context's stream (off by default).
   swa_timing_read: ms[0] seqhash, [1] table+Bloom build, [2] duplicate check,
   [3] d1 network kernels (anchored passes + fallback, or the plain kernel), [4] CSR assembly,
   [5] fastidious light pass, [6] fastidious heavy pass, [7] anchored-index build — durations
   of the most recent launches. */
int  swa_timing_enable(swa_ctx * ctx, int on);
int  swa_timing_read(swa_ctx * ctx, float * ms8);
/* the kernel groups of the streaming d = 1 step (swarm_amd/csrc/d1_stream.inc), same mechanism: ms[0] amplicon keys,
   [1] partition of the key records, [2] groups + work lists (+ identical sequences), [3] pair kernels of the prefix
   groups, [4] of the suffix groups, [5] partition of the links by source, [6] CSR rows, [7] amplicon lines (once per
   uploaded database) — durations of the most recent launches, 0 for a group that did not run. */
int  swa_timing_read_stream(swa_ctx * ctx, float * ms8);

/* ---- L2: database residency ---------------------------------------------------- */
/* copy a host-resident database into HBM (replicated on this context's GPU) */
int swa_db_upload(swa_ctx * ctx, const swa_db_view * host_db);
/* adopt a database whose arrays are ALREADY device memory on this GPU (not copied,
   not freed; must outlive the context's use of it) */
int swa_db_attach(swa_ctx * ctx, const swa_db_view * device_db);
/* The same database handed over as the FASTA reader leaves it (src/db.cc:432-628 packs while it reads; :388-413 sorts
   afterwards): the packed words in FILE order, in `pieces` pools, and per amplicon k of the db order where its words
   begin in the concatenation of the pools.  The GPU puts the words in db order (one gather kernel) — the host never
   copies them.  swa_db_stage_words starts the copy of the pools as soon as they exist (asynchronous, on the context's
   stream; the pools must stay as they are until swa_db_upload_unordered has returned); swa_db_upload_unordered stages
   them itself if that was not done (same pool pointers), copies the three per-amplicon arrays and leaves the context
   exactly as swa_db_upload does. */
typedef struct swa_db_unordered_view {
  uint32_t n;                          /* amplicons */
  uint32_t longest;                    /* longest sequence (nt) */
  uint32_t pieces;                     /* word pools */
  const uint64_t * const * piece_words; /* pools[pieces]: packed words, every sequence from a word boundary */
  const uint64_t * piece_word_count;   /* words per pool */
  const uint64_t * src_off;            /* [n] db order: first word of amplicon k, counted through the pools in order */
  const uint32_t * seqlen;             /* [n] db order */
  const uint64_t * abundance;          /* [n] db order */
} swa_db_unordered_view;
int swa_db_stage_words(swa_ctx * ctx, const uint64_t * const * piece_words, const uint64_t * piece_word_count, uint32_t pieces);
int swa_db_upload_unordered(swa_ctx * ctx, const swa_db_unordered_view * host_db);
/* Result buffers are the caller's (as in the reference: src/algod1.cc:1104-1125).  A caller that wants its downloads — the
   member order of swa_d1_cluster_device, the lists of swa_d1_network — at the link's speed pins them first: the copy is then
   one DMA instead of a staged copy through the runtime's bounce buffers.  Any thread; the memory stays the caller's
   (unpin before freeing it). */
int  swa_host_pin(swa_ctx * ctx, void * ptr, size_t bytes);
void swa_host_unpin(void * ptr);

/* ---- B1: d = 1 network --------------------------------------------------------- */
/* The index the network calls work on — what the reference's hash_insert loop builds (src/algod1.cc:188-208,
   1122-1150).  IDENTICAL SEQUENCES (the reference: fatal inside that loop, src/algod1.cc:1131-1150) are reported with
   SWA_E_DUPLICATES by whichever call meets them (round 6): the index build when it builds a table (*has_duplicates != 0:
   the database-wide table, or the table of the members left to the plain kernel) — and otherwise the NETWORK call, whose
   prefix pass compares the members of every group with one another anyway and sees two whose every word agrees for two
   instructions a pair (rounds 3-5 ran a second hash table over sequence fingerprints inside the index build for that: a
   quarter of its largest kernel).  A binding checks both return codes; the message is the reference's either way.
   What is built depends on the database: sequences of 65..256 nt in abundance order get the two
   anchor indexes of the streaming build (amplicons grouped by their first / last w nt, w = 32, 64 or 128: swa_d1_anchor_width; members and work lists
   in one pass over the amplicon lines: swarm_amd/csrc/d1_stream.inc) and nothing else; the
   database-wide structures of the reference — Zobrist table (bit-identical, src/zobrist.cc:49-80), seqhash[]
   (src/db.cc:761), amplicon hash table + Bloom filter (src/hashtable.cc, src/bloompat.cc) — are built only for what
   needs them: sequences under 65 nt, groups too large for the pair kernels, a database not in abundance order, the
   debug readers. */
int swa_d1_index_build(swa_ctx * ctx, int * has_duplicates);
/* Range form: the index covers the whole database as above; a table-based duplicate check looks only at the amplicons of
   [first, first + count) (their twin may be anywhere in the database).  On the pair route the network call reports a
   pair of identical sequences when one of the two is a seed of that call — [first, first + count) of swa_d1_network*,
   or, under swa_d1_set_ownership, any member of a group the rank owns: over the ranks every such pair is met by exactly
   one of them; OR the codes. */
int swa_d1_index_build_range(swa_ctx * ctx, uint32_t first, uint32_t count, int * has_duplicates);
/* Where the last index build put the two anchor windows that group the amplicons: out2 = {nt from the start, nt from
   the end}; (0, 0) = the first / last w nt, moved inwards (32-nt windows then) when those are (nearly) the same for everybody. */
int swa_d1_anchor_windows(const swa_ctx * ctx, uint32_t * out2);
/* ... and how wide it made them, in nucleotides: 32, 64 or 128 — the widest that leaves the shortest sequence of the
   database two windows and a nucleotide.  Two sequences of at least 2 w + 1 nt one edit apart share their first w or
   their last w nucleotides for any w; the reference has no such notion (it enumerates: src/variants.cc:184-249). */
uint32_t swa_d1_anchor_width(const swa_ctx * ctx);
/* Multi-GPU by ownership (no reference counterpart: src/algod1.cc:641-669 splits the seeds over
   threads that share one table).  With world > 1 this context serves only its share of the
   probes: the anchor groups (amplicons sharing their first / last w nucleotides) whose key maps
   to `rank` — with ALL their members, so the groups' LDS tables are built once per job instead
   of once per rank —, its share of the seeds only the plain kernel can serve, and, when the
   anchored route is not in use, the seeds with id mod world == rank.  swa_d1_network[_device]
   over a range then returns the PARTIAL rows of that range; over all ranks every link of the
   network appears exactly once (the host redistributes them: swarm_amd/sharding.py, and
   swa_d1_network_edges_device below for the form that travels).
   world = 1 restores the complete network.  Takes effect at the next network call.
   An index build made under ownership contains only what the rank's groups need: hashes of their
   members, duplicates found inside the owned prefix groups (identical sequences share one, so
   over all ranks every duplicate pair is reported by exactly one of them — OR the flags), and
   the database-wide table + Bloom only if some seed needs the plain kernel; after changing the
   owner call swa_d1_index_build[_range] again. */
int swa_d1_set_ownership(swa_ctx * ctx, uint32_t rank, uint32_t world);

/* Routed index build of a multi-GPU job: no rank walks the whole (replicated) database any more.
   (The reference has no counterpart: its threads share one table, src/algod1.cc:188-208; this is the distributed
   form of that hash_insert loop.)
   1. every rank keys ITS slice [first, first + count) of the amplicons: swa_d1_route_slice leaves, per anchor index
      (0 = prefix side, 1 = suffix side) and per owning rank, the ids of the slice whose key that rank owns —
      d_ids[(index * world + owner) * cap + ..], d_counts[index * world + owner]; d_counts[2 * world] != 0 reports a
      region that was too small (cap >= 1.5 * count / world + 1024 never is, ownership being hashed).  Device
      pointers; the lists and counts are complete when the call returns.
   2. the ranks exchange the lists all-to-all (RCCL / torch.distributed: 8 bytes per amplicon in total);
   3. every rank builds its indexes from what it received: swa_d1_index_build_routed(ids of index 0, of index 1) —
      as swa_d1_index_build under swa_d1_set_ownership(rank, world), which must have been called, but from the lists.
      Databases the anchored passes cannot serve alone (sequences under 65 nt, oversized groups, not in abundance
      order) take the database-wide route inside the call, as always. */
int swa_d1_route_slice(swa_ctx * ctx, uint32_t first, uint32_t count, uint32_t world, uint32_t * d_ids, uint64_t cap,
                       uint32_t * d_counts);
int swa_d1_index_build_routed(swa_ctx * ctx, const uint32_t * d_ids_prefix, uint32_t n_prefix, const uint32_t * d_ids_suffix,
                              uint32_t n_suffix, int * has_duplicates);
/* The same three steps with the KEY RECORDS travelling instead of the ids (round 6; what the multi-GPU drivers use): the rank
   that holds an amplicon's slice computes its key record (record key << 32 | id: 8 bytes) per index — a streaming pass over
   its own share of the packed database —, and the owner of the key starts at the partition: it no longer fetches one random
   64-byte line per received id out of a line array N times its share.  d_records[(index * world + owner) * cap + ..],
   d_counts as above.  8 + 8 bytes per amplicon through the exchange instead of 4 + 4. */
int swa_d1_route_slice_records(swa_ctx * ctx, uint32_t first, uint32_t count, uint32_t world, uint64_t * d_records,
                               uint64_t cap, uint32_t * d_counts);
int swa_d1_index_build_records(swa_ctx * ctx, const uint64_t * d_records_prefix, uint32_t n_prefix,
                               const uint64_t * d_records_suffix, uint32_t n_suffix, int * has_duplicates);

/* Neighbour lists of amplicons [first, first+count) as CSR: offsets[count+1],
   neighbours[offsets[count]]; row k = { j != first+k : seq_j is a microvariant of
   seq_(first+k) and (no_cluster_breaking or abundance[first+k] >= abundance[j]) },
   each neighbour once, ascending.  cap = capacity of `neighbours` in entries;
   *total = entries needed.  SWA_E_CAPACITY if total > cap (offsets still valid).
   SWA_E_DUPLICATES: two identical sequences among the seeds' groups (see swa_d1_index_build): like the reference, no
   network then. */
int swa_d1_network(swa_ctx * ctx, int no_cluster_breaking, uint32_t first, uint32_t count,
                   uint64_t * offsets, uint32_t * neighbours, uint64_t cap, uint64_t * total);
/* same, results left in HBM: d_offsets / d_neighbours are device pointers; only
   enqueues + one 8-byte readback of *total */
int swa_d1_network_device(swa_ctx * ctx, int no_cluster_breaking, uint32_t first, uint32_t count,
                          uint64_t * d_offsets, uint32_t * d_neighbours, uint64_t cap, uint64_t * total);
/* same links as one flat list in HBM: d_edge_list[i] = source << 32 | target, each link once, in
   no particular order (what the kernels emit before the CSR is assembled).  This is the form a
   multi-GPU job exchanges (swa_d1_set_ownership): nothing in it is proportional to the database
   size.  cap = capacity in entries; SWA_E_CAPACITY / *total as above. */
int swa_d1_network_edges_device(swa_ctx * ctx, int no_cluster_breaking, uint32_t first, uint32_t count,
                                uint64_t * d_edge_list, uint64_t cap, uint64_t * total);
/* The guard (SWA_E_INTERNAL).  The reference's network thread cannot return a partial network
   (src/algod1.cc:630-670); the three calls above compare the counts their kernels hand to one another and,
   should they not balance, make everything derived from the uploaded database again and repeat the step ONCE
   (one line on stderr names the count) before they give up with SWA_E_INTERNAL.  This returns how many steps
   of this context were repeated that way (0 on every run seen so far). */
int swa_d1_guard_retries(const swa_ctx * ctx);

/* Introspection used by the parity tests (bit-exact against the oracle): copies to host.
   what: 0 seqhash u64[n] · 1 Bloom bitmap u64[table_size/8] · 2 Zobrist table
   u64[4*(longest+2)] · 3 probe statistics u64[8] of the last network call
   {variants, bloom_pass, hash_match, verified, hits, 0,0,0} (0..3 build the database-wide structures on demand)
   · 10 / 11 member ids in group order of the streaming prefix / suffix index u32[n] · 12 / 13 their work-item
   buffers · 14 the counters u32[64] · 15 the amplicon lines (tools/check_index.py, tools/check_stream.py) */
int swa_d1_debug_read(swa_ctx * ctx, int what, void * out, size_t out_bytes);
uint64_t swa_d1_table_size(const swa_ctx * ctx);

/* The same network computed into the context's own HBM buffers and kept there (no host copy):
   *total = number of links.  swa_d1_network_fetch copies it out later if somebody wants the lists (-j);
   swa_d1_cluster_device evaluates the agglomeration of src/algod1.cc:1175-1257 on it where it lies:
   swarmid / generation / parent per amplicon (parent = SWA_NO_AMPLICON for seeds), `order` = the members
   swarm after swarm (swarms by seed id, members seed first, then by generation and id), swarm s =
   order[swarm_begin[s], swarm_begin[s + 1]).  swarm_begin holds swarm_cap + 1 entries; SWA_E_CAPACITY
   with *nswarms = the number of swarms when that is too small (n always suffices). */
int swa_d1_network_resident(swa_ctx * ctx, int no_cluster_breaking, uint64_t * total);
int swa_d1_network_fetch(swa_ctx * ctx, uint64_t * offsets, uint32_t * neighbours, uint64_t cap);
int swa_d1_cluster_device(swa_ctx * ctx, uint32_t * swarmid, uint32_t * generation, uint32_t * parent, uint32_t * order,
                          uint32_t * swarm_begin, uint32_t swarm_cap, uint32_t * nswarms);
/* swarmid / generation / parent may be null in swa_d1_cluster_device: they stay in HBM, and a caller that wants them
   after all (the reference's -i, -s, -u, -w outputs and --fastidious read ampinfo[].generation / parent,
   src/algod1.cc:994-1037) fetches them here (any of the three may be null again); valid until the next upload, index
   build, network or clustering call of the context.  swa_d1_cluster_maxgen: the deepest generation ("Max generations"). */
int swa_d1_cluster_fetch(swa_ctx * ctx, uint32_t * swarmid, uint32_t * generation, uint32_t * parent);
uint32_t swa_d1_cluster_maxgen(const swa_ctx * ctx);

/* ---- B2: fastidious second pass ------------------------------------------------- */
/* is_light[n]: != 0 when the amplicon's swarm has mass < boundary.  light_nt: total
   nt of amplicons in light swarms; bloom_bits: --bloom-bits (src/algod1.cc:1337-1357).
   graft_cand[n] <- min heavy amplicon id two microvariant steps away, or
   SWA_NO_AMPLICON.  counters[0..4] = {light variants, heavy variants, graft
   candidates, Bloom m (bits), Bloom k} = the reference's logged numbers
   (src/algod1.cc:1394-1396, 1436-1438, 1469-1470). */
int swa_d1_fastidious(swa_ctx * ctx, const uint8_t * is_light, uint64_t light_nt, uint32_t bloom_bits,
                      uint32_t * graft_cand, uint64_t * counters);
/* Multi-GPU form: the light side (Bloom + light table, src/algod1.cc:453-489) is built whole
   on this GPU, the heavy side (check_heavy_thread, src/algod1.cc:492-552) covers only slice
   `shard` of `nshards` of the heavy amplicons.  Combining the shards: graft_cand = element-wise
   minimum (the reference keeps the smallest heavy id, src/algod1.cc:244-258), counters[1] and
   counters[2] add up, counters[0], [3], [4] are identical on every shard. */
int swa_d1_fastidious_shard(swa_ctx * ctx, const uint8_t * is_light, uint64_t light_nt, uint32_t bloom_bits,
                            uint32_t shard, uint32_t nshards, uint32_t * graft_cand, uint64_t * counters);

/* ---- d = 0: dereplication -------------------------------
   Replaces the bucket search of dereplicating() (src/derep.cc:276-354: Zobrist hash, open
   addressing, exact sequence comparison on a hash match).  first_identical[i] = the smallest
   amplicon index whose sequence is identical to amplicon i's (== i for the first occurrence);
   caller allocates u32[n].  Independent of swa_d1_index_build, and invalidates a d = 1 index
   built on the same context. */
int swa_derep(swa_ctx * ctx, uint32_t * first_identical);

/* ---- B3: q-gram prefilter -------------------------------------------------------- */
/* 1024-bit 5-mer parity signature per amplicon (src/qgram.cc:68-96), kept in HBM */
int swa_qgram_build(swa_ctx * ctx);
/* difflist[i] = ceil(popcount(sig[seed] xor sig[amplist[i]]) / 10)  (src/qgram.cc:247-252) */
int swa_qgram_diff(swa_ctx * ctx, uint64_t seed, uint64_t listlen, const uint64_t * amplist,
                   uint64_t * difflist);
/* copy the signature matrix to host: out = u8[n][128] */
int swa_qgram_debug_read(swa_ctx * ctx, uint8_t * out, size_t out_bytes);

/* ---- B4: alignment scan ----------------------------------------------------------- */
/* penalties after the reference's gcd reduction (src/swarm.cc:466-483): defaults
   mismatch 18, gapopen 24, gapextend 13.  resolution d = -d value. */
int swa_search_begin(swa_ctx * ctx, uint64_t mismatch, uint64_t gapopen, uint64_t gapextend, uint64_t d);
/* 1 when the penalties / d given to swa_search_begin admit the wavefront kernel (every cost <= T
   decomposes uniquely, so the optimal cost alone fixes diff and alignment length), 0 when the
   banded tie-tracking kernel is used.  Same results either way; informational. */
int swa_search_uses_wavefront(const swa_ctx * ctx);
/* diffs[i] == the reference's search8/search16 + backtrack value whenever that value
   is <= d; otherwise some value > d (the caller only tests diff <= d, src/algo.cc:460,
   554).  scores / alignlengths may be NULL (never read by the reference's caller). */
int swa_search_do(swa_ctx * ctx, uint64_t query_no, uint64_t listlength, const uint64_t * targets,
                  uint64_t * scores, uint64_t * diffs, uint64_t * alignlengths);

/* ---- B3 + B4 fused: one (sub)seed of the d >= 2 greedy loop, pool state in HBM ------
   Replaces, per step of src/algo.cc:384-602, the host-side candidate list construction
   (abundance rule + `diffestimate <= radius + d` prune), qgram_diff_fast, the `qdiff <= d`
   filter, search_do and the `diff <= d` filter.  Requires swa_qgram_build and
   swa_search_begin.  swa_scan_begin: every amplicon unswarmed.
   swa_scan_step(seed, lowest_unswarmed, first_generation, radius, ncb, ...):
     candidates = unswarmed amplicons i >= lowest_unswarmed, i != seed, with
       (first_generation or est[i] <= radius + d) and (ncb or abundance[i] <= abundance[seed]);
     first_generation != 0 also marks `seed` as swarmed and stores est[i] = q-gram bound
       against this seed (src/algo.cc:442);
     hits = candidates with q-gram bound <= d and alignment diff <= d, returned in ascending
       id order (= the reference's pool order) with their diffs, and marked swarmed. */
int swa_scan_begin(swa_ctx * ctx);
int swa_scan_step(swa_ctx * ctx, uint32_t seed, uint32_t lowest_unswarmed, int first_generation, uint32_t radius,
                  int no_cluster_breaking, uint32_t * hit_ids, uint32_t * hit_diffs, uint32_t cap, uint32_t * nhits);
/* Batched form: all sub-seeds of one generation in one launch sequence.  seeds[k] with radius
   radii[k]; hits come back as triples (index k into seeds[], target id, diff) sorted by
   (k, id) and are computed against the pool as it was when the call started: a target may
   appear under several seeds, the caller keeps it for the first (queue order) and drops the
   others — which is exactly what processing the sub-seeds one by one gives.  Every returned
   target leaves the pool.  first_generation != 0 requires nseeds == 1. */
int swa_scan_batch(swa_ctx * ctx, uint32_t nseeds, const uint32_t * seeds, const uint32_t * radii,
                   uint32_t lowest_unswarmed, int first_generation, int no_cluster_breaking,
                   uint32_t * hit_seedidx, uint32_t * hit_ids, uint32_t * hit_diffs, uint32_t cap, uint32_t * nhits);
/* the (sorted) hits of the most recent swa_scan_batch again — for a caller whose buffers were
   too small (SWA_E_CAPACITY, *nhits = need): grow to *nhits and fetch; hit_seedidx may be NULL */
int swa_scan_fetch(swa_ctx * ctx, uint32_t * hit_seedidx, uint32_t * hit_ids, uint32_t * hit_diffs, uint32_t cap);
/* out3 = {q-gram comparisons, aligned pairs, launch sequences} since swa_scan_begin */
int swa_scan_totals(swa_ctx * ctx, uint64_t * out3);

/* ---- B3 + B4 in bulk: the whole d >= 2 search as ONE graph --------------------------------
   Everything qgram_diff_fast + search_do can ever answer during algo_run (src/algo.cc:423-602: per seed
   and sub-seed, the pool amplicons with q-gram bound <= d and alignment diff <= d) depends only on the
   pair, not on the pool, so it is computed for the whole database at once:
     row q of the CSR = every target t != q with diff(query q, target t) <= d that the abundance rule
     allows q to take (abundance[t] <= abundance[q], or all with no_cluster_breaking), ascending t,
     diffs[e] = that diff (the reference's value, src/algo.cc:460, 554).
   The caller's greedy loop then needs no further device call.  Candidate pairs come from d + 1 disjoint
   windows per sequence (an alignment with <= d differences leaves one of them intact, shifted by at most
   d), which every sequence must have room for: swa_dn_graph_supported != 0 (else use swa_scan_*).
   Requires swa_qgram_build and swa_search_begin.  Buffers as swa_d1_network: offsets[n + 1] always
   filled, *total = entries needed, SWA_E_CAPACITY when total > cap (call again with room: nothing is
   recomputed). */
int swa_dn_graph_supported(swa_ctx * ctx);
int swa_dn_graph(swa_ctx * ctx, int no_cluster_breaking, uint64_t * offsets, uint32_t * neighbours, uint8_t * diffs,
                 uint64_t cap, uint64_t * total);
/* The same graph left in HBM as the resident network: swa_d1_cluster_device then runs the greedy agglomeration of
   src/algo.cc:384-602 on it where it lies (seeds by lowest id, members by generation then id: src/algo.cc:205-219 — the
   same function of the directed graph as for d = 1), and swa_dn_parent_diffs returns, per amplicon, the differences to
   its parent in that clustering (0 for seeds; n bytes): radius(v) = radius(parent) + that (src/algo.cc:560-574). */
int swa_dn_graph_resident(swa_ctx * ctx, int no_cluster_breaking, uint64_t * total);
int swa_dn_parent_diffs(swa_ctx * ctx, uint8_t * pdiff);
/* out3 = {q-gram comparisons, aligned pairs, kernel launches} of the last swa_dn_graph */
int swa_dn_graph_totals(swa_ctx * ctx, uint64_t * out3);

/* ---- d = 1 on several GPUs of one node --------------------------------
   Replaces the thread fan-out of src/algod1.cc:1166-1167 / src/utils/threads.h:145-162: one context, stream and
   host thread per listed device inside the calling process; the database replicated, the probing divided by
   ownership of anchor groups (swa_d1_set_ownership): routed index build (the ranks' id lists all-to-all, grouped
   ncclSend / ncclRecv over xGMI), every rank's flat link list gathered on rank 0 (grouped ncclSend / ncclRecv: the one
   consumer of the network is the host behind rank 0 — an all-gather, which round 2 used and `bench.py --gpus N` still
   times as the north star names it, moves world x the bytes), CSR assembled there on the device; fastidious: heavy
   amplicons split, graft_cand combined with ncclAllReduce(min).  librccl.so is loaded when the first handle with ranks
   on distinct devices is created, not with the library.  A device may be listed more than once (several ranks on one GPU: the exchange then uses
   device-to-device copies — RCCL admits one rank per GPU); results never depend on the device list. */
typedef struct swa_multi swa_multi;
int  swa_multi_create(const int * devices, int ndevices, swa_multi ** out);
void swa_multi_destroy(swa_multi * m);
int  swa_multi_size(const swa_multi * m);
int  swa_multi_uses_rccl(const swa_multi * m);
swa_ctx * swa_multi_ctx(swa_multi * m, int rank);
const char * swa_multi_last_error(const swa_multi * m);
int  swa_multi_db_upload(swa_multi * m, const swa_db_view * host);
/* = swa_d1_index_build + swa_d1_network over the whole database (same buffers, same capacity protocol);
   *has_duplicates as swa_d1_index_build (SWA_E_DUPLICATES returned) */
int  swa_multi_d1_network(swa_multi * m, int no_cluster_breaking, uint64_t * offsets, uint32_t * neighbours, uint64_t cap,
                          uint64_t * total, int * has_duplicates);
/* = swa_d1_fastidious */
int  swa_multi_d1_fastidious(swa_multi * m, const uint8_t * is_light, uint64_t light_nt, uint32_t bloom_bits,
                             uint32_t * graft_cand, uint64_t * counters);


/* ---- d >= 2 on several GPUs (second paragraph) --------------------------------------------------
   Replaces the scan fan-out of src/scan.cc:221-256 under the loop of src/algo.cc:505-602.  swa_dn_set_ownership(rank,
   world): this context makes only the window groups of swa_dn_graph whose key maps to `rank`; a pair is reported
   through the first window it shares, and that window's group lives on one rank, so over all ranks every pair of the
   graph is found exactly once.  swa_multi_dn_begin = swa_qgram_build + swa_search_begin on every rank;
   swa_multi_dn_graph = swa_dn_graph: every rank finds and aligns its share, the accepted (query, target, diff)
   triples travel to rank 0 (RCCL send / receive over xGMI, device-to-device copies for ranks sharing a GPU), which
   sorts them into the CSR; same buffers and capacity protocol as swa_dn_graph. */
int  swa_dn_set_ownership(swa_ctx * ctx, uint32_t rank, uint32_t world);
int  swa_multi_dn_begin(swa_multi * m, uint64_t mismatch, uint64_t gapopen, uint64_t gapextend, uint64_t d);
int  swa_multi_dn_graph_supported(swa_multi * m);
int  swa_multi_dn_graph(swa_multi * m, int no_cluster_breaking, uint64_t * offsets, uint32_t * neighbours, uint8_t * diffs,
                        uint64_t cap, uint64_t * total);
int  swa_multi_dn_graph_totals(swa_multi * m, uint64_t * out3);

#ifdef __cplusplus
}
#endif
#endif /* SWARM_AMD_H */
