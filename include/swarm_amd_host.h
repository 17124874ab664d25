/*
 * swarm_amd_host.h — host-side C ABI of libswarm_amd.so: the pieces either side of the
 * GPU path that the reference keeps in plain host C++ (SURVEY.md §8f "next" rows):
 * FASTA ingest into the packed database, and the greedy single-linkage clustering +
 * writers that consume the GPU-returned neighbour lists.  No GPU is needed for any
 * function in this header.
 */
#ifndef SWARM_AMD_HOST_H
#define SWARM_AMD_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "swarm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct swa_hostdb swa_hostdb;

/* Replaces db_read (src/db.h:29-33, src/db.cc:432-803).  path "-" = stdin.
   usearch_abundance = -z; append_abundance = -a (0 = off); check_duplicate_sequences
   != 0 reproduces the d > 1 duplicate check of src/db.cc:763-790.
   On failure *out still receives a handle whose swa_hostdb_error() is the reference's
   fatal() text (the caller prints it and exits 1). */
int swa_hostdb_read_fasta(const char * path, int usearch_abundance, int64_t append_abundance,
                          int check_duplicate_sequences, swa_hostdb ** out);
/* The same with a notice as soon as the packed words are final (right after the parse, before the sort): on_words(user,
   pools, words per pool, pools) runs on the calling thread; the pools stay where they are for the life of the handle
   (they are the database's storage), so a GPU context coming up on another thread can start swa_db_stage_words on
   them while this call sorts.  on_words may be NULL. */
typedef void (*swa_words_ready_fn)(void * user, const uint64_t * const * piece_words, const uint64_t * piece_word_count, uint32_t pieces);
int swa_hostdb_read_fasta_staged(const char * path, int usearch_abundance, int64_t append_abundance, int check_duplicate_sequences,
                                 swa_words_ready_fn on_words, void * user, swa_hostdb ** out);
void swa_hostdb_free(swa_hostdb * db);
const char * swa_hostdb_error(const swa_hostdb * db);
/* host pointers into the handle (valid until swa_hostdb_free).  swa_hostdb_view: the packed sequences contiguous in db
   order — gathered on the host the first time it is asked for; swa_hostdb_unordered_view: the database as the reader
   keeps it (words in file order + where each amplicon's begin), what swa_db_upload_unordered takes — no host copy. */
void swa_hostdb_view(const swa_hostdb * db, swa_db_view * view);
void swa_hostdb_unordered_view(const swa_hostdb * db, swa_db_unordered_view * view);
uint64_t swa_hostdb_nucleotides(const swa_hostdb * db);
/* header of amplicon i (db order), NUL terminated; replaces db_getheader (src/db.h:47) */
const char * swa_hostdb_header(const swa_hostdb * db, uint32_t i, uint32_t * len);

/* ---- d = 1 host side: clustering over the neighbour CSR, grafting, writers --------- */
typedef struct swa_d1_result swa_d1_result;

/* Greedy breadth-first agglomeration over the CSR returned by swa_d1_network
   (replaces the clustering loop of algo_d1_run, src/algod1.cc:1185-1280). */
int  swa_d1_cluster(const swa_hostdb * db, const uint64_t * offsets, const uint32_t * neighbours,
                    swa_d1_result ** out);
/* The same result from the network that swa_d1_network_resident left in HBM: evaluated on the GPU
   (swa_d1_cluster_device), per-swarm sums on the host.  Needs the context that holds the network. */
int  swa_d1_cluster_resident(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result ** out);
/* The command line's form of the above: swarm / generation / parent and the per-swarm sums stay in HBM until an accessor,
   swa_d1_light_flags, swa_d1_graft or one of the -i -s -u -w writers asks for them (a plain `-o` run never does).  The
   caller keeps `ctx` alive, and its clustering untouched, until swa_d1_result_detach returned or the result is freed.
   swa_d1_result_detach fetches what is still on the device (SWA_E_DEVICE and swa_d1_result_error when that fails);
   after it the result does not refer to the context any more. */
int  swa_d1_cluster_resident_lazy(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result ** out);
/* The lazy form in two steps, for a caller that has a thread to spare while the network is being built: _prepare sizes the
   result's arrays for `db` and pins them (swa_host_pin) — any thread, any time after the context exists —, _prepared then
   clusters into them: the member order comes home as one DMA. */
int  swa_d1_result_prepare(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result ** out);
int  swa_d1_cluster_resident_prepared(swa_ctx * ctx, const swa_hostdb * db, swa_d1_result * prepared);
int  swa_d1_result_detach(swa_d1_result * res);
const char * swa_d1_result_error(const swa_d1_result * res);
void swa_d1_result_free(swa_d1_result * res);
/* out4 = {swarms after grafting, largest swarm, max generations, swarms before grafting}
   (the numbers of the log's summary lines, src/algod1.cc:1484-1487) */
void swa_d1_result_summary(const swa_d1_result * res, uint64_t * out4);
const uint32_t * swa_d1_result_swarmid(const swa_d1_result * res);     /* [n] */
const uint32_t * swa_d1_result_parent(const swa_d1_result * res);      /* [n], SWA_NO_AMPLICON for seeds */
const uint32_t * swa_d1_result_generation(const swa_d1_result * res);  /* [n] */
/* is_light[n] <- swarm mass < boundary; stats5 = {light swarms, amplicons in light swarms,
   nt in light swarms, heavy swarms, amplicons in heavy swarms} (src/algod1.cc:1291-1328) */
int  swa_d1_light_flags(const swa_d1_result * res, int64_t boundary, uint8_t * is_light, uint64_t * stats5);
/* attach_candidates (src/algod1.cc:274-336): graft_cand[n] as returned by swa_d1_fastidious;
   returns the number of grafts made */
uint32_t swa_d1_graft(swa_d1_result * res, const uint32_t * graft_cand);
/* writers (path "-" = stdout): -o/-r, -s, -i, -w, -j  (src/algod1.cc:755-1062) */
int swa_d1_write_swarms(const swa_d1_result * res, const swa_hostdb * db, const char * path, int mothur,
                        int usearch_abundance, int64_t append_abundance, int64_t differences);
int swa_d1_write_stats(const swa_d1_result * res, const swa_hostdb * db, const char * path, int usearch_abundance);
int swa_d1_write_structure(const swa_d1_result * res, const swa_hostdb * db, const char * path, int usearch_abundance);
int swa_d1_write_seeds(const swa_d1_result * res, const swa_hostdb * db, const char * path, int usearch_abundance);
int swa_d1_write_network(const swa_hostdb * db, const uint64_t * offsets, const uint32_t * neighbours,
                         const char * path, int usearch_abundance, int64_t append_abundance);

/* -u for d = 1 (src/algod1.cc:849-932); penalties after gcd reduction (18/24/13 by default) */
int swa_d1_write_uclust(const swa_d1_result * res, const swa_hostdb * db, const char * path, int usearch_abundance,
                        int64_t append_abundance, uint64_t mismatch, uint64_t gapopen, uint64_t gapextend);

/* ---- d >= 2 host side: the greedy loop of algo_run over the GPU's fused scan step ------- */
typedef struct swa_dn_result swa_dn_result;

/* Replaces algo_run's clustering loop (src/algo.cc:384-602).  Needs a context with the
   database resident (swa_db_upload of the same hostdb); runs swa_qgram_build,
   swa_search_begin, swa_scan_begin itself, then one swa_scan_step per seed / sub-seed.
   Penalties after the reference's gcd reduction (src/swarm.cc:466-483). */
int  swa_dn_cluster(swa_ctx * ctx, const swa_hostdb * db, int64_t differences, int no_cluster_breaking,
                    uint64_t mismatch, uint64_t gapopen, uint64_t gapextend, swa_dn_result ** out);
/* the same on the GPUs of a swa_multi (d >= 2 graph divided by ownership of window groups, swa_multi_dn_graph) */
int  swa_dn_cluster_multi(swa_multi * multi, const swa_hostdb * db, int64_t differences, int no_cluster_breaking,
                          uint64_t mismatch, uint64_t gapopen, uint64_t gapextend, swa_dn_result ** out);
void swa_dn_result_free(swa_dn_result * res);
const char * swa_dn_result_error(const swa_dn_result * res);
/* out3 = {number of swarms, largest swarm, max generations} (src/algo.cc:699-705) */
void swa_dn_result_summary(const swa_dn_result * res, uint64_t * out3);
/* writers: -o/-r, -s, -i, -w, -u  (src/algo.cc:122-325, 473-488, 573-589, 608-674) */
int swa_dn_write_swarms(const swa_dn_result * res, const swa_hostdb * db, const char * path, int mothur,
                        int usearch_abundance, int64_t append_abundance);
int swa_dn_write_stats(const swa_dn_result * res, const swa_hostdb * db, const char * path, int usearch_abundance);
int swa_dn_write_structure(const swa_dn_result * res, const swa_hostdb * db, const char * path, int usearch_abundance);
int swa_dn_write_seeds(const swa_dn_result * res, const swa_hostdb * db, const char * path, int usearch_abundance);
int swa_dn_write_uclust(const swa_dn_result * res, const swa_hostdb * db, const char * path, int usearch_abundance,
                        int64_t append_abundance);

/* ---- d = 0: dereplication (src/derep.cc) ------------------------------------------------
   Clusters of identical sequences from swa_derep's array: members in db order, the first one
   is the seed; clusters by decreasing mass, then by seed index (src/derep.cc:67-90). */
typedef struct swa_d0_result swa_d0_result;
int  swa_d0_cluster(const swa_hostdb * db, const uint32_t * first_identical, swa_d0_result ** out);
void swa_d0_result_free(swa_d0_result * res);
/* out3 = {clusters, largest cluster (members), heaviest cluster (mass)}  (src/derep.cc:405-410) */
void swa_d0_result_summary(const swa_d0_result * res, uint64_t * out3);
/* writers: src/derep.cc:207-273 (-o / -r), 188-204 (-w), 107-124 (-s), 127-148 (-i), 151-185 (-u) */
int swa_d0_write_swarms(const swa_d0_result * res, const swa_hostdb * db, const char * path, int mothur,
                        int usearch_abundance, int64_t append_abundance, int64_t differences);
int swa_d0_write_seeds(const swa_d0_result * res, const swa_hostdb * db, const char * path, int usearch_abundance);
int swa_d0_write_stats(const swa_d0_result * res, const swa_hostdb * db, const char * path, int usearch_abundance);
int swa_d0_write_structure(const swa_d0_result * res, const swa_hostdb * db, const char * path, int usearch_abundance);
int swa_d0_write_uclust(const swa_d0_result * res, const swa_hostdb * db, const char * path, int usearch_abundance,
                        int64_t append_abundance);

/* ---- the command line (src/swarm.cc:96-124, 269-463, 486-630) ---------------------------
   `swarm` itself: options, log lines, FASTA in, output files out, exit status; everything above driven the way the
   reference's main() drives its own functions.  The executable swarm_amd/bin/swarm is a launcher that sets the OpenMP
   wait policy and calls this (host/launcher.cpp). */
int swa_cli_main(int argc, char ** argv);

#ifdef __cplusplus
}
#endif
#endif
