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
void swa_hostdb_free(swa_hostdb * db);
const char * swa_hostdb_error(const swa_hostdb * db);
/* host pointers into the handle (valid until swa_hostdb_free) */
void swa_hostdb_view(const swa_hostdb * db, swa_db_view * view);
uint64_t swa_hostdb_nucleotides(const swa_hostdb * db);
/* header of amplicon i (db order), NUL terminated; replaces db_getheader (src/db.h:47) */
const char * swa_hostdb_header(const swa_hostdb * db, uint32_t i, uint32_t * len);

#ifdef __cplusplus
}
#endif
#endif
