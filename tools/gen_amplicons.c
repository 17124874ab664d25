/*
 * tools/gen_amplicons.c — deterministic synthetic amplicon sets.
 *
 * This repo's own generator (nothing here comes from the reference).  Shapes:
 *   - n/50 random centroids of length L (uniform ACGT), heavy-tailed
 *     abundances  floor(20 * U^(-1/0.8)) + 2   (Pareto alpha = 0.8)
 *   - the rest: pick a random existing (non-light) amplicon, apply `e` random
 *     edits (60 % substitution, 20 % deletion, 20 % insertion; e = 1 for d=1
 *     sets, uniform 1..max_edits otherwise), abundance =
 *     max(1, floor(parent * U(0,0.3))), duplicates rejected
 *   - with light_frac > 0 that fraction of the amplicons are abundance-1
 *     "light" sequences at 2..3 edits from a non-light one and are never used
 *     as parents (exercises --fastidious)
 *   - output order shuffled; headers ">s<i>_<abundance>"
 *
 * Build:  gcc -O2 -o gen_amplicons tools/gen_amplicons.c          (CLI)
 *         gcc -O2 -shared -fPIC -DGEN_NO_MAIN ...                   (library)
 * CLI:    gen_amplicons <n> <L> <seed> <max_edits> <light_frac> <out.fasta>
 *         GEN_FLANK=<k> in the environment: all centroids share their first and last k nucleotides
 *         GEN_CORE=<k>:  all centroids are identical except for a k-nt core in the middle (conserved everywhere but
 *                        a hypervariable region)
 *         GEN_CONSERVED=<pct>: that share of the positions (in alternating conserved / variable stretches of 8..40 nt, both
 *                        ends conserved, the same layout and the same nucleotides in every centroid) is identical in all
 *                        centroids: what a 16S V4 read looks like to anything that groups sequences by windows
 *         GEN_ZIPF=<s>:  family sizes follow Zipf's law: the family of centroid r (r = 1, 2, ...) receives the share
 *                        s / r of all derived amplicons (until the shares add up to 1/2); the rest as usual.  s = 0.1
 *                        makes the largest family a tenth of the set: swarms of 10^5 members at n = 10^6
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { uint64_t s; } rng_t;
static uint32_t g_flank = 0;        /* GEN_FLANK environment variable (see below) */
static uint32_t g_core = 0;         /* GEN_CORE */
static double g_zipf = 0.0;         /* GEN_ZIPF */
static uint32_t g_conserved = 0;    /* GEN_CONSERVED (per cent) */
#define ZIPF_FAMILIES 64

static uint64_t rng_next(rng_t * r) {            /* splitmix64 */
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static double rng_unit(rng_t * r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static uint64_t rng_below(rng_t * r, uint64_t n) { return (uint64_t)(rng_unit(r) * (double)n); }

typedef struct {
  uint64_t off;     /* offset into the nucleotide pool */
  uint16_t len;
  uint8_t  light;
  uint64_t abundance;
  uint32_t family;  /* the centroid this amplicon descends from */
} amp_t;

static uint64_t seq_hash(const uint8_t * s, uint32_t len) {   /* FNV-1a over bases */
  uint64_t h = 0xcbf29ce484222325ULL ^ len;
  for (uint32_t i = 0; i < len; ++i) { h ^= s[i]; h *= 0x100000001b3ULL; }
  return h ? h : 1;
}

typedef struct { uint64_t * key; uint32_t * val; uint64_t mask; } set_t;

/* returns 1 if inserted, 0 if an identical sequence is already present */
static int set_insert(set_t * t, const uint8_t * pool, const amp_t * amps,
                      const uint8_t * s, uint32_t len, uint32_t id) {
  const uint64_t h = seq_hash(s, len);
  uint64_t i = h & t->mask;
  while (t->key[i]) {
    if (t->key[i] == h) {
      const amp_t * a = &amps[t->val[i]];
      if (a->len == len && memcmp(pool + a->off, s, len) == 0) return 0;
    }
    i = (i + 1) & t->mask;
  }
  t->key[i] = h; t->val[i] = id;
  return 1;
}

static uint32_t apply_edit(rng_t * r, uint8_t * s, uint32_t len) {
  const double u = rng_unit(r);
  if (u < 0.6 || len < 8) {                        /* substitution */
    const uint32_t p = (uint32_t)rng_below(r, len);
    s[p] = (uint8_t)((s[p] + 1 + rng_below(r, 3)) & 3);
    return len;
  }
  if (u < 0.8) {                                   /* deletion */
    const uint32_t p = (uint32_t)rng_below(r, len);
    memmove(s + p, s + p + 1, len - p - 1);
    return len - 1;
  }
  {                                                /* insertion */
    const uint32_t p = (uint32_t)rng_below(r, len + 1);
    memmove(s + p + 1, s + p, len - p);
    s[p] = (uint8_t)rng_below(r, 4);
    return len + 1;
  }
}

/* Generates the set and writes FASTA.  Returns 0 on success. */
static uint64_t g_id_offset = 0;   /* first header number (blocks generated separately get disjoint names) */

int gen_amplicons_fasta(uint64_t n, uint32_t L, uint64_t seed, uint32_t max_edits,
                        double light_frac, const char * path) {
  if (n == 0 || L < 8 || L > 60000 || max_edits == 0) return 1;
  rng_t rng = { seed * 0x2545F4914F6CDD1DULL + 0x1234567ULL };
  const uint32_t maxlen = L + 3 * max_edits + 8;
  amp_t * amps = (amp_t *)calloc(n, sizeof(amp_t));
  uint8_t * pool = (uint8_t *)malloc((size_t)n * maxlen);
  uint32_t * heavy_ids = (uint32_t *)malloc(n * sizeof(uint32_t));
  uint8_t * tmp = (uint8_t *)malloc(maxlen + 16);
  set_t set; uint64_t cap = 4; while (cap < 2 * n + 16) cap <<= 1;
  set.mask = cap - 1;
  set.key = (uint64_t *)calloc(cap, sizeof(uint64_t));
  set.val = (uint32_t *)calloc(cap, sizeof(uint32_t));
  if (!amps || !pool || !heavy_ids || !tmp || !set.key || !set.val) return 2;

  uint64_t centroids = n / 50; if (centroids < 1) centroids = 1;
  uint64_t n_light = (uint64_t)(light_frac * (double)n);
  if (n_light + centroids > n) n_light = n - centroids;
  const uint64_t n_heavy = n - n_light;
  uint64_t count = 0, heavy_count = 0;
  size_t pool_used = 0;
  /* GEN_ZIPF: member lists of the first ZIPF_FAMILIES families and the cumulative shares they are drawn with */
  uint32_t * fam_members[ZIPF_FAMILIES];
  uint64_t fam_count[ZIPF_FAMILIES], fam_cap[ZIPF_FAMILIES];
  double fam_cum[ZIPF_FAMILIES];
  uint32_t fam_used = 0;
  if (g_zipf > 0.0) {
    double total = 0.0;
    for (uint32_t r = 0; r < ZIPF_FAMILIES && r < centroids; ++r) {
      if (total + g_zipf / (double)(r + 1) > 0.5) break;
      total += g_zipf / (double)(r + 1);
      fam_cum[r] = total; fam_count[r] = 0; fam_cap[r] = 1024;
      fam_members[r] = (uint32_t *)malloc(fam_cap[r] * sizeof(uint32_t));
      if (!fam_members[r]) return 2;
      fam_used = r + 1;
    }
  }

  while (count < n) {
    uint32_t len;
    uint64_t abundance;
    uint8_t light = 0;
    uint32_t family_of_new = (uint32_t)count;
    if (count < centroids) {
      len = L;
      for (uint32_t i = 0; i < len; ++i) tmp[i] = (uint8_t)(rng_next(&rng) >> 62);
      /* GEN_FLANK=<k>: every centroid starts and ends with the same k nucleotides (conserved flanks / primers left
         on: the skewed case for anything that groups sequences by their ends); drawn from a generator of their own,
         so the rest of the set does not depend on the option */
      if (g_core > 0 && g_core < len) {            /* everything but the core from a generator of its own: the same for all */
        rng_t cr; cr.s = 0xC0DEC0DEULL;
        const uint32_t lo = (len - g_core) / 2, hi = lo + g_core;
        for (uint32_t i = 0; i < len; ++i) { const uint8_t b = (uint8_t)(rng_next(&cr) >> 62); if (i < lo || i >= hi) tmp[i] = b; }
      }
      if (g_conserved > 0 && g_conserved < 100) {
        /* stretches alternate, conserved first; a variable stretch follows a conserved one of c nt with
           c * (100 - pct) / pct nt, so the share holds along the sequence; the last 12 nt are conserved too */
        rng_t lay; lay.s = 0xC0115E27ULL;
        rng_t cons; cons.s = 0x0BA5E5ULL;
        uint32_t at = 0;
        while (at < len) {
          const uint32_t c = 8u + (uint32_t)rng_below(&lay, 33);
          for (uint32_t i = at; i < at + c && i < len; ++i) tmp[i] = (uint8_t)(rng_next(&cons) >> 62);
          at += c + (c * (100u - g_conserved) + g_conserved / 2u) / g_conserved;
        }
        for (uint32_t i = len > 12 ? len - 12 : 0; i < len; ++i) tmp[i] = (uint8_t)(rng_next(&cons) >> 62);
      }
      if (g_flank > 0 && 2u * g_flank < len) {
        rng_t fr; fr.s = 0x5EEDF1A2ULL;
        for (uint32_t i = 0; i < g_flank; ++i) tmp[i] = (uint8_t)(rng_next(&fr) >> 62);
        for (uint32_t i = 0; i < g_flank; ++i) tmp[len - g_flank + i] = (uint8_t)(rng_next(&fr) >> 62);
      }
      double u = rng_unit(&rng); if (u < 1e-12) u = 1e-12;
      double a = 20.0 * pow(u, -1.0 / 0.8) + 2.0;
      if (a > 1e12) a = 1e12;
      abundance = (uint64_t)a;
    } else {
      const amp_t * parent = &amps[heavy_ids[rng_below(&rng, heavy_count)]];
      if (fam_used != 0) {                         /* Zipf: a family by its share, then any of its members */
        const double u = rng_unit(&rng);
        for (uint32_t r = 0; r < fam_used; ++r) {
          if (u < fam_cum[r]) { if (fam_count[r] != 0) { parent = &amps[fam_members[r][rng_below(&rng, fam_count[r])]]; } break; }
        }
      }
      family_of_new = parent->family;
      len = parent->len;
      memcpy(tmp, pool + parent->off, len);
      uint32_t edits;
      if (count >= n_heavy) {                    /* light: 2..3 edits, abundance 1 */
        light = 1;
        edits = 2 + (uint32_t)rng_below(&rng, 2);
        abundance = 1;
      } else {
        edits = 1 + (max_edits > 1 ? (uint32_t)rng_below(&rng, max_edits) : 0);
        abundance = (uint64_t)((double)parent->abundance * (rng_unit(&rng) * 0.3));
        if (abundance < 1) abundance = 1;
      }
      for (uint32_t e = 0; e < edits; ++e) len = apply_edit(&rng, tmp, len);
    }
    if (!set_insert(&set, pool, amps, tmp, len, (uint32_t)count)) continue;   /* duplicate */
    amp_t * a = &amps[count];
    a->off = pool_used; a->len = (uint16_t)len; a->light = light; a->abundance = abundance; a->family = family_of_new;
    if (!light && family_of_new < fam_used) {
      const uint32_t r = family_of_new;
      if (fam_count[r] == fam_cap[r]) {
        fam_cap[r] *= 2;
        fam_members[r] = (uint32_t *)realloc(fam_members[r], fam_cap[r] * sizeof(uint32_t));
        if (!fam_members[r]) return 2;
      }
      fam_members[r][fam_count[r]++] = (uint32_t)count;
    }
    memcpy(pool + pool_used, tmp, len);
    pool_used += len;
    if (!light) heavy_ids[heavy_count++] = (uint32_t)count;
    ++count;
  }

  /* shuffled output order */
  uint32_t * order = (uint32_t *)malloc(n * sizeof(uint32_t));
  if (!order) return 2;
  for (uint64_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
  for (uint64_t i = n - 1; i > 0; --i) {
    const uint64_t j = rng_below(&rng, i + 1);
    const uint32_t t = order[i]; order[i] = order[j]; order[j] = t;
  }

  FILE * fp = fopen(path, "w");
  if (!fp) return 4;
  static const char sym[4] = { 'A', 'C', 'G', 'T' };
  char * line = (char *)malloc(maxlen + 64);
  setvbuf(fp, NULL, _IOFBF, 1 << 22);
  for (uint64_t k = 0; k < n; ++k) {
    const amp_t * a = &amps[order[k]];
    int w = sprintf(line, ">s%llu_%llu\n", (unsigned long long)(g_id_offset + order[k]), (unsigned long long)a->abundance);
    for (uint32_t i = 0; i < a->len; ++i) line[w + (int)i] = sym[pool[a->off + i]];
    line[w + a->len] = '\n';
    fwrite(line, 1, (size_t)w + a->len + 1, fp);
  }
  fclose(fp);
  for (uint32_t r = 0; r < fam_used; ++r) free(fam_members[r]);
  free(line); free(order); free(set.key); free(set.val); free(tmp); free(heavy_ids); free(pool); free(amps);
  return 0;
}

#ifndef GEN_NO_MAIN
int main(int argc, char ** argv) {
  if (argc != 7 && argc != 8) {
    fprintf(stderr, "usage: %s <n> <L> <seed> <max_edits> <light_frac> <out.fasta> [first header number]\n", argv[0]);
    return 2;
  }
  if (argc == 8) g_id_offset = strtoull(argv[7], NULL, 10);
  if (getenv("GEN_FLANK") != NULL) g_flank = (uint32_t)atoi(getenv("GEN_FLANK"));
  if (getenv("GEN_CORE") != NULL) g_core = (uint32_t)atoi(getenv("GEN_CORE"));
  if (getenv("GEN_ZIPF") != NULL) g_zipf = atof(getenv("GEN_ZIPF"));
  if (getenv("GEN_CONSERVED") != NULL) g_conserved = (uint32_t)atoi(getenv("GEN_CONSERVED"));
  const int rc = gen_amplicons_fasta(strtoull(argv[1], NULL, 10), (uint32_t)atoi(argv[2]),
                                     strtoull(argv[3], NULL, 10), (uint32_t)atoi(argv[4]),
                                     atof(argv[5]), argv[6]);
  if (rc) fprintf(stderr, "gen_amplicons: failed (%d)\n", rc);
  return rc;
}
#endif
