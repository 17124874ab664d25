// d1.hip — seam B1: the d = 1 amplicon network on gfx950.
//
// Replaces (behaviourally) the reference's per-thread loop
//   network_thread -> check_variants -> generate_variants + find_variant_matches
// (src/algod1.cc:558-670, src/variants.cc:184-249) and the serial table build
// (src/algod1.cc:188-208, 1122-1150).  Nothing here is a translation of that code:
//
//   * one 64-lane wavefront owns one query amplicon; lane l owns the K = ceil((L+1)/64)
//     consecutive sequence positions [l*K, (l+1)*K).  The reference's serially
//     incremental deletion / insertion hashes become three wave-wide XOR scans
//     (exclusive prefix of Z[p][s_p], suffixes of Z[p-1][s_p] and Z[p+1][s_p]);
//   * default route (d1_anchor.inc): amplicons grouped by their first / last 32 nucleotides,
//     a group's members in an LDS table + LDS Bloom, every probe answered from LDS;
//   * plain route (k_d1_probe, this file: statistics, sequences the anchored passes cannot
//     serve, the fastidious second level): the 2-bit packed query, the Zobrist table and the
//     1024 Bloom patterns in LDS, Bloom-word loads issued 8 at a time per lane, survivors
//     compacted with __ballot/__popcll into a per-wave LDS queue and drained 64 at a time so
//     that table walk, abundance rule and exact verification run with full lanes;
//   * the hash table is one 16-byte slot per entry (hash, amplicon id): one transaction
//     per probe step instead of the reference's three arrays;
//   * hits leave a wave through its own segment of the edge buffer (one writer per segment,
//     no atomics); a scan + scatter + per-row sort turns the segments into the canonical CSR.
//
// All arithmetic is 64-bit integer XOR/shift/compare; results are bit-exact.
#include "swa_internal.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <memory>
#include <thread>

namespace {

constexpr int kWaves = 4;                 // waves per workgroup
constexpr int kThreads = kWaves * 64;
constexpr int kQueueCap = 128;            // per-wave candidate queue (>= 64 + 63)
constexpr uint32_t kEmpty = SWA_NO_AMPLICON;
constexpr size_t kMaxZobristLds = 96 * 1024;

struct alignas(16) swa_task {   // one surviving first-level microvariant of a heavy amplicon
  uint64_t hash;
  uint32_t heavy;
  uint32_t code;
};

struct NetArgs {
  const uint64_t * seqs;
  const uint64_t * seq_off;
  const uint32_t * seqlen;
  const uint64_t * abundance;
  const uint64_t * zobrist;     // 4 * zlen
  uint32_t zlen;
  uint32_t maxwords;            // ceil(longest / 32)
  const swa_slot * table;
  uint64_t tmask;
  const uint64_t * bloom;
  uint64_t bmask;
  const uint64_t * patterns;    // 1024
  int no_cluster_breaking;
  uint32_t first;
  uint32_t count;
  uint64_t * edges;             // (src << 32) | dst
  uint64_t seg_cap;             // edges are appended to per-wave segments: segment w = edges[w*seg_cap ..)
  uint32_t * seg_fill;          // [launch waves] fill of each segment (no atomics: one writer per segment)
  uint32_t * counts;            // per query amplicon (index k = amp - first)
  unsigned long long * stats;   // [0] variants [1] bloom pass [2] hash match [3] verified
  // MODE 2 (seeds the anchored passes cannot serve): explicit (seed, position range) list
  const struct swa_fallback * fallback;
  const uint32_t * fallback_count;
  const struct swa_aux * aux;
  // MODE 0 in a multi-GPU job (swa_d1_set_ownership, world > 1): only seeds with id mod world == rank
  uint32_t owner_rank, owner_world;
  // MODE 1 (fastidious second level)
  const swa_task * tasks;
  uint32_t * graft;
  unsigned long long * cand_counter;
  uint32_t fast_min_len;        // a (heavy, light) pair with both lengths >= this belongs to the pair route (d1_fast.inc)
  // MODE 2 in window mode (anchor windows moved inwards): a fallback seed's share is defined by the window itself —
  // range 1 = neighbours with the same prefix-side window (word win_word), range 2 = the others; all positions are
  // enumerated and the hits filtered
  uint32_t window_mode, win_word;
  uint32_t anchor_w;            // width of the anchor windows in nt (32 / 64 / 128): the window is words [win_word, win_word + anchor_w / 32)
};

#define SWA_ANCHOR_TYPES_ONLY
#include "d1_anchor.inc"
#undef SWA_ANCHOR_TYPES_ONLY

// ---- sequence hashes (db.cc:761 zobrist_hash) --------------------------------------
template <bool ZLDS>
__global__ __launch_bounds__(256) void k_seqhash(const uint64_t * __restrict__ seqs,
                                                 const uint64_t * __restrict__ seq_off,
                                                 const uint32_t * __restrict__ seqlen,
                                                 const uint64_t * __restrict__ zobrist, uint32_t zlen,
                                                 uint32_t n, uint64_t * __restrict__ seqhash,
                                                 swa_aux * __restrict__ aux, uint32_t anchor_w, const uint8_t * __restrict__ only) {
  extern __shared__ uint64_t lds[];
  const uint64_t * zob = zobrist;
  if (ZLDS) {
    for (uint32_t i = threadIdx.x; i < 4u * zlen; i += blockDim.x) { lds[i] = zobrist[i]; }
    __syncthreads();
    zob = lds;
  }
  for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) {
    if (only != nullptr && only[a] == 0) { continue; }         // (the member table of the groups left to the plain kernel: only those)
    const uint64_t * s = seqs + seq_off[a];
    const uint32_t len = seqlen[a];
    uint64_t h = 0, dall = 0, iall = 0;
    uint64_t word = 0;
    swa_aux ax{};
    // pb: first position of the run that contains position anchor_w - 1 (see swa_aux; 31 for the 32-nt windows of rounds
    // 1-3), and the three streams below it: kept as of the start of the current run until that position is reached
    uint32_t run_start = 0, prev = 4u;
    uint64_t rh = 0, rd = 0, ri = 0;
    ax.pb = anchor_w;
    for (uint32_t p = 0; p < len; ++p) {
      if ((p & 31u) == 0u) { word = s[p >> 5]; }
      const uint32_t c = (uint32_t)(word & 3u);
      if (c != prev) { run_start = p; rh = h; rd = dall; ri = iall; prev = c; }
      if (p + 1u == anchor_w) { ax.pb = run_start; ax.a32 = rh; ax.d32 = rd; ax.i32 = ri; }
      h ^= zob[4u * p + c];
      if (p >= 1u) { dall ^= zob[4u * (p - 1u) + c]; }
      iall ^= zob[4u * (p + 1u) + c];
      word >>= 2;
    }
    if (len < anchor_w) { ax.a32 = h; ax.d32 = dall; ax.i32 = iall; }
    ax.h = h; ax.dall = dall; ax.iall = iall;
    seqhash[a] = h;
    aux[a] = ax;
  }
}

// ---- table + Bloom build (algod1.cc:188-208; insertion order is free) ---------------
__global__ __launch_bounds__(256) void k_table_clear(swa_slot * table, uint64_t slots, uint64_t * bloom,
                                                     uint64_t bloom_words) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += stride) {
    swa_slot s;
    s.hash = 0; s.amp = kEmpty; s.pad = 0;
    table[i] = s;
  }
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < bloom_words; i += stride) {
    bloom[i] = ~0ull;          // inverted polarity, bloompat.cc:93-97
  }
}

// inserts amplicons for which (is_member == nullptr || is_member[a] != 0)
__global__ __launch_bounds__(256) void k_table_insert(const uint64_t * __restrict__ seqhash, uint32_t n,
                                                      const uint8_t * __restrict__ is_member,
                                                      swa_slot * table, uint64_t tmask,
                                                      unsigned long long * bloom, uint64_t bmask,
                                                      const uint64_t * __restrict__ patterns) {
  for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) {
    if (is_member != nullptr && is_member[a] == 0) { continue; }
    const uint64_t h = seqhash[a];
    uint64_t idx = (h >> 32) & tmask;                          // hashtable.cc:47-53
    for (;;) {
      const uint32_t old = atomicCAS(&table[idx].amp, kEmpty, a);
      if (old == kEmpty) { break; }
      idx = (idx + 1) & tmask;
    }
    table[idx].hash = h;
    atomicAnd(&bloom[(h >> 10) & bmask], ~(unsigned long long)patterns[h & 1023u]);   // bloompat.cc:62-65
  }
}

// identical sequences => flag (algod1.cc:174-208 detects this while inserting)
__global__ __launch_bounds__(256) void k_dup_check(const uint64_t * __restrict__ seqs,
                                                   const uint64_t * __restrict__ seq_off,
                                                   const uint32_t * __restrict__ seqlen,
                                                   const uint64_t * __restrict__ seqhash, uint32_t first,
                                                   uint32_t count, const swa_slot * __restrict__ table, uint64_t tmask,
                                                   uint32_t * flag) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
    const uint32_t a = first + k;
    const uint64_t h = seqhash[a];
    const uint32_t len = seqlen[a];
    uint64_t idx = (h >> 32) & tmask;
    for (;;) {
      const swa_slot s = table[idx];
      if (s.amp == kEmpty) { break; }
      if (s.hash == h && s.amp != a && seqlen[s.amp] == len) {
        const uint64_t * x = seqs + seq_off[a];
        const uint64_t * y = seqs + seq_off[s.amp];
        bool same = true;
        for (uint32_t w = 0; w < ((len + 31u) >> 5); ++w) { same = same && (x[w] == y[w]); }
        if (same) { atomicOr(flag, 1u); }
      }
      idx = (idx + 1) & tmask;
    }
  }
}

// ---- the probe kernels ------------------------------------------------------------------

// Wave-cooperative enumeration of all microvariants of the staged sequence `sw` (len nt).
// Lane l owns positions [l*K, (l+1)*K), K = ceil((len+1)/64); `f(slots)` is called once
// per owned position (K times, wave-uniformly) with that position's 8 candidate slots:
//   [0..3] insertion of base b before p        (variants.cc:226-246)
//   [4..6] substitution of p by the 3 others   (variants.cc:192-206)
//   [7]    deletion of p, once per run          (variants.cc:210-222)
// code = type | base << 2 | pos << 4.  Returns the sequence's own hash.
struct Slots {
  uint64_t hs[8];
  bool ok[8];
  uint32_t code[8];
};

template <class F>
__device__ __forceinline__ uint64_t enumerate_variants(const uint64_t * sw, uint32_t len, const uint64_t * zob,
                                                       int lane, F && f) {
  // pass 1: per-lane XOR of the three Zobrist streams over the owned positions
  const uint32_t K = (len + 64u) >> 6;                     // ceil((len + 1) / 64)
  const uint32_t p0 = (uint32_t)lane * K;
  uint64_t xa = 0, xd = 0, xi = 0;
  for (uint32_t j = 0; j < K; ++j) {
    const uint32_t p = p0 + j;
    if (p < len) {
      const uint32_t c = swa_nt(sw, p);
      xa ^= zob[4u * p + c];
      if (p >= 1u) { xd ^= zob[4u * (p - 1u) + c]; }
      xi ^= zob[4u * (p + 1u) + c];
    }
  }
  // wave scans: exclusive prefix of xa; inclusive suffixes of xd, xi
  uint64_t pa = xa, sd = xd, si = xi;
#pragma unroll
  for (unsigned d = 1; d < 64; d <<= 1) {
    const uint64_t ta = swa_shfl_up_u64(pa, d);
    const uint64_t td = swa_shfl_down_u64(sd, d);
    const uint64_t ti = swa_shfl_down_u64(si, d);
    if (lane >= (int)d) { pa ^= ta; }
    if (lane + (int)d < 64) { sd ^= td; si ^= ti; }
  }
  const uint64_t H = swa_shfl_u64(pa, 63);                 // hash of the whole sequence
  pa ^= xa;                                                // exclusive

  // pass 2: the variants of the owned positions
  uint32_t prevc = (p0 >= 1u && p0 - 1u < len) ? swa_nt(sw, p0 - 1u) : 4u;
  for (uint32_t j = 0; j < K; ++j) {
    const uint32_t p = p0 + j;
    const bool in_seq = p < len;
    const bool in_ins = p <= len;
    const uint32_t c = in_seq ? swa_nt(sw, p) : 4u;
    uint64_t z[4];
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) { z[b] = in_ins ? zob[4u * p + b] : 0ull; }
    // z[c] without dynamic register indexing
    const uint64_t zc = (c == 0u) ? z[0] : (c == 1u) ? z[1] : (c == 2u) ? z[2] : z[3];
    const uint64_t za = in_seq ? zc : 0ull;
    const uint64_t zd = (in_seq && p >= 1u) ? zob[4u * (p - 1u) + c] : 0ull;
    const uint64_t zi = in_seq ? zob[4u * (p + 1u) + c] : 0ull;

    Slots s;
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) {
      s.hs[b] = pa ^ z[b] ^ si;
      s.ok[b] = in_ins && (p == 0u || b != prevc);
      s.code[b] = 2u | (b << 2) | (p << 4);
    }
#pragma unroll
    for (uint32_t t = 0; t < 3u; ++t) {
      const uint32_t b = (t < c) ? t : t + 1u;               // the t-th base that is not c
      s.hs[4 + t] = H ^ za ^ ((t < c) ? z[t] : z[t + 1u]);
      s.ok[4 + t] = in_seq;
      s.code[4 + t] = 0u | (b << 2) | (p << 4);
    }
    s.hs[7] = pa ^ sd ^ zd;
    s.ok[7] = in_seq && (p == 0u || c != prevc);
    s.code[7] = 1u | (p << 4);

    f(s);

    // advance the running prefix / suffixes past position p
    pa ^= za;
    sd ^= zd;
    si ^= zi;
    prevc = c;
  }
  return H;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one probe of the candidate (hash h, edit code) drawn from the queue: walk the cluster,
// apply the abundance rule, verify exactly (algod1.cc:558-603, variants.cc:118-165).
// SECOND = the fastidious second level (hash_check_attach, algod1.cc:339-371): no self /
// abundance test there.
template <bool SECOND>
__device__ __forceinline__ bool probe_and_verify(const NetArgs & a, const uint64_t * sw, uint32_t slen,
                                                 uint32_t snw, uint32_t seed, uint64_t seed_abundance,
                                                 uint64_t h, uint32_t code, uint32_t & out_amp,
                                                 uint32_t & n_match, int window_filter = 0, uint32_t win_word = 0, uint32_t nwin = 1) {
  const uint32_t type = code & 3u;
  const uint32_t base = (code >> 2) & 3u;
  const uint32_t pos = code >> 4;
  const uint32_t vlen = (type == 0u) ? slen : (type == 1u ? slen - 1u : slen + 1u);
  const uint32_t vnw = (vlen + 31u) >> 5;
  uint64_t idx = (h >> 32) & a.tmask;
  for (;;) {
    const swa_slot s = a.table[idx];
    if (s.amp == kEmpty) { return false; }
    if (s.hash == h) {
      ++n_match;
      const uint32_t amp = s.amp;
      const uint32_t alen = a.seqlen[amp];
      bool allowed;
      if (SECOND) { allowed = min(a.seqlen[seed], alen) < a.fast_min_len; }
      else { allowed = amp != seed && (a.no_cluster_breaking != 0 || seed_abundance >= a.abundance[amp]); }
      if (allowed && alen == vlen) {
        const uint64_t * y = a.seqs + a.seq_off[amp];
        // window_filter 1: only neighbours that share the seed's prefix-side window, 2: only those that do not
        bool shared = true;                                    // the seed's prefix-side window, in the neighbour
        for (uint32_t w = 0; w < nwin; ++w) { shared = shared && y[win_word + w] == sw[win_word + w]; }
        bool same = window_filter == 0 || (window_filter == 1 ? shared : !shared);
        for (uint32_t w = 0; w < vnw; ++w) {
          same = same && (swa_variant_word(sw, snw, type, pos, base, w) == y[w]);
        }
        if (same) {
          out_amp = amp;
          return true;
        }
      }
    }
    idx = (idx + 1) & a.tmask;
  }
}

// MODE 0: the d=1 network (query = amplicon first+k of the database).
// MODE 1: fastidious second level (query k = the microvariant tasks[k] of a heavy amplicon;
//         every verified light amplicon gets graft_cand = min(heavy id), algod1.cc:244-258).
template <bool ZLDS, bool STATS, int MODE>
__global__ __launch_bounds__(kThreads) void k_d1_probe(const NetArgs a) {
  extern __shared__ uint64_t lds[];
  // LDS carve-up (all 8-byte aligned)
  uint64_t * zob_lds = lds;
  uint64_t * pat = lds + (ZLDS ? 4u * a.zlen : 0u);
  uint64_t * wave_base = pat + 1024;
  const uint32_t seed_words = a.maxwords + 2u;               // + zero padding words
  const uint32_t per_wave = seed_words + kQueueCap + kQueueCap / 2;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t * sw = wave_base + (size_t)wave * per_wave;       // staged query words
  uint64_t * qh = sw + seed_words;                           // queue: hashes
  uint32_t * qc = reinterpret_cast<uint32_t *>(qh + kQueueCap);   // queue: edit codes

  if (ZLDS) {
    for (uint32_t i = threadIdx.x; i < 4u * a.zlen; i += kThreads) { zob_lds[i] = a.zobrist[i]; }
  }
  for (uint32_t i = threadIdx.x; i < 1024u; i += kThreads) { pat[i] = a.patterns[i]; }
  __syncthreads();
  const uint64_t * zob = ZLDS ? zob_lds : a.zobrist;

  unsigned long long st_var = 0, st_pass = 0, st_match = 0, st_ver = 0;
  unsigned long long cand_total = 0;                         // MODE 1: graft candidates found by this wave
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  // this wave's private output segment (a single shared edge counter costs one contended
  // returning atomic per query — measured: ~88 atomics/us on one address bound the kernel)
  const uint32_t gwave = blockIdx.x * kWaves + wave;
  uint32_t seg_at = (MODE == 1) ? 0u : a.seg_fill[gwave];

  const uint32_t nwaves = gridDim.x * kWaves;
  const uint32_t nqueries = (MODE == 2) ? *a.fallback_count : a.count;
  for (uint32_t k = blockIdx.x * kWaves + wave; k < nqueries; k += nwaves) {
    uint32_t seed, len, nw;
    uint32_t range = 0;                                      // MODE 2: 0 all positions, 1 p >= 32, 2 p < 32
    uint64_t seed_ab = 0;
    if (MODE == 0 || MODE == 2) {
      if (MODE == 2) { seed = a.fallback[k].seed; range = a.fallback[k].range; }
      else {
        seed = a.first + k;
        if (a.owner_world > 1u && seed % a.owner_world != a.owner_rank) { continue; }   // another rank's seed (wave-uniform)
      }
      len = a.seqlen[seed];
      nw = (len + 31u) >> 5;
      const uint64_t * gs = a.seqs + a.seq_off[seed];
      seed_ab = a.abundance[seed];
      for (uint32_t w = lane; w < nw + 2u; w += 64u) { sw[w] = (w < nw) ? gs[w] : 0ull; }
    } else {
      // materialise the first-level microvariant (generate_variant_sequence, variants.cc:78-115)
      const swa_task t = a.tasks[k];
      seed = t.heavy;
      const uint32_t hlen = a.seqlen[seed];
      const uint32_t hnw = (hlen + 31u) >> 5;
      const uint64_t * gs = a.seqs + a.seq_off[seed];
      const uint32_t type = t.code & 3u;
      len = (type == 0u) ? hlen : (type == 1u ? hlen - 1u : hlen + 1u);
      nw = (len + 31u) >> 5;
      for (uint32_t w = lane; w < nw + 2u; w += 64u) {
        sw[w] = (w < nw) ? swa_variant_word(gs, hnw, type, t.code >> 4, (t.code >> 2) & 3u, w) : 0ull;
      }
    }
    wave_lds_sync();

    uint32_t qn = 0;          // queue fill (wave uniform)
    uint32_t row = 0;         // hits of this query (wave uniform)

    auto drain = [&](uint32_t cnt) {
      bool hit = false;
      uint32_t amp = 0;
      uint32_t nmatch = 0;
      if ((uint32_t)lane < cnt) {
        // MODE 2: a seed's two halves are divided by the prefix-side window itself, as the pair kernels divide them — range 1
        // keeps only neighbours that share it, range 2 only those that do not.  (By position alone the halves overlap: a
        // deletion at position 31, or inside a run that ends there, is listed at a position >= pb AND changes the window.)
        const int wf = MODE != 2 ? 0 : (int)range;
        hit = probe_and_verify<MODE == 1>(a, sw, len, nw, seed, seed_ab, qh[lane], qc[lane], amp, nmatch, wf, a.win_word, MODE == 2 ? a.anchor_w / 32u : 1u);
      }
      const uint64_t hm = __ballot(hit);
      if (STATS) {
        for (int o = 32; o > 0; o >>= 1) { nmatch += __shfl_down(nmatch, o, 64); }
        st_match += __shfl(nmatch, 0, 64);
      }
      if (hm != 0ull) {
        const uint32_t nh = (uint32_t)__popcll(hm);
        if (MODE == 0 || MODE == 2) {
          if (hit) {
            const uint64_t at = (uint64_t)seg_at + (uint64_t)__popcll(hm & lane_lt);
            if (at < a.seg_cap) { a.edges[(uint64_t)gwave * a.seg_cap + at] = ((uint64_t)seed << 32) | amp; }
          }
          seg_at += nh;
        } else {
          if (hit) { atomicMin(&a.graft[amp], seed); }
        }
        row += nh;
        if (STATS) { st_ver += nh; }
      }
    };

    auto enqueue = [&](bool pass, uint64_t h, uint32_t code) {
      const uint64_t m = __ballot(pass);
      if (m == 0ull) { return; }
      if (pass) {
        const uint32_t at = qn + (uint32_t)__popcll(m & lane_lt);
        qh[at] = h;
        qc[at] = code;
      }
      qn += (uint32_t)__popcll(m);
      if (STATS) { st_pass += (unsigned long long)__popcll(m); }
      if (qn >= 64u) {
        wave_lds_sync();
        drain(64u);
        const uint32_t rest = qn - 64u;
        uint64_t th = 0;
        uint32_t tc = 0;
        if ((uint32_t)lane < rest) { th = qh[64 + lane]; tc = qc[64 + lane]; }
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < rest) { qh[lane] = th; qc[lane] = tc; }
        qn = rest;
      }
    };

    // 8 Bloom-word loads in flight per lane, then test + compact
    auto probe_slots = [&](const auto & s) {
      uint64_t word[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        word[i] = s.ok[i] ? a.bloom[(s.hs[i] >> 10) & a.bmask] : ~0ull;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool pass = s.ok[i] && ((word[i] & pat[s.hs[i] & 1023u]) == 0ull);   // bloompat.cc:68-71
        if (STATS) { st_var += (unsigned long long)__popcll(__ballot(s.ok[i])); }
        enqueue(pass, s.hs[i], s.code[i]);
      }
    };
    if (MODE == 2) {
      const swa_aux ax = a.aux[seed];
      const bool by_position = a.window_mode == 0u;           // (window mode: every position, the hits are filtered)
      const uint32_t pb = by_position && range == 1u ? ax.pb : 0u;   // range 1 = what the prefix pass would have done (swa_aux::pb)
      const uint32_t pe = by_position && range == 2u ? a.anchor_w : len + 1u;
      const uint32_t erange = by_position ? range : 0u;       // (`range` itself still selects the filter in drain())
      enumerate_range(sw, len, zob, lane, pb, pe < len + 1u ? pe : len + 1u, ax.h, erange == 1u ? ax.a32 : 0ull,
                      erange == 1u ? (ax.dall ^ ax.d32) : ax.dall, erange == 1u ? (ax.iall ^ ax.i32) : ax.iall,
                      probe_slots);
    } else {
      (void)enumerate_variants(sw, len, zob, lane, probe_slots);
    }
    if (qn > 0u) {
      wave_lds_sync();
      drain(qn);
    }
    if (MODE == 0) {
      if (lane == 0) { a.counts[k] = row; }
    } else if (MODE == 2) {
      if (lane == 0 && row != 0u && a.counts != nullptr) { atomicAdd(&a.counts[seed - a.first], row); }
    } else {
      cand_total += row;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (MODE != 1 && lane == 0) { a.seg_fill[gwave] = seg_at; }
  if (MODE == 1 && lane == 0 && cand_total != 0ull) { atomicAdd(a.cand_counter, cand_total); }
  if (STATS && lane == 0) {
    atomicAdd(&a.stats[0], st_var);
    atomicAdd(&a.stats[1], st_pass);
    atomicAdd(&a.stats[2], st_match);
    atomicAdd(&a.stats[3], st_ver);
  }
}

#include "d1_anchor.inc"

// ---- fastidious first level (algod1.cc:495-518 mark_light_var, 398-450 check_heavy_var) ---
struct FlexArgs {
  const uint64_t * seqs;
  const uint64_t * seq_off;
  const uint32_t * seqlen;
  const uint64_t * zobrist;
  uint32_t zlen;
  uint32_t maxwords;
  unsigned long long * fbits;     // the flexible Bloom, inverted polarity (bloomflex.cc:61-70)
  uint64_t fsize;                 // words
  double finv;                    // 1.0 / fsize
  const uint64_t * fpat;          // 65536 patterns with k bits
  const uint32_t * list;          // amplicon ids to process
  uint32_t count;
  swa_task * tasks;               // PASS 1: surviving (heavy, variant) pairs
  unsigned long long * task_counter;
  uint64_t task_cap;
  unsigned long long * variant_counter;
};

// (h >> 16) % fsize for an arbitrary (non power-of-two) fsize: bloomflex.cc:43-49.
// x < 2^48 is exact in a double, so the quotient estimate is off by at most one.
__device__ __forceinline__ uint64_t flex_word(uint64_t h, uint64_t fsize, double finv) {
  const uint64_t x = h >> 16;
  uint64_t q = (uint64_t)((double)x * finv);
  int64_t r = (int64_t)(x - q * fsize);
  if (r < 0) { r += (int64_t)fsize; }
  else if ((uint64_t)r >= fsize) { r -= (int64_t)fsize; }
  return (uint64_t)r;
}

// PASS 0: clear the pattern bits of every microvariant of a light amplicon.
// PASS 1: test every microvariant of a heavy amplicon, emit the survivors as tasks.
template <bool ZLDS, int PASS>
__global__ __launch_bounds__(kThreads) void k_d1_flex(const FlexArgs a) {
  extern __shared__ uint64_t lds[];
  uint64_t * zob_lds = lds;
  uint64_t * wave_base = lds + (ZLDS ? 4u * a.zlen : 0u);
  const uint32_t seed_words = a.maxwords + 2u;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t * sw = wave_base + (size_t)wave * seed_words;
  // PASS 1 stages its surviving tasks in LDS and reserves space in the global task list in
  // bulk: one returning atomic per ~200 tasks instead of one per ballot (a single address
  // sustains only ~88 atomics/us, which bound this kernel)
  constexpr uint32_t kStage = 256;
  swa_task * stage = reinterpret_cast<swa_task *>(wave_base + (size_t)kWaves * seed_words) + (size_t)wave * kStage;
  uint32_t nstage = 0;
  if (ZLDS) {
    for (uint32_t i = threadIdx.x; i < 4u * a.zlen; i += kThreads) { zob_lds[i] = a.zobrist[i]; }
    __syncthreads();
  }
  const uint64_t * zob = ZLDS ? zob_lds : a.zobrist;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  unsigned long long nvar = 0;
  auto flush = [&]() {
    wave_lds_sync();
    unsigned long long base = 0;
    if (lane == 0) { base = atomicAdd(a.task_counter, (unsigned long long)nstage); }
    base = swa_shfl_u64(base, 0);
    for (uint32_t i = lane; i < nstage; i += 64u) {
      if (base + i < a.task_cap) { a.tasks[base + i] = stage[i]; }
    }
    nstage = 0;
    wave_lds_sync();
  };
  const uint32_t nwaves = gridDim.x * kWaves;
  for (uint32_t k = blockIdx.x * kWaves + wave; k < a.count; k += nwaves) {
    const uint32_t amp = a.list[k];
    const uint32_t len = a.seqlen[amp];
    const uint32_t nw = (len + 31u) >> 5;
    const uint64_t * gs = a.seqs + a.seq_off[amp];
    for (uint32_t w = lane; w < nw + 2u; w += 64u) { sw[w] = (w < nw) ? gs[w] : 0ull; }
    wave_lds_sync();
    (void)enumerate_variants(sw, len, zob, lane, [&](const Slots & s) {
      if (PASS == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (s.ok[i]) {
            atomicAnd(&a.fbits[flex_word(s.hs[i], a.fsize, a.finv)], ~(unsigned long long)a.fpat[s.hs[i] & 0xFFFFu]);
          }
          nvar += (unsigned long long)__popcll(__ballot(s.ok[i]));
        }
      } else {
        uint64_t word[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          word[i] = s.ok[i] ? (uint64_t)a.fbits[flex_word(s.hs[i], a.fsize, a.finv)] : ~0ull;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool pass = s.ok[i] && ((word[i] & a.fpat[s.hs[i] & 0xFFFFu]) == 0ull);   // bloomflex_get
          nvar += (unsigned long long)__popcll(__ballot(s.ok[i]));
          const uint64_t m = __ballot(pass);
          if (m != 0ull) {
            if (pass) {
              swa_task t;
              t.hash = s.hs[i]; t.heavy = amp; t.code = s.code[i];
              stage[nstage + (uint32_t)__popcll(m & lane_lt)] = t;
            }
            nstage += (uint32_t)__popcll(m);
            if (nstage > kStage - 64u) { flush(); }
          }
        }
      }
    });
    __builtin_amdgcn_wave_barrier();
  }
  if (PASS == 1 && nstage != 0u) { flush(); }
  if (lane == 0 && nvar != 0ull) { atomicAdd(a.variant_counter, nvar); }
}

// ---- CSR assembly: counts -> offsets (exclusive scan), edges -> rows, sort rows ------
constexpr int kScanItems = 8;
constexpr int kScanBlock = 256;
constexpr int kScanTile = kScanItems * kScanBlock;

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t * smem, uint64_t & total) {
  // v: per-thread value; returns exclusive prefix within the block (256 threads)
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint64_t incl = v;
#pragma unroll
  for (unsigned d = 1; d < 64; d <<= 1) {
    const uint64_t t = swa_shfl_up_u64(incl, d);
    if (lane >= (int)d) { incl += t; }
  }
  if (lane == 63) { smem[wave] = incl; }
  __syncthreads();
  uint64_t wave_off = 0;
  uint64_t tot = 0;
  for (int w = 0; w < kScanBlock / 64; ++w) {
    if (w < wave) { wave_off += smem[w]; }
    tot += smem[w];
  }
  __syncthreads();
  total = tot;
  return wave_off + incl - v;
}

// (T = uint32_t: plain counts; T = unsigned long long: the slots of an anchor table, whose low halves are the group sizes)
template <class T>
__global__ __launch_bounds__(kScanBlock) void k_scan_tiles(const T * __restrict__ counts, uint32_t n,
                                                           uint64_t * __restrict__ tile_sums) {
  __shared__ uint64_t smem[4];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  uint64_t v = 0;
  for (int i = 0; i < kScanItems; ++i) { if (base + i < n) { v += (uint32_t)counts[base + i]; } }
  uint64_t total;
  (void)block_exclusive_scan(v, smem, total);
  if (threadIdx.x == 0) { tile_sums[blockIdx.x] = total; }
}

__global__ __launch_bounds__(kScanBlock) void k_scan_sums(uint64_t * tile_sums, uint32_t tiles) {
  __shared__ uint64_t smem[4];
  uint64_t carry = 0;
  for (uint32_t base = 0; base < tiles; base += kScanBlock) {
    const uint32_t i = base + threadIdx.x;
    const uint64_t v = (i < tiles) ? tile_sums[i] : 0;
    uint64_t total;
    const uint64_t ex = block_exclusive_scan(v, smem, total);
    if (i < tiles) { tile_sums[i] = carry + ex; }
    carry += total;
  }
}

template <class T>
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const T * __restrict__ counts, uint32_t n,
                                                           const uint64_t * __restrict__ tile_sums,
                                                           uint64_t * __restrict__ offsets) {
  __shared__ uint64_t smem[4];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  uint32_t c[kScanItems];
  uint64_t v = 0;
  for (int i = 0; i < kScanItems; ++i) {
    c[i] = (base + i < n) ? (uint32_t)counts[base + i] : 0u;
    v += c[i];
  }
  uint64_t total;
  uint64_t run = tile_sums[blockIdx.x] + block_exclusive_scan(v, smem, total);
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) { offsets[base + i] = run; }
    run += c[i];
    if (base + i + 1 == n) { offsets[n] = run; }
  }
}

// total and largest fill of the per-wave edge segments -> out[0], out[1]; the members the pair kernels staged (the
// guard; seg_fill + nseg, + 2 nseg: pass 0, pass 1) -> out[2], out[3]
// (one workgroup of 1024 threads, eight loads in flight per thread: the 8192 fills of an MI355X in one round — with 256
// threads and a load per turn this single workgroup took 11 us of every step)
__global__ __launch_bounds__(1024) void k_seg_reduce(const uint32_t * __restrict__ seg_fill, uint32_t nseg,
                                                     unsigned long long * out) {
  __shared__ unsigned long long ssum[16], smax[16], sst0[16], sst1[16];
  unsigned long long sum = 0, mx = 0, st0 = 0, st1 = 0;
  for (uint32_t base = 0; base < nseg; base += 8192u) {
    uint32_t v[8], s0[8], s1[8];
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
      const uint32_t i = base + k * 1024u + threadIdx.x;
      v[k] = i < nseg ? seg_fill[i] : 0u;
      s0[k] = i < nseg ? seg_fill[(uint64_t)nseg + i] : 0u;
      s1[k] = i < nseg ? seg_fill[2ull * nseg + i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) { sum += v[k]; mx = mx > v[k] ? mx : v[k]; st0 += s0[k]; st1 += s1[k]; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long s2 = swa_shfl_xor_u64(sum, o), m2 = swa_shfl_xor_u64(mx, o);
    sum += s2; mx = mx > m2 ? mx : m2;
    st0 += swa_shfl_xor_u64(st0, o); st1 += swa_shfl_xor_u64(st1, o);
  }
  if ((threadIdx.x & 63u) == 0u) { ssum[threadIdx.x >> 6] = sum; smax[threadIdx.x >> 6] = mx; sst0[threadIdx.x >> 6] = st0; sst1[threadIdx.x >> 6] = st1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0, m = 0, a0 = 0, a1 = 0;
    for (int w = 0; w < 16; ++w) { s += ssum[w]; m = m > smax[w] ? m : smax[w]; a0 += sst0[w]; a1 += sst1[w]; }
    out[0] = s; out[1] = m; out[2] = a0; out[3] = a1;
  }
}

// start of every segment in the compacted edge list: exclusive prefix sum of the fills
__global__ __launch_bounds__(256) void k_seg_bases(const uint32_t * __restrict__ seg_fill, uint32_t nseg,
                                                   unsigned long long * __restrict__ bases) {
  __shared__ unsigned long long part[256];
  const uint32_t chunk = (nseg + 255u) / 256u;
  const uint32_t lo = threadIdx.x * chunk, hi = min(nseg, lo + chunk);
  unsigned long long sum = 0;
  for (uint32_t i = lo; i < hi; ++i) { sum += seg_fill[i]; }
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int t = 0; t < 256; ++t) { const unsigned long long v = part[t]; part[t] = run; run += v; }
  }
  __syncthreads();
  unsigned long long at = part[threadIdx.x];
  for (uint32_t i = lo; i < hi; ++i) { bases[i] = at; at += seg_fill[i]; }
}

// per-wave segments -> one flat list of (source << 32 | target) links, in no particular order
__global__ __launch_bounds__(256) void k_seg_compact(const uint64_t * __restrict__ edges,
                                                     const uint32_t * __restrict__ seg_fill, uint32_t nseg, uint64_t seg_cap,
                                                     const unsigned long long * __restrict__ bases,
                                                     uint64_t * __restrict__ out, uint64_t cap) {
  for (uint32_t s = blockIdx.x; s < nseg; s += gridDim.x) {
    const uint64_t fill = min((uint64_t)seg_fill[s], seg_cap);
    const uint64_t base = bases[s];
    for (uint64_t i = threadIdx.x; i < fill; i += blockDim.x) {
      if (base + i < cap) { out[base + i] = edges[(uint64_t)s * seg_cap + i]; }
    }
  }
}

__global__ __launch_bounds__(256) void k_scatter_edges(const uint64_t * __restrict__ edges,
                                                       const uint32_t * __restrict__ seg_fill, uint32_t nseg,
                                                       uint64_t seg_cap, uint32_t first,
                                                       const uint64_t * __restrict__ offsets, uint32_t * cursor,
                                                       uint32_t * __restrict__ neighbours, uint64_t cap) {
  for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const uint32_t filled = seg_fill[seg];                    // may exceed the capacity (the run is then repeated)
    const uint32_t fill = filled < seg_cap ? filled : (uint32_t)seg_cap;
    const uint64_t * e = edges + (uint64_t)seg * seg_cap;
    for (uint32_t j = threadIdx.x; j < fill; j += blockDim.x) {
      const uint64_t edge = e[j];
      const uint32_t k = (uint32_t)(edge >> 32) - first;
      const uint64_t at = offsets[k] + atomicAdd(&cursor[k], 1u);
      if (at < cap) { neighbours[at] = (uint32_t)edge; }
    }
  }
}

// rows are tiny (about two neighbours per amplicon): one thread insertion-sorts a row; rows
// longer than 64 are queued (long_rows[0] = count, then the row indices) and sorted by a whole
// wave each with an odd-even transposition sort
__global__ __launch_bounds__(256) void k_sort_rows(const uint64_t * __restrict__ offsets, uint32_t count,
                                                   uint32_t * neighbours, uint64_t cap, uint32_t * long_rows,
                                                   uint32_t long_cap) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
    const uint64_t b = offsets[k];
    const uint64_t e = offsets[k + 1];
    if (e > cap || e - b < 2) { continue; }
    if (e - b > 64) {
      const uint32_t at = atomicAdd(long_rows, 1u);
      if (at < long_cap) { long_rows[1u + at] = k; }
      continue;
    }
    for (uint64_t i = b + 1; i < e; ++i) {
      const uint32_t v = neighbours[i];
      uint64_t j = i;
      while (j > b && neighbours[j - 1] > v) { neighbours[j] = neighbours[j - 1]; --j; }
      neighbours[j] = v;
    }
  }
}

__global__ __launch_bounds__(64) void k_sort_long_rows(const uint64_t * __restrict__ offsets, uint32_t count,
                                                       uint32_t * neighbours, uint64_t cap,
                                                       const uint32_t * __restrict__ long_rows, uint32_t long_cap) {
  // one wave per queued row; if the queue overflowed (more than long_cap long rows) every row is
  // visited instead, as a fallback
  const int lane = threadIdx.x;
  const uint32_t queued = long_rows[0];
  const bool all = queued > long_cap;
  const uint32_t todo = all ? count : queued;
  for (uint32_t q = blockIdx.x; q < todo; q += gridDim.x) {
    const uint32_t k = all ? q : long_rows[1u + q];
    const uint64_t b = offsets[k];
    const uint64_t e = offsets[k + 1];
    const uint64_t len = e - b;
    if (e > cap || len <= 64) { continue; }
    for (uint64_t phase = 0; phase < len; ++phase) {
      for (uint64_t i = (phase & 1u) + 2ull * lane; i + 1 < len; i += 128) {
        const uint32_t x = neighbours[b + i];
        const uint32_t y = neighbours[b + i + 1];
        if (x > y) { neighbours[b + i] = y; neighbours[b + i + 1] = x; }
      }
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
    }
  }
}

#include "d1_fast.inc"
#include "d1_stream.inc"

int grid_for(const swa_ctx * ctx, uint64_t items, int per_block, int max_per_cu) {
  uint64_t blocks = (items + per_block - 1) / per_block;
  const uint64_t cap = (uint64_t)ctx->num_cus * max_per_cu;
  if (blocks > cap) { blocks = cap; }
  if (blocks < 1) { blocks = 1; }
  return (int)blocks;
}

}  // namespace

// loads this translation unit's code object (see swa_ctx_warmup): an empty launch
void swa_warm_d1(swa_ctx * ctx) {
  hipLaunchKernelGGL(k_table_clear, dim3(1), dim3(64), 0, ctx->stream, static_cast<swa_slot *>(nullptr), (uint64_t)0,
                     static_cast<uint64_t *>(nullptr), (uint64_t)0);
}

// shared with fastidious.hip
int swa_d1_rebuild_table(swa_ctx * ctx, const uint8_t * d_is_member) {
  const uint32_t n = ctx->db.n;
  hipLaunchKernelGGL(k_table_clear, dim3(grid_for(ctx, ctx->table_size, 256, 8)), dim3(256), 0, ctx->stream,
                     static_cast<swa_slot *>(ctx->d_table.ptr), ctx->table_size,
                     static_cast<uint64_t *>(ctx->d_bloom.ptr), ctx->bloom_words);
  hipLaunchKernelGGL(k_table_insert, dim3(grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream,
                     static_cast<const uint64_t *>(ctx->d_seqhash.ptr), n, d_is_member,
                     static_cast<swa_slot *>(ctx->d_table.ptr), ctx->table_size - 1,
                     static_cast<unsigned long long *>(ctx->d_bloom.ptr), ctx->bloom_words - 1,
                     static_cast<const uint64_t *>(ctx->d_patterns.ptr));
  SWA_HIP(ctx, hipGetLastError());
  return SWA_OK;
}


// ---- the anchored index (see d1_anchor.inc, d1_stream.inc) ----------------------------------------------------------
static bool anchored_enabled() {
  const char * e = getenv("SWA_D1_PLAIN");
  return !(e != nullptr && e[0] == '1');
}

// SWA_D1_OWNED_FULL=1: the database-wide table and Bloom filter are built as well (test hook: the plain kernel's structures)
static bool owned_index_enabled() {
  const char * e = getenv("SWA_D1_OWNED_FULL");
  return !(e != nullptr && e[0] == '1');
}

// shortest seed the anchored passes serve: two windows and a nucleotide (65 with the 32-nt windows of rounds 1-3)
static uint32_t anchor_minlen(const swa_ctx * ctx) { return ctx->anchor_a + ctx->anchor_b + 2u * ctx->anchor_w + 1u; }

// whether the anchored passes may be used at all for this database (decided at index build)
static bool anchor_applicable(const swa_ctx * ctx) {
  return anchored_enabled() && ctx->db.longest >= kMinAnchoredLen && ctx->db.longest <= 64u * kPrefetchWords * 32u - 64u;
}

// 16-byte quads per amplicon line: by the longest sequence of the database (64 bytes hold 7 words, 128 hold 15, 256 hold 31;
// longer sequences are cut off in their lines and served by the plain kernel)
static uint32_t line_quads_for(const swa_ctx * ctx) { return ctx->db.longest <= 160u ? 4u : (ctx->db.longest <= 480u ? 8u : 16u); }

// widest anchor windows, in words of 32 nt (anchor_nwin_for).  SWA_D1_ANCHOR_W=32 / 64 / 128 caps it (comparison switch).
static uint32_t anchor_max_nwin(const swa_ctx *) {
  uint32_t cap = 4u;
  if (const char * e = getenv("SWA_D1_ANCHOR_W")) { cap = std::min(4u, std::max(1u, (uint32_t)atoi(e) / 32u)); }
  return cap;
}
// groups of 65..pair_big members go to the pair kernel as well (one workgroup each); SWA_D1_PAIR_BIG=64 leaves
// them to the tiled kernel (test switch)
static bool member_index_enabled();
static uint32_t pair_big_limit() {
  const char * env = getenv("SWA_D1_PAIR_BIG");
  return env != nullptr ? std::min<uint32_t>(kPairBigCap, std::max<uint32_t>(kSeedsPerItem, (uint32_t)atoi(env))) : kPairBigCap;
}

// Shortest sequence and the population of every width class: facts of the uploaded database (k_db_lengths), read once per
// upload.  The work lists of an index are laid out from the populations (list_regions).
static int ensure_db_lengths(swa_ctx * ctx) {
  if (ctx->props_ready) { return SWA_OK; }
  SWA_TRY(swa_reserve(ctx, ctx->d_rank_tmp, std::max<size_t>(ctx->d_rank_tmp.bytes, 64)));
  auto * out = static_cast<uint32_t *>(ctx->d_rank_tmp.ptr);
  SWA_HIP(ctx, hipMemsetAsync(out, 0, 8 * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL(k_db_lengths, dim3(grid_for(ctx, ctx->db.n, 256, 8)), dim3(256), 0, ctx->stream, ctx->db.seqlen, ctx->db.n, out);
  uint32_t host[8] = {};
  SWA_HIP(ctx, hipMemcpyAsync(host, out, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->db_shortest = 0xFFFFFFFFu - host[0];
  for (uint32_t c = 0; c <= kWidthClasses; ++c) { ctx->class_pop[c] = host[1 + c]; }
  ctx->props_ready = true;
  return SWA_OK;
}

// Where the work lists of an index lie in its item buffer.  A group of width class c has a member of that class, and all
// its members are of classes <= c; a group on size list k has at least least[k] members: so list (c, k) holds at most
// min(pop[c], pop[<= c] / least[k]) groups, and the row tiles of class c (64 members each, of groups of more than 256)
// number at most pop[<= c] / 64 + pop[<= c] / (pair_big + 1), pair_big = the largest group a workgroup takes by pairs
// (256 unless SWA_D1_PAIR_BIG lowers it).  A database of one class — the usual one — pays for one class.
static ListRegions list_regions(const swa_ctx * ctx, uint64_t * total_items) {
  const uint32_t least[kListKinds] = {2, 5, 9, 17, 33, 65, pair_big_limit() + 1u};
  ListRegions r{};
  uint64_t at = 0, upto = 0;
  for (uint32_t c = 0; c < kWidthClasses; ++c) {
    upto += ctx->class_pop[c];
    for (uint32_t k = 0; k < kListKinds; ++k) {
      r.at[c][k] = at;
      uint64_t room = 64;
      // (the tiled kernel's items: a row tile against kTiledCols column tiles — per row tile at most what the largest group has)
      constexpr uint64_t per_row_tile = (tiled_item_count(kStreamGroupCap) + kStreamGroupCap / 64u - 1u) / (kStreamGroupCap / 64u);
      if (ctx->class_pop[c] != 0) { room += k + 1 < kListKinds ? std::min<uint64_t>(ctx->class_pop[c], upto / least[k]) : per_row_tile * (upto / 64 + upto / least[k]); }
      at += room;
    }
    r.at[c][kListKinds] = at;
  }
  *total_items = at;
  return r;
}

// abundance rank of every amplicon (k_abundance_rank: three streaming passes); flags[1] is raised when the database is
// not in abundance order
static int launch_abundance_rank(swa_ctx * ctx) {
  if (ctx->rank_ready) {                                    // (a fact of the uploaded database: once per upload)
    if (ctx->db_unordered) { SWA_HIP(ctx, hipMemsetAsync(static_cast<uint32_t *>(ctx->d_flags.ptr) + 1, 1, sizeof(uint32_t), ctx->stream)); }
    return SWA_OK;
  }
  ctx->rank_ready = true;
  const uint32_t n = ctx->db.n;
  const uint32_t tiles = (n + 255u) / 256u;
  SWA_TRY(swa_reserve(ctx, ctx->d_arank, uint64_t(n) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_rank_tmp, uint64_t(tiles) * sizeof(uint32_t)));
  auto * rank = static_cast<uint32_t *>(ctx->d_arank.ptr);
  auto * tile_last = static_cast<uint32_t *>(ctx->d_rank_tmp.ptr);
  hipLaunchKernelGGL(k_abundance_rank, dim3(grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, ctx->db.abundance, n, rank,
                     static_cast<uint32_t *>(ctx->d_flags.ptr), tile_last);
  hipLaunchKernelGGL(k_abundance_rank_carry, dim3(1), dim3(256), 0, ctx->stream, tile_last, tiles);
  hipLaunchKernelGGL(k_abundance_rank_fill, dim3(grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, n, rank, tile_last);
  return SWA_OK;
}

__global__ void k_set_flags(uint32_t * flags, uint32_t unserved, uint32_t shortest_code) { flags[3] = unserved; flags[6] = shortest_code; }
__global__ void k_set_guard_made(unsigned long long * guard, uint32_t made0, uint32_t made1) { guard[0] = made0; guard[1] = made1; }

// ---- the streaming build (d1_stream.inc) ------------------------------------------------------
enum { kSbLines = 0, kSbRec = 1, kSbFp = 5, kSbCnt = 8, kSbTile = 10, kSbStart = 12, kSbPartial = 14, kSbScal = 16,
       kSbMembers = 17, kSbOver = 19, kSbKind = 20, kSbLinkA = 22, kSbLinkB = 23, kSbHeavy = 24, kSbMTable = 26, kSbMBloom = 27, kSbSched = 28 };

struct PartPlan { uint32_t levels; uint32_t bits[4]; uint32_t total; };
static PartPlan plan_levels(uint32_t total_bits, uint32_t max_bits = kPartMaxBits) {
  PartPlan p{};
  p.total = std::max(1u, total_bits);
  p.levels = (p.total + max_bits - 1) / max_bits;
  for (uint32_t l = 0; l < p.levels; ++l) { p.bits[l] = p.total / p.levels + (l < p.total % p.levels ? 1u : 0u); }
  return p;
}

// One multi-level partition (d1_stream.inc) over up to kMaxIdx record sets.  Level l reads what level l - 1 wrote
// (ping / pong) and leaves, in starts[], the first record of every bucket; the last level's buckets are the result.
struct PartJob {
  uint32_t nidx = 1;
  const unsigned long long * in[kMaxIdx] = {};      // level-0 input (may be buf[i][1])
  const uint32_t * in_f[kMaxIdx] = {};
  unsigned long long * buf[kMaxIdx][2] = {};
  uint32_t * buf_f[kMaxIdx][2] = {};
  const uint64_t * cstart0[kMaxIdx] = {};           // level-0 chunks (nullptr: regular, PartIdx::cstride)
  uint64_t cstride0[kMaxIdx] = {};
  const uint32_t * csize0[kMaxIdx] = {};
  uint32_t csize_cap = 0, chunks0 = 1;
  bool single0 = true;
  uint64_t max_records = 0, max_tiles0 = 0;         // host-side upper bounds (per set)
  uint64_t out_cap = 0;                             // entries every buf[][] holds
  uint32_t bias = 0, top_bit = 32;
  uint32_t tile = 4096;                             // records per tile (2048 for the records that carry fingerprints)
  bool hist0_done = false;                          // the flat counts of level 0 are there already (k_keys<W, true>)
  PartPlan plan{};
  uint32_t * out32[kMaxIdx] = {};                   // != nullptr: the last level writes only the low halves, here
  // scratch, per set
  uint32_t * cnt[kMaxIdx] = {}; uint32_t * ctile[kMaxIdx] = {}; uint64_t * starts[kMaxIdx] = {}; uint32_t * partial[kMaxIdx] = {};
  uint32_t * total[kMaxIdx] = {};
  uint64_t starts_stride = 0;
  // results
  const unsigned long long * out[kMaxIdx] = {}; const uint32_t * out_f[kMaxIdx] = {}; const uint64_t * bstart[kMaxIdx] = {};
  uint32_t buckets = 0;
  int last = 0;                                     // buf[i][last] holds the result, buf[i][last ^ 1] is free
};

// scratch sizes of a job (entries): flat counts, tile table, chunk starts (per half), scan partials
static void part_scratch(const PartJob & j, uint64_t * cnt, uint64_t * ctile, uint64_t * starts_half, uint64_t * partial) {
  uint64_t chunks = j.chunks0, c_max = 0, t_max = 0, s_max = 0;
  bool single = j.single0;
  for (uint32_t l = 0; l < j.plan.levels; ++l) {
    const uint64_t tiles = (l == 0 ? j.max_tiles0 : j.max_records / j.tile + chunks + 1);
    c_max = std::max(c_max, (tiles << j.plan.bits[l]) + 2);
    t_max = std::max(t_max, chunks + 2 + tiles + 2);              // (the tile table of the chunks, and behind it the chunk of every tile)
    chunks = (single ? 1 : chunks) << j.plan.bits[l];
    single = false;
    s_max = std::max(s_max, chunks + 2);
  }
  *cnt = c_max; *ctile = t_max; *starts_half = s_max; *partial = c_max / kFlatChunk + 2;
}

// several small buffers zeroed by ONE launch (a step clears about ten; each hipMemsetAsync is a launch of 4-5 us)
struct ClearList { uint32_t * p[8]; uint64_t words[8]; uint32_t n; };
__global__ __launch_bounds__(256) void k_clear_many(const ClearList c) {
  for (uint32_t k = 0; k < c.n; ++k) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < c.words[k]; i += (uint64_t)gridDim.x * blockDim.x) { c.p[k][i] = 0u; }
  }
}
static void clear_add(ClearList & c, void * p, uint64_t bytes) {          // (bytes: a multiple of 4)
  c.p[c.n] = static_cast<uint32_t *>(p); c.words[c.n] = bytes / 4; ++c.n;
}
static int clear_launch(swa_ctx * ctx, const ClearList & c) {
  uint64_t most = 0;
  for (uint32_t k = 0; k < c.n; ++k) { most = std::max(most, c.words[k]); }
  if (c.n == 0 || most == 0) { return SWA_OK; }
  hipLaunchKernelGGL(k_clear_many, dim3(grid_for(ctx, most, 256, 8)), dim3(256), 0, ctx->stream, c);
  SWA_HIP(ctx, hipGetLastError());
  return SWA_OK;
}

// (experiment: SWA_D1_LINK_SPAN=0 cuts the first level's tiles chunk by chunk, as rounds 3-5 did)
static bool part_span_enabled() {
  static const bool on = [] { const char * e = getenv("SWA_D1_LINK_SPAN"); return !(e != nullptr && e[0] == '0'); }();
  return on;
}

static int run_partition(swa_ctx * ctx, PartJob & j) {
  uint64_t chunks = j.chunks0;
  bool single = j.single0;
  uint32_t used_bits = 0;
  const int cu_grid = ctx->num_cus * 8;
  for (uint32_t l = 0; l < j.plan.levels; ++l) {
    const uint32_t bits = j.plan.bits[l];
    used_bits += bits;
    const bool last_level = l + 1 == j.plan.levels;
    PartArgs a{};
    a.single_seg = single ? 1u : 0u; a.bits = bits; a.shift = j.top_bit - used_bits; a.bias = j.bias; a.tile = j.tile;
    // (the link segments: tiles over the chunks laid end to end — the kernels of the usual link tile have that form)
    a.span = (single && l == 0 && chunks > 1 && !j.hist0_done && j.tile == 4096 && bits <= kPartMaxBits && j.buf_f[0][0] == nullptr && part_span_enabled()) ? 1u : 0u;
    // runs of 16 consecutive tiles per XCD (xcd_tile; measured at 10 M amplicons: key partition 0.44 -> 0.33 ms, link partition
    // 0.39 -> 0.36, the same for runs of 8 .. 64) — unless there are fewer tiles than workgroups in flight, where the runs
    // would leave XCDs without work
    const uint64_t tiles = (l == 0 ? j.max_tiles0 : j.max_records / j.tile + chunks + 1);
    a.run_bits = tiles >= (uint64_t)cu_grid ? 4u : 0u;
    if (const char * e = getenv("SWA_D1_XCD_RUN_BITS")) { a.run_bits = (uint32_t)std::min(8, std::max(0, atoi(e))); }   // (experiment)
    for (uint32_t i = 0; i < j.nidx; ++i) {
      PartIdx & p = a.p[i];
      p.in = l == 0 ? j.in[i] : j.buf[i][(l - 1) & 1u];
      p.in_f = l == 0 ? j.in_f[i] : j.buf_f[i][(l - 1) & 1u];
      p.out = j.buf[i][l & 1u]; p.out_f = j.buf_f[i][l & 1u];
      p.out32 = last_level ? j.out32[i] : nullptr;
      p.out_cap = j.out_cap;
      p.cstart = l == 0 ? j.cstart0[i] : j.starts[i] + ((l - 1) & 1u) * j.starts_stride;
      p.cstride = l == 0 ? j.cstride0[i] : 0;
      p.csize = l == 0 ? j.csize0[i] : nullptr;
      p.csize_cap = j.csize_cap;
      p.chunks = (uint32_t)chunks;
      p.ctile = j.ctile[i]; p.cnt = j.cnt[i];
      p.tchunk = j.ctile[i] + chunks + 1; p.tchunk_cap = (uint32_t)std::min<uint64_t>(tiles + 1, 0xFFFFFFFFu);
      p.next_start = j.starts[i] + (l & 1u) * j.starts_stride;
      p.total = j.total[i];
    }
    // (a multiple of 8 workgroups: turn v of the tile loops then stays on XCD v mod 8 — xcd_tile)
    const dim3 grid_t((unsigned)((std::min<uint64_t>(std::max<uint64_t>(tiles, 1), (uint64_t)cu_grid) + 7) & ~7ull), j.nidx);
    if (chunks > 2048) { hipLaunchKernelGGL(k_part_tiles<1024>, dim3(1, j.nidx), dim3(1024), 0, ctx->stream, a); }
    else { hipLaunchKernelGGL(k_part_tiles<256>, dim3(1, j.nidx), dim3(256), 0, ctx->stream, a); }
    const bool bins1024 = bits > kPartMaxBits;                 // (the key records of the one-level form: 10 bits)
    if (l == 0 && j.hist0_done) { /* (k_keys has left the counts) */ }
    else if (bins1024) { hipLaunchKernelGGL((k_part_hist<1024, kPartTileMax, 256>), grid_t, dim3(256), 0, ctx->stream, a); }
    else if (j.tile == 8192) { hipLaunchKernelGGL((k_part_hist<512, 8192, 512>), grid_t, dim3(512), 0, ctx->stream, a); }
    else if (a.span != 0u) { hipLaunchKernelGGL((k_part_hist<512, kPartTileMax, 256, true>), grid_t, dim3(256), 0, ctx->stream, a); }
    else { hipLaunchKernelGGL((k_part_hist<512, kPartTileMax, 256>), grid_t, dim3(256), 0, ctx->stream, a); }
    FlatScanArgs f{};
    f.unit_bits = bits;
    for (uint32_t i = 0; i < j.nidx; ++i) { f.v[i] = j.cnt[i]; f.units[i] = j.ctile[i] + chunks; f.partial[i] = j.partial[i]; f.total[i] = j.total[i]; }
    const dim3 grid_f((unsigned)std::min<uint64_t>(((tiles << bits) + 1 + kFlatChunk - 1) / kFlatChunk, (uint64_t)cu_grid), j.nidx);
    hipLaunchKernelGGL(k_flat_sums, grid_f, dim3(256), 0, ctx->stream, f);
    hipLaunchKernelGGL(k_flat_apply, grid_f, dim3(256), 0, ctx->stream, f);
    {
      static const int part_threads = [] { const char * e = getenv("SWA_D1_PART_THREADS"); const int v = e != nullptr ? atoi(e) : 512; return v == 256 || v == 1024 ? v : 512; }();   // (512: key partition 0.269 -> 0.239, link partition 0.317 -> 0.303 ms at 10 M against 256; 1024: 0.244 / 0.384)
#define SWA_SCATTER(M, T, B)                                                                                                          \
      do {                                                                                                                            \
        constexpr size_t lds = part_scatter_lds(M, T, B);                                                                             \
        if (part_threads == 1024) { hipLaunchKernelGGL((k_part_scatter<M, T, B, 1024>), grid_t, dim3(1024), lds, ctx->stream, a); }   \
        else if (part_threads == 512) { hipLaunchKernelGGL((k_part_scatter<M, T, B, 512>), grid_t, dim3(512), lds, ctx->stream, a); } \
        else { hipLaunchKernelGGL((k_part_scatter<M, T, B, 256>), grid_t, dim3(256), lds, ctx->stream, a); }                          \
      } while (0)
      // (beyond 64 KB of dynamic LDS the attribute is asked for: it belongs to the function on a device — once per context)
#define SWA_SCATTER_WIDE(M, B, BIT)                                                                                                   \
      do {                                                                                                                            \
        constexpr size_t lds = part_scatter_lds(M, 8192, B);                                                                          \
        if ((ctx->part_lds_opt_in & (1u << BIT)) == 0u) {                                                                             \
          SWA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter<M, 8192, B, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
          ctx->part_lds_opt_in |= 1u << BIT;                                                                                          \
        }                                                                                                                             \
        hipLaunchKernelGGL((k_part_scatter<M, 8192, B, 1024>), grid_t, dim3(1024), lds, ctx->stream, a);                              \
      } while (0)
      const bool payload = j.buf_f[0][0] != nullptr;           // (a second array travels with index 0's records: nobody's since round 6)
      if (!payload && j.tile == 8192 && !bins1024) {
        if (last_level && j.out32[0] != nullptr) { SWA_SCATTER_WIDE(2, 512, 1); } else { SWA_SCATTER_WIDE(0, 512, 2); }
      }
      else if (a.span != 0u) {
        // (4096 x 512, 512 threads: the form the link partition runs in; SWA_D1_PART_THREADS keeps the chunk-by-chunk kernels)
        constexpr size_t lds = part_scatter_lds(0, 4096, 512);
        if (last_level && j.out32[0] != nullptr) { hipLaunchKernelGGL((k_part_scatter<2, 4096, 512, 512, true>), grid_t, dim3(512), lds, ctx->stream, a); }
        else { hipLaunchKernelGGL((k_part_scatter<0, 4096, 512, 512, true>), grid_t, dim3(512), lds, ctx->stream, a); }
      }
      else if (last_level && j.out32[0] != nullptr) { SWA_SCATTER(2, 4096, 512); }
      else if (bins1024 && j.tile == 8192) { if (payload) { SWA_SCATTER_WIDE(1, 1024, 0); } else { SWA_SCATTER_WIDE(0, 1024, 3); } }
      else if (bins1024 && j.tile == 4096) { if (payload) { SWA_SCATTER(1, 4096, 1024); } else { SWA_SCATTER(0, 4096, 1024); } }
      else if (bins1024) { if (payload) { SWA_SCATTER(1, 2048, 1024); } else { SWA_SCATTER(0, 2048, 1024); } }
      else if (payload) { SWA_SCATTER(1, 2048, 512); }
      else if (j.tile == 2048) { SWA_SCATTER(0, 2048, 512); }
      else { SWA_SCATTER(0, 4096, 512); }
#undef SWA_SCATTER
#undef SWA_SCATTER_WIDE
    }
    chunks = (single ? 1 : chunks) << bits;
    single = false;
    hipLaunchKernelGGL(k_part_starts, dim3((unsigned)std::min<uint64_t>((chunks + 256) / 256, (uint64_t)cu_grid), j.nidx), dim3(256), 0, ctx->stream, a);
    SWA_HIP(ctx, hipGetLastError());
  }
  j.buckets = (uint32_t)chunks;
  j.last = (int)((j.plan.levels - 1) & 1u);
  for (uint32_t i = 0; i < j.nidx; ++i) {
    j.out[i] = j.buf[i][j.last]; j.out_f[i] = j.buf_f[i][j.last];
    j.bstart[i] = j.starts[i] + (uint64_t)j.last * j.starts_stride;
  }
  return SWA_OK;
}

// the amplicon lines of the uploaded database (once per upload; needs the abundance ranks)
static int ensure_lines(swa_ctx * ctx) {
  const uint32_t lq = line_quads_for(ctx);
  if (ctx->lines_ready && ctx->lines_quads == lq) { return SWA_OK; }
  const uint32_t n = ctx->db.n;
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbLines], (uint64_t)n * lq * 16u + 64u));   // (+ what window_at may read behind the last line)
  auto * lines = static_cast<uint4 *>(ctx->d_stream[kSbLines].ptr);
  const auto * rank = static_cast<const uint32_t *>(ctx->d_arank.ptr);
  SWA_HIP(ctx, hipMemsetAsync(reinterpret_cast<uint8_t *>(lines) + (uint64_t)n * lq * 16u, 0, 64, ctx->stream));
  swa_t0(ctx, 15);
  const dim3 grid(grid_for(ctx, n, 256, 8));
  if (lq == 4u) { hipLaunchKernelGGL(k_lines_build<4>, grid, dim3(256), 0, ctx->stream, ctx->db.seqs, ctx->db.seq_off, ctx->db.seqlen, rank, n, lines); }
  else if (lq == 8u) { hipLaunchKernelGGL(k_lines_build<8>, grid, dim3(256), 0, ctx->stream, ctx->db.seqs, ctx->db.seq_off, ctx->db.seqlen, rank, n, lines); }
  else { hipLaunchKernelGGL(k_lines_build<16>, grid, dim3(256), 0, ctx->stream, ctx->db.seqs, ctx->db.seq_off, ctx->db.seqlen, rank, n, lines); }
  swa_t1(ctx, 15);
  SWA_HIP(ctx, hipGetLastError());
  ctx->lines_ready = true;
  ctx->lines_quads = lq;
  return SWA_OK;
}

// The two anchor indexes of the whole database (or of the routed id lists) by the streaming build: members in group
// order (d_stream[kSbMembers + which]: ids), the work lists of the pair kernels, the flags build_owned_index reads
// ([0] identical sequences [2] partition too coarse [3] unserved seeds [4, 5] oversized groups [6] shortest sequence).
static int build_stream_index(swa_ctx * ctx, uint32_t dup_first, uint32_t dup_count) {
  ctx->anchor_ready = false;
  ctx->stream_index = false;
  ctx->list_counts_ready = false;
  const uint32_t n = ctx->db.n;
  swa_lap(ctx, "(build starts)");
  SWA_TRY(ensure_db_lengths(ctx));
  swa_lap(ctx, "db lengths");
  SWA_TRY(ensure_lines(ctx));
  swa_lap(ctx, "amplicon lines");
  const uint32_t lq = ctx->lines_quads;
  uint64_t item_room = 0;
  const ListRegions regions = list_regions(ctx, &item_room);
  const bool by_rec = ctx->route_rec[0] != nullptr;           // routed, and the key records themselves arrived: no k_keys pass
  const bool routed = ctx->route_ids[0] != nullptr || by_rec;
  const uint64_t records = routed ? std::max<uint64_t>(std::max(ctx->route_m[0], ctx->route_m[1]), 1) : n;
  // buckets of ~10 000 records for k_group1: ONE partition level of up to 10 bits at 10 M amplicons
  const uint32_t target = kG1Target, level_bits = 10u;
  uint32_t total_bits = 1;
  while ((records >> total_bits) > target && total_bits < 3 * kPartMaxBits) { ++total_bits; }
  total_bits = std::min<uint32_t>(total_bits + ctx->stream_extra_bits, 3 * kPartMaxBits);
  PartJob j;
  j.nidx = 2;
  j.plan = plan_levels(total_bits, level_bits);
  j.max_records = records;
  j.out_cap = records + 1;
  // (tiles of 2048 records for 512 bins; one level of 1024 bins: 4096, or the flat count array — bins x tiles — and the 16-byte runs
  // a tile leaves per bin cost more than the saved level: 0.56 -> 0.43 ms at 10 M amplicons)
  {
    // tiles of 2048 records for 512 bins; one level of 1024 bins: 4096 — or 8192 when k_keys takes the histogram (k_part_hist
    // holds 4096 a workgroup): a tile then leaves runs of 64 bytes per bin, whole lines (key partition 0.239 -> 0.225 ms at 10 M)
    const bool one_wide_level = total_bits > kPartMaxBits && j.plan.levels == 1;
    const char * kh = getenv("SWA_D1_KEYS_HIST");
    const bool hist_by_keys = !routed && !(kh != nullptr && kh[0] == '0');
    j.tile = one_wide_level ? (hist_by_keys ? 8192 : 4096) : 2048;
    if (const char * e = getenv("SWA_D1_KEY_TILE")) {          // (experiment)
      const int v = atoi(e);
      if (v == 2048 || (v == 4096 && one_wide_level) || (v == 8192 && one_wide_level && hist_by_keys)) { j.tile = (uint32_t)v; }
    }
  }
  j.max_tiles0 = records / j.tile + 2;
  j.chunks0 = 1; j.single0 = true; j.top_bit = 32; j.bias = 0;
  uint64_t e_cnt, e_tile, e_start, e_partial;
  part_scratch(j, &e_cnt, &e_tile, &e_start, &e_partial);
  const uint64_t buckets = 1ull << total_bits;
  e_partial = std::max<uint64_t>(e_partial, ((uint64_t)kListsPerIndex * buckets + 1) / kFlatChunk + 2);
  for (int i = 0; i < 2; ++i) {
    for (int h = 0; h < 2; ++h) { SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbRec + 2 * i + h], (records + 1) * sizeof(uint64_t))); }
    SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbCnt + i], e_cnt * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbTile + i], e_tile * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbStart + i], (2 * e_start + 4) * sizeof(uint64_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbPartial + i], e_partial * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbMembers + i], (records + 1) * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbKind + i], ((uint64_t)kListsPerIndex * buckets + 2) * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_aitems[i], (item_room + 1) * sizeof(swa_item)));
  }
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbScal], 64 * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbOver], ((uint64_t)n + 8) & ~3ull));
  SWA_TRY(swa_reserve(ctx, ctx->d_acounters, kCounterWords * sizeof(uint32_t)));
  {
    ClearList c{};
    clear_add(c, ctx->d_acounters.ptr, kCounterWords * sizeof(uint32_t));
    clear_add(c, ctx->d_guard.ptr, 8 * sizeof(uint64_t));      // the guard's index counters: this build's
    clear_add(c, ctx->d_stream[kSbOver].ptr, ((uint64_t)n + 8) & ~3ull);
    // (the entry behind the last bucket's counts of each index: the scan of the counts reads one past the end)
    for (int i = 0; i < 2; ++i) { clear_add(c, static_cast<uint32_t *>(ctx->d_stream[kSbKind + i].ptr) + (uint64_t)kListsPerIndex * buckets, sizeof(uint32_t)); }
    SWA_TRY(clear_launch(ctx, c));
  }
  swa_lap(ctx, "buffers reserved, counters cleared");
  auto * dflags = static_cast<uint32_t *>(ctx->d_flags.ptr);                // (cleared by the caller: index build, or the retry)
  auto * scal = static_cast<uint64_t *>(ctx->d_stream[kSbScal].ptr);      // [0..3] level-0 chunk tables, [8 + i] totals (u32)
  const uint32_t win_a = ctx->anchor_a, win_b = ctx->anchor_b;
  const uint32_t minlen = anchor_minlen(ctx), window_mode = (win_a != 0 || win_b != 0) ? 1u : 0u;

  // ---- keys: records into the PONG halves (level 0 reads them from there), fingerprints likewise
  KeyArgs k{};
  k.lines = static_cast<const uint4 *>(ctx->d_stream[kSbLines].ptr);
  k.line_quads = lq;
  k.seqs = ctx->db.seqs; k.seq_off = ctx->db.seq_off;
  k.n = n;
  for (int i = 0; i < 2; ++i) {
    k.list[i] = routed && !by_rec ? ctx->route_ids[i] : nullptr;
    k.list_count[i] = routed ? ctx->route_m[i] : 0;
    k.rec[i] = static_cast<unsigned long long *>(ctx->d_stream[kSbRec + 2 * i + 1].ptr);
  }
  k.owner_rank = ctx->owner_rank; k.owner_world = ctx->owner_world;
  k.win_a = win_a; k.win_b = win_b; k.minlen = minlen; k.window_mode = window_mode;
  k.nwin = ctx->anchor_w / 32u;
  k.flags = dflags;
  k.guard = static_cast<unsigned long long *>(ctx->d_guard.ptr);
  // (single GPU, the whole database keyed: the short sequences and their possible neighbours become members of the plain
  // kernel's table; a rank of a multi-GPU job keeps the database-wide table for them)
  k.mark_short = (!routed && ctx->owner_world == 1u && member_index_enabled()) ? static_cast<uint8_t *>(ctx->d_stream[kSbOver].ptr) : nullptr;
  if (const char * e = getenv("SWA_D1_GUARD_TEST")) {         // (test hook: a wrong index, on purpose; "...-once": only the first build)
    static int builds = 0;
    const bool once = strstr(e, "-once") != nullptr;
    if (!once || builds++ == 0) { k.fault = e[0] == 'm' ? 1u : (e[0] == 'd' ? 2u : 0u); }
  }
  if (routed) {
    hipLaunchKernelGGL(k_set_flags, dim3(1), dim3(1), 0, ctx->stream, dflags,
                       ctx->db_shortest < minlen ? 1u : 0u, 0xFFFFFFFFu - ctx->db_shortest);
  }
  if (by_rec) {   // (the records were made by the ranks that hold the amplicons' slices: what k_keys would have counted)
    hipLaunchKernelGGL(k_set_guard_made, dim3(1), dim3(1), 0, ctx->stream, k.guard, ctx->route_m[0], ctx->route_m[1]);
  }
  // the first partition level's histogram is taken on the way (one read pass over the records less: 0.07 ms at 10 M);
  // not for routed id lists (their length is the device's to know), SWA_D1_KEYS_HIST=0: comparison switch
  const char * env_kh = getenv("SWA_D1_KEYS_HIST");
  const bool keys_hist = !routed && !(env_kh != nullptr && env_kh[0] == '0');
  swa_t0(ctx, 8);
  if (by_rec) { /* nothing to key */ }
  else if (keys_hist) {
    const uint32_t ntiles = (uint32_t)((records + j.tile - 1) / j.tile);
    for (int i = 0; i < 2; ++i) { k.hist_cnt[i] = static_cast<uint32_t *>(ctx->d_stream[kSbCnt + i].ptr); }
    k.hist_bits = j.plan.bits[0]; k.hist_tile = j.tile; k.hist_ntiles = ntiles;
    k.hist_run_bits = j.max_tiles0 >= (uint64_t)ctx->num_cus * 8 ? 4u : 0u;
    const dim3 hgrid((unsigned)((std::min<uint64_t>(ntiles, (uint64_t)ctx->num_cus * 8) + 7) & ~7ull), 1u);
    hipLaunchKernelGGL(k_keys<true>, hgrid, dim3(256), 0, ctx->stream, k);
    j.hist0_done = true;
  } else {
    const dim3 kgrid((unsigned)grid_for(ctx, records, 256, 8), routed ? 2u : 1u);
    hipLaunchKernelGGL(k_keys<false>, kgrid, dim3(256), 0, ctx->stream, k);
  }
  swa_t1(ctx, 8);
  swa_lap(ctx, "k_keys");

  // ---- partition by the top bits of the key
  for (int i = 0; i < 2; ++i) {
    j.buf[i][0] = static_cast<unsigned long long *>(ctx->d_stream[kSbRec + 2 * i].ptr);
    j.buf[i][1] = static_cast<unsigned long long *>(ctx->d_stream[kSbRec + 2 * i + 1].ptr);
    j.in[i] = by_rec ? ctx->route_rec[i] : j.buf[i][1];
    j.cstart0[i] = nullptr; j.cstride0[i] = routed ? ctx->route_m[i] : n;   // (one chunk: [0, records of this index))
    j.cnt[i] = static_cast<uint32_t *>(ctx->d_stream[kSbCnt + i].ptr);
    j.ctile[i] = static_cast<uint32_t *>(ctx->d_stream[kSbTile + i].ptr);
    j.starts[i] = static_cast<uint64_t *>(ctx->d_stream[kSbStart + i].ptr);
    j.partial[i] = static_cast<uint32_t *>(ctx->d_stream[kSbPartial + i].ptr);
    j.total[i] = reinterpret_cast<uint32_t *>(scal + 8 + i);
  }
  // (no second array travels with the records: the fingerprints of rounds 3-5 served k_group1's search for identical
  // sequences, which the prefix pass of the pair kernels now does on the sequences themselves)
  j.starts_stride = e_start + 2;
  swa_t0(ctx, 9);
  SWA_TRY(run_partition(ctx, j));
  swa_t1(ctx, 9);
  swa_lap(ctx, "key partition");
  if (!ctx->guard_keys_done) {
    // the guard's second opinion on the key records, once per uploaded database: from the packed database against what the
    // partition holds (k_guard_db / k_guard_records); compared at the next guard_check
    auto * gsum = static_cast<unsigned long long *>(ctx->d_guard.ptr) + 16;
    SWA_HIP(ctx, hipMemsetAsync(gsum, 0, 8 * sizeof(uint64_t), ctx->stream));
    GuardDbArgs gd{};
    gd.seqs = ctx->db.seqs; gd.seq_off = ctx->db.seq_off; gd.seqlen = ctx->db.seqlen; gd.n = n;
    for (int i = 0; i < 2; ++i) { gd.list[i] = k.list[i]; gd.list_count[i] = routed ? ctx->route_m[i] : 0; gd.list_rec[i] = by_rec ? ctx->route_rec[i] : nullptr; }
    gd.owner_rank = k.owner_rank; gd.owner_world = k.owner_world; gd.win_a = win_a; gd.win_b = win_b; gd.nwin = k.nwin;
    gd.out = gsum;
    hipLaunchKernelGGL(k_guard_db, dim3((unsigned)grid_for(ctx, records, 256, 8), routed ? 2u : 1u), dim3(256), 0, ctx->stream, gd);
    GuardRecArgs gr{};
    for (int i = 0; i < 2; ++i) { gr.rec[i] = j.out[i]; gr.total[i] = j.total[i]; }
    gr.out = gsum + 3;
    hipLaunchKernelGGL(k_guard_records, dim3((unsigned)grid_for(ctx, records, 256, 8), 2u), dim3(256), 0, ctx->stream, gr);
    ctx->guard_keys_pending = true;
  }

  swa_lap(ctx, "guard: second opinion");
  // ---- groups
  GroupArgs g{};
  for (int i = 0; i < 2; ++i) {
    GroupIdx & x = g.g[i];
    x.rec = const_cast<unsigned long long *>(j.out[i]);
    x.bstart = j.bstart[i];
    x.buckets = j.buckets;
    x.members = static_cast<uint32_t *>(ctx->d_stream[kSbMembers + i].ptr);
    x.items_tmp = j.buf[i][j.last ^ 1];
    x.kind_cnt = static_cast<uint32_t *>(ctx->d_stream[kSbKind + i].ptr);
  }
  g.pair_big = pair_big_limit(); g.group_cap = kStreamGroupCap;   // (the tiled pair kernel serves every group up to that)
  if (const char * env_cap = getenv("SWA_D1_GROUP_CAP")) { g.group_cap = std::max<uint32_t>(g.pair_big, (uint32_t)atoi(env_cap)); }   // (experiments)
  g.flags = dflags;
  g.guard = static_cast<unsigned long long *>(ctx->d_guard.ptr);
  g.over = static_cast<uint8_t *>(ctx->d_stream[kSbOver].ptr);
  swa_t0(ctx, 10);
  {
    if (!ctx->g1_lds_opt_in) {
      // (128 KB of dynamic LDS: above the 64 KB a kernel gets unasked.  The attribute belongs to the function ON A DEVICE:
      // once per context, not once per process — swa_multi_* runs a context per GPU in one process)
      SWA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_group1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kG1LdsBytes));
      ctx->g1_lds_opt_in = true;
    }
    hipLaunchKernelGGL(k_group1, dim3((unsigned)std::min<uint64_t>(buckets, (uint64_t)ctx->num_cus), 2), dim3(kG1Threads), kG1LdsBytes, ctx->stream, g);
  }
  swa_t1(ctx, 10);                                            // (slot 10: k_group1 alone; the work lists below count in slot 7, the whole build)

  // ---- work lists
  FlatScanArgs f{};
  ListArgs la{};
  auto * acounters = static_cast<uint32_t *>(ctx->d_acounters.ptr);
  for (int i = 0; i < 2; ++i) {
    f.v[i] = g.g[i].kind_cnt; f.units[i] = nullptr; f.fixed[i] = (uint64_t)kListsPerIndex * buckets + 1;
    f.partial[i] = j.partial[i]; f.total[i] = reinterpret_cast<uint32_t *>(scal + 10 + i);
    ListIdx & x = la.x[i];
    x.items_tmp = g.g[i].items_tmp; x.bstart = g.g[i].bstart; x.kind_pos = g.g[i].kind_cnt; x.buckets = (uint32_t)buckets;
    x.l.items = static_cast<swa_item *>(ctx->d_aitems[i].ptr);
    x.l.region = regions;
    x.l.counters = acounters + kCounterBase + (uint32_t)i * kWidthClasses * 8u;
    x.l.pair_big = g.pair_big;
  }
  const dim3 grid_k((unsigned)std::min<uint64_t>(((uint64_t)kListsPerIndex * buckets + kFlatChunk) / kFlatChunk, (uint64_t)ctx->num_cus * 8), 2);
  hipLaunchKernelGGL(k_flat_sums, grid_k, dim3(256), 0, ctx->stream, f);
  hipLaunchKernelGGL(k_flat_apply, grid_k, dim3(256), 0, ctx->stream, f);
  hipLaunchKernelGGL(k_group_lists, dim3((unsigned)std::min<uint64_t>((buckets + 3) / 4, (uint64_t)ctx->num_cus * 8), 2), dim3(256), 0, ctx->stream, la);
  SWA_HIP(ctx, hipGetLastError());
  swa_lap(ctx, "groups + work lists");
  ctx->list_regions_items = item_room;
  ctx->anchor_ready = true;
  ctx->stream_index = true;
  ctx->guard_index = true;
  return SWA_OK;
}

// CSR of the links in the per-wave segments by the streaming route: the links partitioned by the top bits of their
// source (relative to `first`) until a bucket holds 2^r consecutive sources, then one wave per bucket (k_csr_bucket).
// link_cap: entries of the two internal link buffers; a run with more links than that leaves garbage and is repeated
// by the caller with bigger buffers (the total is known after the host has synchronised).
static bool stream_csr_enabled() {
  const char * e = getenv("SWA_D1_CSR");
  return !(e != nullptr && e[0] == 't');
}

// (the chunks of the first level: `chunks` runs of links, run c = links[cstart[c] .. + min(csize[c], csize_cap)) — the per-wave
// segments of the pair kernels, or the ranks' lists of a multi-GPU job gathered on one device)
static int csr_from_chunks(swa_ctx * ctx, uint32_t first, uint32_t count, const unsigned long long * links, const uint64_t * d_cstart, uint64_t cstride,
                           const uint32_t * d_csize, uint32_t chunks, uint32_t csize_cap, uint64_t max_tiles0, uint64_t link_cap,
                           uint64_t * d_offsets, uint32_t * d_neighbours, uint64_t cap) {
  uint32_t nbits = 1;
  while (nbits < 32 && ((uint64_t)1 << nbits) < count) { ++nbits; }
  uint32_t r = std::min<uint32_t>(8, nbits - 1);
  if (nbits - r > 2 * kPartMaxBits) { r = std::min<uint32_t>(kCsrMaxR, nbits - 2 * kPartMaxBits); }
  PartJob j;
  j.nidx = 1;
  j.plan = plan_levels(nbits - r);
  j.max_records = link_cap;
  j.out_cap = link_cap;
  j.tile = 4096;
  if (const char * e = getenv("SWA_D1_LINK_TILE")) { if (atoi(e) == 8192) { j.tile = 8192; } }   // (experiment: runs of 128 bytes, half the flat counts)
  j.max_tiles0 = max_tiles0;
  j.chunks0 = chunks; j.single0 = true; j.top_bit = nbits; j.bias = first;
  j.csize_cap = csize_cap;
  uint64_t e_cnt, e_tile, e_start, e_partial;
  part_scratch(j, &e_cnt, &e_tile, &e_start, &e_partial);
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbLinkA], (link_cap + 1) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbLinkB], (link_cap + 1) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbCnt], e_cnt * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbTile], e_tile * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbStart], (2 * e_start + 4) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbPartial], e_partial * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbScal], 64 * sizeof(uint64_t)));
  j.in[0] = links;
  j.buf[0][0] = static_cast<unsigned long long *>(ctx->d_stream[kSbLinkA].ptr);
  j.buf[0][1] = static_cast<unsigned long long *>(ctx->d_stream[kSbLinkB].ptr);
  j.cstart0[0] = d_cstart; j.cstride0[0] = cstride;
  j.csize0[0] = d_csize;
  j.cnt[0] = static_cast<uint32_t *>(ctx->d_stream[kSbCnt].ptr);
  j.ctile[0] = static_cast<uint32_t *>(ctx->d_stream[kSbTile].ptr);
  j.starts[0] = static_cast<uint64_t *>(ctx->d_stream[kSbStart].ptr);
  j.partial[0] = static_cast<uint32_t *>(ctx->d_stream[kSbPartial].ptr);
  auto * extra = static_cast<uint64_t *>(ctx->d_status.ptr) + 48;   // (status block, bytes [384, 512): [0] last CSR offset [1] links sorted)
  j.total[0] = reinterpret_cast<uint32_t *>(extra + 1);
  j.starts_stride = e_start + 2;
  swa_t0(ctx, 13);
  SWA_TRY(run_partition(ctx, j));
  swa_t1(ctx, 13);
  CsrArgs c{};
  c.links = j.out[0]; c.bstart = j.bstart[0]; c.buckets = j.buckets; c.r = r; c.first = first; c.count = count;
  c.offsets = d_offsets; c.neighbours = d_neighbours; c.cap = d_neighbours != nullptr ? cap : 0;
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbHeavy], ((uint64_t)j.buckets + 2) * sizeof(uint32_t)));
  c.heavy = static_cast<uint32_t *>(ctx->d_stream[kSbHeavy].ptr) + 1;
  c.heavy_count = static_cast<uint32_t *>(ctx->d_stream[kSbHeavy].ptr);
  c.end_copy = extra;
  SWA_HIP(ctx, hipMemsetAsync(c.heavy_count, 0, sizeof(uint32_t), ctx->stream));
  swa_t0(ctx, 14);
  // (as many workgroups as are resident together — every wave walks its share of the buckets; a grid of 8 a CU ran a second,
  // partly empty round behind the first)
  static const int csr_per_cu[2] = {
    [] { int nb = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_csr_bucket<8>, 256, 0) == hipSuccess && nb > 0 ? nb : 4; }(),
    [] { int nb = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_csr_bucket<9>, 256, 0) == hipSuccess && nb > 0 ? nb : 4; }() };
  const dim3 cgrid((unsigned)std::min<uint64_t>(((uint64_t)j.buckets + 3) / 4, (uint64_t)ctx->num_cus * (uint64_t)csr_per_cu[r <= 8 ? 0 : 1]));
  if (r <= 8) { hipLaunchKernelGGL(k_csr_bucket<8>, cgrid, dim3(256), 0, ctx->stream, c); }
  else { hipLaunchKernelGGL(k_csr_bucket<9>, cgrid, dim3(256), 0, ctx->stream, c); }
  hipLaunchKernelGGL(k_csr_bucket_big, dim3((unsigned)ctx->num_cus * 4), dim3(256), 0, ctx->stream, c);
  swa_t1(ctx, 14);
  SWA_HIP(ctx, hipGetLastError());
  return SWA_OK;
}

static int launch_csr_stream(swa_ctx * ctx, uint32_t first, uint32_t count, uint32_t nseg, uint64_t link_cap, uint64_t * d_offsets,
                             uint32_t * d_neighbours, uint64_t cap) {
  const uint64_t tiles_per_seg = (ctx->seg_cap + 4096 - 1) / 4096;
  return csr_from_chunks(ctx, first, count, static_cast<const unsigned long long *>(ctx->d_edges.ptr), nullptr, ctx->seg_cap,
                         static_cast<const uint32_t *>(ctx->d_seg_fill.ptr), nseg, (uint32_t)std::min<uint64_t>(ctx->seg_cap, 0xFFFFFFFFu),
                         (uint64_t)nseg * tiles_per_seg + 2, link_cap, d_offsets, d_neighbours, cap);
}

// the pair kernels by (pass, record width W in words, window width NW in words): W = 5 / 8 / 15 / 21 (width_words), NW = 1 / 2
// — and 4 with W >= 15: 128-nt windows need sequences of 257 nt
#define SWA_PAIR_CASE(KERNEL, P, WW, NN) \
  if (pass == P && width == WW && nwin == NN) { hipLaunchKernelGGL((KERNEL<P, WW, NN>), dim3(grid), dim3(kThreads), 0, ctx->stream, a); return SWA_OK; }
#define SWA_PAIR_WIDTH(KERNEL, WW) \
  SWA_PAIR_CASE(KERNEL, 0, WW, 1) SWA_PAIR_CASE(KERNEL, 1, WW, 1) SWA_PAIR_CASE(KERNEL, 0, WW, 2) SWA_PAIR_CASE(KERNEL, 1, WW, 2)
#define SWA_PAIR_CASES(KERNEL) \
  SWA_PAIR_WIDTH(KERNEL, 5) SWA_PAIR_WIDTH(KERNEL, 8) SWA_PAIR_WIDTH(KERNEL, 15) SWA_PAIR_WIDTH(KERNEL, 21) \
  SWA_PAIR_CASE(KERNEL, 0, 15, 4) SWA_PAIR_CASE(KERNEL, 1, 15, 4) SWA_PAIR_CASE(KERNEL, 0, 21, 4) SWA_PAIR_CASE(KERNEL, 1, 21, 4)
// (k_d1_group_pairs keeps a record per thread in LDS: static up to W = 8, dynamic — opted in per launch: the attribute belongs
// to the function on a device — for W = 15 / 21)
#define SWA_PAIR_CASE_DYN(KERNEL, P, WW, NN) \
  if (pass == P && width == WW && nwin == NN) { \
    const int bytes = (int)PairLayout<WW>::kDynamicBytes; \
    constexpr uint32_t bit_ = 1u << (P + (WW == 21 ? 2 : 0) + (NN == 2 ? 4 : (NN == 4 ? 8 : 0))); \
    if (bytes != 0 && (ctx->pair_lds_opt_in & bit_) == 0u) {     /* (once per context: the call costs tens of microseconds now and then) */ \
      SWA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL<P, WW, NN>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); \
      ctx->pair_lds_opt_in |= bit_; \
    } \
    hipLaunchKernelGGL((KERNEL<P, WW, NN>), dim3(grid), dim3(kThreads), (size_t)bytes, ctx->stream, a); return SWA_OK; }
#define SWA_PAIR_WIDTH_DYN(KERNEL, WW) \
  SWA_PAIR_CASE_DYN(KERNEL, 0, WW, 1) SWA_PAIR_CASE_DYN(KERNEL, 1, WW, 1) SWA_PAIR_CASE_DYN(KERNEL, 0, WW, 2) SWA_PAIR_CASE_DYN(KERNEL, 1, WW, 2)
static int launch_group_pairs(swa_ctx * ctx, int pass, int width, int nwin, int grid, const AnchorArgs & a) {
  SWA_PAIR_WIDTH_DYN(k_d1_group_pairs, 5) SWA_PAIR_WIDTH_DYN(k_d1_group_pairs, 8) SWA_PAIR_WIDTH_DYN(k_d1_group_pairs, 15) SWA_PAIR_WIDTH_DYN(k_d1_group_pairs, 21)
  SWA_PAIR_CASE_DYN(k_d1_group_pairs, 0, 15, 4) SWA_PAIR_CASE_DYN(k_d1_group_pairs, 1, 15, 4) SWA_PAIR_CASE_DYN(k_d1_group_pairs, 0, 21, 4) SWA_PAIR_CASE_DYN(k_d1_group_pairs, 1, 21, 4)
  return swa_fail_msg(ctx, SWA_E_ARG, "pair kernels: no kernel for this record / window width");
}
static int launch_pairs_tiled(swa_ctx * ctx, int pass, int width, int nwin, int grid, const AnchorArgs & a) {
  SWA_PAIR_CASES(k_d1_pairs_tiled)
  return swa_fail_msg(ctx, SWA_E_ARG, "pair kernels: no kernel for this record / window width");
}

// workgroups of k_d1_group_pairs<*, W, *> that one CU holds at a time (registers and LDS: the occupancy API; 1..8)
static int pair_blocks_per_cu(swa_ctx * ctx, uint32_t cls) {
  int & c = ctx->pair_blocks[cls];                        // (per context: swa_multi runs one context per GPU on its own host thread)
  if (c == 0) {
    int nb = 0;
    const hipError_t e = cls == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_d1_group_pairs<0, 5, 1>, kThreads, 0)
                       : cls == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_d1_group_pairs<0, 8, 1>, kThreads, 0)
                       : cls == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_d1_group_pairs<0, 15, 1>, kThreads, PairLayout<15>::kDynamicBytes)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_d1_group_pairs<0, 21, 1>, kThreads, PairLayout<21>::kDynamicBytes);
    c = (e == hipSuccess && nb >= 1) ? std::min(nb, 8) : (cls <= 1u ? 4 : (cls == 2u ? 2 : 1));
  }
  return c;
}

// per-wave link segments: one per wave of the largest launch (d_seg_fill: [fills | members staged, pass 0 | pass 1])
static uint32_t seg_count(const swa_ctx * ctx) { return (uint32_t)ctx->num_cus * 8u * kWaves; }

// anchored network over [first, first+count): per pass and width class the pair kernels over that class's work lists,
// then the seeds (or halves of seeds) left to the plain kernel; links in the per-wave segments
static int launch_network_anchored(swa_ctx * ctx, int ncb, uint32_t first, uint32_t count, bool count_links) {
  // d_acounters: [2] fallback seeds [8 + 2 (4 pass + class) + {0, 1}] work counters of the 65..256 groups / of the tiled kernel
  // [kCounterBase + 8 (4 index + class) + kind] the lists of k_group_lists (made at index build, not cleared here)
  auto * acounters = static_cast<uint32_t *>(ctx->d_acounters.ptr);
  SWA_TRY(swa_reserve(ctx, ctx->d_afallback, (2ull * count + 16) * sizeof(swa_fallback)));
  ClearList clears{};
  clear_add(clears, ctx->d_stats.ptr, 16 * sizeof(uint64_t));
  clear_add(clears, static_cast<uint64_t *>(ctx->d_guard.ptr) + 8, 8 * sizeof(uint64_t));   // the guard's counters of this network call
  if (count_links) { clear_add(clears, ctx->d_counts.ptr, uint64_t(count) * sizeof(uint32_t)); }
  clear_add(clears, acounters, kCounterBase * sizeof(uint32_t));
  clear_add(clears, ctx->d_seg_fill.ptr, 3ull * seg_count(ctx) * sizeof(uint32_t));   // segment fills | members staged, per pass
  // work counters of k_d1_group_pairs (per pass and class): 2^shard_bits of them, sched_stride entries apart
  uint32_t pair_batch = 0, shard_bits = 6, sched_stride = 64;   // (pair_batch 0: by the number of bundles, below)
  if (const char * e = getenv("SWA_D1_PAIR_BATCH")) { pair_batch = (uint32_t)std::max(1, atoi(e)); }                           // (experiments)
  if (const char * e = getenv("SWA_D1_PAIR_SHARD_BITS")) { shard_bits = (uint32_t)std::min(10, std::max(0, atoi(e))); }
  if (const char * e = getenv("SWA_D1_SCHED_STRIDE")) { sched_stride = (uint32_t)std::min(4096, std::max(1, atoi(e))); }
  // (a workgroup serves the bundles of shard blockIdx.x mod 2^shard_bits: every shard needs a workgroup — ADVICE r03)
  while (shard_bits > 0 && (1u << shard_bits) > (uint32_t)ctx->num_cus) { --shard_bits; }
  {
    const uint64_t bytes = ((2ull * kWidthClasses) << shard_bits) * sched_stride * sizeof(uint32_t);
    SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbSched], bytes));
    clear_add(clears, ctx->d_stream[kSbSched].ptr, bytes);
  }
  SWA_TRY(clear_launch(ctx, clears));
  const bool zlds = 4ull * ctx->zobrist_len * sizeof(uint64_t) <= kMaxZobristLds;
  const uint32_t maxwords = (ctx->db.longest + 31u) >> 5;
  auto * stats = static_cast<unsigned long long *>(ctx->d_stats.ptr);
  swa_t0(ctx, 3);
  const bool window_mode = ctx->anchor_a != 0 || ctx->anchor_b != 0;
  uint64_t item_room = 0;
  const ListRegions regions = list_regions(ctx, &item_room);
  if (item_room != ctx->list_regions_items) { return swa_fail_msg(ctx, SWA_E_ARG, "d=1 network: the work lists in place were laid out for another database"); }
  const int nwin = (int)(ctx->anchor_w / 32u);
  for (int pass = 0; pass < 2; ++pass) {
    swa_t0(ctx, 11 + pass);
    AnchorArgs a{};
    a.pass = pass;
    a.no_cluster_breaking = ncb;
    a.first = first; a.count = count;
    a.edges = static_cast<uint64_t *>(ctx->d_edges.ptr);
    a.seg_cap = ctx->seg_cap;
    a.seg_fill = static_cast<uint32_t *>(ctx->d_seg_fill.ptr);
    a.counts = count_links ? static_cast<uint32_t *>(ctx->d_counts.ptr) : nullptr;
    a.minlen = anchor_minlen(ctx);
    a.win_word = ctx->anchor_a / 32u;
    a.win_word_b = ctx->anchor_b / 32u;
    a.window_mode = window_mode ? 1u : 0u;
    a.seg_staged = static_cast<uint32_t *>(ctx->d_seg_fill.ptr) + (uint64_t)(1 + pass) * seg_count(ctx);
    a.guard = static_cast<unsigned long long *>(ctx->d_guard.ptr);
    a.flags = static_cast<uint32_t *>(ctx->d_flags.ptr);
    a.batch = pair_batch; a.shard_bits = shard_bits; a.sched_stride = sched_stride;
    a.ids = static_cast<const uint32_t *>(ctx->d_stream[kSbMembers + pass].ptr);
    a.lines = static_cast<const uint4 *>(ctx->d_stream[kSbLines].ptr);
    a.line_quads = ctx->lines_quads;
    a.pair_items = static_cast<const swa_item *>(ctx->d_aitems[pass].ptr);
    for (uint32_t cls = 0; cls < kWidthClasses; ++cls) {
      // (a group of class c has a member of class c: no such sequences, no such groups.  Lines too narrow for the class's
      // records cannot occur: the lines are as wide as the longest sequence needs)
      if (ctx->class_pop[cls] == 0) { continue; }
      const int width = width_words(cls);
      if (nwin == 4 && width < 15) { return swa_fail_msg(ctx, SWA_E_ARG, "pair kernels: 128-nt windows with sequences of 256 nt or less"); }
      const uint32_t pc = (uint32_t)pass * kWidthClasses + cls;
      for (uint32_t k = 0; k <= kPairClasses; ++k) { a.pair_region[k] = regions.at[cls][k]; }
      a.pair_counters = acounters + kCounterBase + pc * 8u;
      a.items = a.pair_items + regions.at[cls][kPairClasses + 1u];
      a.item_count = a.pair_counters + kPairClasses + 1u;
      a.sched_big = acounters + 8 + 2 * pc;
      a.sched_tiled = acounters + 9 + 2 * pc;
      a.sched_wide = static_cast<uint32_t *>(ctx->d_stream[kSbSched].ptr) + ((uint64_t)pc << shard_bits) * sched_stride;
      // (the pair kernel hands its work out through counters: exactly the workgroups that are resident together, no second round;
      // no launch over lists the index build found empty)
      const uint32_t * have = ctx->list_counts_ready ? ctx->list_counts + pc * 8u : nullptr;
      bool any_pairs = have == nullptr;
      for (uint32_t k = 0; k <= kPairClasses && have != nullptr; ++k) { any_pairs = any_pairs || have[k] != 0u; }
      // Bundles per visit of a work counter: 4 where every wave has dozens of bundles to go through (10 M amplicons: the counters
      // are what limits then — 0.78 / 1.21 / 2.2 ms with 4 / 2 / 1), fewer where the waves' shares are what limits (1 M: 2.4
      // bundles a wave — 0.072 / 0.049 / 0.037 ms a pass with 4 / 2 / 1)
      const int pgrid = ctx->num_cus * pair_blocks_per_cu(ctx, cls);
      a.batch = pair_batch;
      if (pair_batch == 0) {
        uint64_t bundles = 0;
        for (uint32_t k = 0; k < kPairClasses && have != nullptr; ++k) { bundles += ((uint64_t)have[k] + (16u >> k) - 1u) >> (4u - k); }
        const uint64_t waves = (uint64_t)pgrid * 4u;
        a.batch = have == nullptr || bundles >= 16u * waves ? 4u : (bundles >= 8u * waves ? 2u : 1u);
      }
      if (any_pairs) { SWA_TRY(launch_group_pairs(ctx, pass, width, nwin, pgrid, a)); }
      if (have == nullptr || have[kPairClasses + 1u] != 0u) { SWA_TRY(launch_pairs_tiled(ctx, pass, width, nwin, ctx->num_cus * (cls <= 1u ? 8 : 4), a)); }
    }
    swa_t1(ctx, 11 + pass);
    SWA_HIP(ctx, hipGetLastError());
  }
  // seeds (or halves of seeds) the anchored passes skipped (a lean build has made sure there are none: no sequence
  // too short, no oversized group — and what holds for the whole database holds for every part of it)
  const bool use_member_table = ctx->member_index && !ctx->full_index;
  if (ctx->full_index || ctx->member_index) {
    hipLaunchKernelGGL(k_stream_fallback, dim3(grid_for(ctx, count, 256, 8)), dim3(256), 0, ctx->stream, ctx->db.seqlen, first, count,
                       static_cast<const uint8_t *>(ctx->d_stream[kSbOver].ptr), static_cast<swa_fallback *>(ctx->d_afallback.ptr), acounters + 2,
                       ctx->owner_rank, ctx->owner_world, ctx->db.seqs, ctx->db.seq_off, anchor_minlen(ctx));
  }
  NetArgs f{};
  f.seqs = ctx->db.seqs; f.seq_off = ctx->db.seq_off; f.seqlen = ctx->db.seqlen; f.abundance = ctx->db.abundance;
  f.zobrist = static_cast<const uint64_t *>(ctx->d_zobrist.ptr);
  f.zlen = ctx->zobrist_len; f.maxwords = maxwords;
  f.table = static_cast<const swa_slot *>(use_member_table ? ctx->d_stream[kSbMTable].ptr : ctx->d_table.ptr);
  f.tmask = (use_member_table ? ctx->mtable_size : ctx->table_size) - 1;
  f.bloom = static_cast<const uint64_t *>(use_member_table ? ctx->d_stream[kSbMBloom].ptr : ctx->d_bloom.ptr);
  f.bmask = (use_member_table ? ctx->mbloom_words : ctx->bloom_words) - 1;
  f.patterns = static_cast<const uint64_t *>(ctx->d_patterns.ptr);
  f.no_cluster_breaking = ncb;
  f.first = first; f.count = count;
  f.edges = static_cast<uint64_t *>(ctx->d_edges.ptr);
  f.seg_cap = ctx->seg_cap;
  f.seg_fill = static_cast<uint32_t *>(ctx->d_seg_fill.ptr);
  f.stats = stats;
  f.counts = count_links ? static_cast<uint32_t *>(ctx->d_counts.ptr) : nullptr;
  f.fallback = static_cast<const swa_fallback *>(ctx->d_afallback.ptr);
  f.fallback_count = acounters + 2;
  f.aux = static_cast<const swa_aux *>(ctx->d_aux.ptr);
  f.owner_rank = 0; f.owner_world = 1;                       // the list already is this rank's share
  f.window_mode = window_mode ? 1u : 0u; f.win_word = ctx->anchor_a / 32u; f.anchor_w = ctx->anchor_w;
  const size_t flds = sizeof(uint64_t) * ((zlds ? 4ull * ctx->zobrist_len : 0ull) + 1024ull +
                                          kWaves * ((size_t)(maxwords + 2u) + kQueueCap + kQueueCap / 2));
  const int fgrid = grid_for(ctx, count, kWaves, 8);
  // (a rank that built only its owned groups has no table: by construction its fallback list is
  // empty — network_run checks the count and builds the full index if that ever fails to hold)
  if (!ctx->full_index && !ctx->member_index) { /* nothing to probe against */ }
  else if (zlds) { hipLaunchKernelGGL((k_d1_probe<true, false, 2>), dim3(fgrid), dim3(kThreads), flds, ctx->stream, f); }
  else { hipLaunchKernelGGL((k_d1_probe<false, false, 2>), dim3(fgrid), dim3(kThreads), flds, ctx->stream, f); }
  swa_t1(ctx, 3);
  SWA_HIP(ctx, hipGetLastError());
  return SWA_OK;
}

// Zobrist table + per-amplicon sequence hashes (src/db.cc:761, src/zobrist.cc:89-112) and the
// anchored-index metadata; shared by the d = 1 index and the d = 0 dereplication
// Zobrist table in HBM (it only depends on its length: uploaded once) + room for the hashes
static int prepare_hashing(swa_ctx * ctx) {
  const uint32_t n = ctx->db.n;
  ctx->zobrist_len = ctx->db.longest + 2;                  // db.cc:652-653 (sequence part)
  SWA_TRY(swa_reserve(ctx, ctx->d_seqhash, uint64_t(n) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_aux, uint64_t(n) * sizeof(swa_aux)));
  if (ctx->zobrist_resident != ctx->zobrist_len) {
    std::vector<uint64_t> zob;
    swa_zobrist_table(ctx->zobrist_len, zob);
    SWA_TRY(swa_reserve(ctx, ctx->d_zobrist, zob.size() * sizeof(uint64_t)));
    SWA_HIP(ctx, hipMemcpyAsync(ctx->d_zobrist.ptr, zob.data(), zob.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));      // zob is a host temporary
    ctx->zobrist_resident = ctx->zobrist_len;
  }
  return SWA_OK;
}

// sequence hashes + the XOR streams the plain kernel's restricted enumeration reads (swa_aux)
static int launch_seqhash(swa_ctx * ctx, const uint8_t * only = nullptr) {
  const uint32_t n = ctx->db.n;
  const size_t zbytes = 4ull * ctx->zobrist_len * sizeof(uint64_t);
  const bool zlds = zbytes <= kMaxZobristLds;
  swa_t0(ctx, 0);
  const int hgrid = grid_for(ctx, n, 256, 8);
  if (zlds) {
    hipLaunchKernelGGL(k_seqhash<true>, dim3(hgrid), dim3(256), zbytes, ctx->stream, ctx->db.seqs, ctx->db.seq_off,
                       ctx->db.seqlen, static_cast<const uint64_t *>(ctx->d_zobrist.ptr), ctx->zobrist_len, n,
                       static_cast<uint64_t *>(ctx->d_seqhash.ptr), static_cast<swa_aux *>(ctx->d_aux.ptr), ctx->anchor_w, only);
  } else {
    hipLaunchKernelGGL(k_seqhash<false>, dim3(hgrid), dim3(256), 0, ctx->stream, ctx->db.seqs, ctx->db.seq_off,
                       ctx->db.seqlen, static_cast<const uint64_t *>(ctx->d_zobrist.ptr), ctx->zobrist_len, n,
                       static_cast<uint64_t *>(ctx->d_seqhash.ptr), static_cast<swa_aux *>(ctx->d_aux.ptr), ctx->anchor_w, only);
  }
  SWA_HIP(ctx, hipGetLastError());
  swa_t1(ctx, 0);
  return SWA_OK;
}

// Zobrist table + per-amplicon sequence hashes (src/db.cc:761, src/zobrist.cc:89-112) and the
// anchored-index metadata; shared by the d = 1 index and the d = 0 dereplication
int swa_hash_sequences(swa_ctx * ctx) {
  SWA_TRY(prepare_hashing(ctx));
  return launch_seqhash(ctx);
}

extern "C" int swa_d1_index_build(swa_ctx * ctx, int * has_duplicates) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  return swa_d1_index_build_range(ctx, 0, ctx->db.n, has_duplicates);
}

// hashes of ALL amplicons + the database-wide table and Bloom filter (src/algod1.cc:188-208,
// src/bloompat.cc): what the plain kernel, the fastidious pass and the debug readers work on.
// Always built by a single-GPU index build; a rank that serves only the anchor groups it owns
// builds it on demand (rarely: see build_owned_index).
static int ensure_full_index(swa_ctx * ctx) {
  if (ctx->full_index) { return SWA_OK; }
  SWA_TRY(swa_reserve(ctx, ctx->d_table, ctx->table_size * sizeof(swa_slot)));
  SWA_TRY(swa_reserve(ctx, ctx->d_bloom, ctx->bloom_words * sizeof(uint64_t)));
  SWA_TRY(swa_hash_sequences(ctx));
  swa_t0(ctx, 1);
  SWA_TRY(swa_d1_rebuild_table(ctx, nullptr));
  swa_t1(ctx, 1);
  ctx->full_index = true;
  return SWA_OK;
}

// Anchor windows for this database, from a sample (k_anchor_sample): the smallest offset whose estimated share of
// amplicons in oversized groups and of too-short seeds is below 1 / 64 each; (0, 0) when nothing is skewed, which is
// the normal case.  Window mode needs the pair kernels (sequences up to 416 nt); SWA_D1_WINDOWS=0 switches it off.
static int choose_anchor_windows(swa_ctx * ctx) {
  ctx->anchor_a = ctx->anchor_b = 0;
  ctx->anchor_w = 32;
  SWA_TRY(ensure_db_lengths(ctx));
  const uint32_t shortest = ctx->db_shortest;
  const char * env_win = getenv("SWA_D1_WINDOWS");
  const uint32_t n = ctx->db.n;
  const uint32_t max_nwin = anchor_max_nwin(ctx);
  // the ends, as wide as the shortest sequence allows — unless the sample finds them skewed (conserved flanks), then 32-nt
  // windows moved inwards
  ctx->anchor_w = 32u * anchor_nwin_for(shortest, 0u, max_nwin);
  if (env_win != nullptr && env_win[0] == '0') { return SWA_OK; }
  const uint32_t stride = std::max<uint32_t>(1u, n / 65536u);
  const uint32_t samples = (n + stride - 1) / stride;
  const size_t slots = (size_t)2 * kSampleCandidates * kSampleSlots;
  SWA_TRY(swa_reserve(ctx, ctx->d_akeys[0], std::max<size_t>(ctx->d_akeys[0].bytes, slots * sizeof(uint64_t))));
  SWA_TRY(swa_reserve(ctx, ctx->d_acounts[0], std::max<size_t>(ctx->d_acounts[0].bytes, (slots + 16) * sizeof(uint32_t))));
  auto * keys = static_cast<unsigned long long *>(ctx->d_akeys[0].ptr);
  auto * counts = static_cast<uint32_t *>(ctx->d_acounts[0].ptr);
  uint32_t * stats = counts + slots;                         // [0..4) too short, [4..8) mass
  SWA_HIP(ctx, hipMemsetAsync(keys, 0xFF, slots * sizeof(uint64_t), ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(counts, 0, (slots + 16) * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL(k_anchor_sample, dim3(grid_for(ctx, samples, 256, 8)), dim3(256), 0, ctx->stream, ctx->db.seqs, ctx->db.seq_off,
                     ctx->db.seqlen, n, stride, keys, counts, stats, shortest, max_nwin);
  hipLaunchKernelGGL(k_sample_mass, dim3(grid_for(ctx, slots, 256, 8)), dim3(256), 0, ctx->stream, counts, stride, stats + 4);
  uint32_t host[8] = {};
  SWA_HIP(ctx, hipMemcpyAsync(host, stats, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t c = 0; c < kSampleCandidates; ++c) {
    const bool few_short = host[c] <= samples / 64u || c == 0;
    if (host[4 + c] <= samples / 64u && few_short) {
      ctx->anchor_a = ctx->anchor_b = 32u * c;
      ctx->anchor_w = 32u * anchor_nwin_for(shortest, 32u * c, max_nwin);
      return SWA_OK;
    }
    if (!few_short) { break; }                               // (larger offsets only strand more seeds)
  }
  return SWA_OK;                                             // nothing qualifies: the ends it is (and the plain kernel for the giants)
}

// the anchor windows of this database: chosen from a sample once per upload (and corrected by the safety net of the
// first build if the sample misjudged), then reused
static int ensure_anchor_windows(swa_ctx * ctx) {
  if (!ctx->windows_ready) {
    SWA_TRY(choose_anchor_windows(ctx));
    ctx->windows_chosen = ctx->anchor_a;
    ctx->windows_w = ctx->anchor_w;
    ctx->windows_ready = true;
  }
  ctx->anchor_a = ctx->anchor_b = ctx->windows_chosen;
  ctx->anchor_w = ctx->windows_w;
  return SWA_OK;
}

// Hash table + Bloom filter of the members of oversized groups ONLY (the streaming build marks them: one byte per
// amplicon), for the plain kernel that serves their halves: a neighbour that shares the window of an oversized group is
// a member of that group, so nothing else needs to be in the table — and a Bloom filter of a few megabytes stays in L2
// where the database-wide one (one byte per table slot: 33 MB at 10 M) is a random HBM / Infinity-Cache line per probe.
// Identical sequences among the members are found through the same table (k_dup_check), as the reference finds them
// while it inserts (src/algod1.cc:1131-1150).  SWA_D1_MEMBER_TABLE=0: the database-wide structures instead.
static bool member_index_enabled() {
  const char * e = getenv("SWA_D1_MEMBER_TABLE");
  return !(e != nullptr && e[0] == '0');
}

static int build_member_index(swa_ctx * ctx, uint32_t dup_first, uint32_t dup_count) {
  ctx->member_index = false;
  const uint32_t n = ctx->db.n;
  // Zobrist hashes + the XOR streams of the members (k_seqhash: nobody else is inserted, probed for or enumerated)
  SWA_TRY(prepare_hashing(ctx));
  SWA_TRY(launch_seqhash(ctx, static_cast<const uint8_t *>(ctx->d_stream[kSbOver].ptr)));
  uint64_t slots = 1024;
  while (slots < 2ull * ctx->over_mass) { slots <<= 1; }
  const uint64_t words = std::max<uint64_t>(slots >> 3, 1);  // one byte of filter per slot, as the reference sizes it
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbMTable], slots * sizeof(swa_slot)));
  SWA_TRY(swa_reserve(ctx, ctx->d_stream[kSbMBloom], words * sizeof(uint64_t)));
  auto * table = static_cast<swa_slot *>(ctx->d_stream[kSbMTable].ptr);
  auto * bloom = static_cast<uint64_t *>(ctx->d_stream[kSbMBloom].ptr);
  swa_t0(ctx, 1);
  hipLaunchKernelGGL(k_table_clear, dim3(grid_for(ctx, slots, 256, 8)), dim3(256), 0, ctx->stream, table, slots, bloom, words);
  hipLaunchKernelGGL(k_table_insert, dim3(grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, static_cast<const uint64_t *>(ctx->d_seqhash.ptr), n,
                     static_cast<const uint8_t *>(ctx->d_stream[kSbOver].ptr), table, slots - 1, reinterpret_cast<unsigned long long *>(bloom), words - 1,
                     static_cast<const uint64_t *>(ctx->d_patterns.ptr));
  swa_t1(ctx, 1);
  swa_t0(ctx, 2);
  if (dup_count != 0) {
    hipLaunchKernelGGL(k_dup_check, dim3(grid_for(ctx, dup_count, 256, 8)), dim3(256), 0, ctx->stream, ctx->db.seqs, ctx->db.seq_off, ctx->db.seqlen,
                       static_cast<const uint64_t *>(ctx->d_seqhash.ptr), dup_first, dup_count, table, slots - 1, static_cast<uint32_t *>(ctx->d_flags.ptr));
  }
  swa_t1(ctx, 2);
  SWA_HIP(ctx, hipGetLastError());
  ctx->mtable_size = slots;
  ctx->mbloom_words = words;
  ctx->member_index = true;
  return SWA_OK;
}

// Index build of a rank that serves only the anchor groups it owns (swa_d1_set_ownership,
// world > 1), without anything proportional to the database except streaming passes: abundance
// ranks, the anchor indexes of the owned groups (all their members), hashes and XOR streams of
// those members only, duplicates inside the owned prefix groups.  *needs_table is set when some
// seed can only be served by the plain kernel (a sequence shorter than 65 nt anywhere, a group
// too large for LDS) or the db order does not hold: the caller then builds the full index.
static int build_owned_index(swa_ctx * ctx, uint32_t first, uint32_t count, bool * needs_table, uint32_t * dups, uint32_t * oversized_mass = nullptr,
                             uint32_t * shortest = nullptr) {
  const uint32_t n = ctx->db.n;
  auto * dflags = static_cast<uint32_t *>(ctx->d_flags.ptr);
  swa_lap(ctx, "(owned index starts)");
  // (no Zobrist table, no hash / XOR-stream arrays here: the pair route needs none of them — 240 MB of HBM at 10 M and an
  // upload with its synchronisation per build; the member table and the database-wide table prepare them when they are built)
  SWA_TRY(launch_abundance_rank(ctx));
  swa_lap(ctx, "abundance ranks");
  // a rank of a multi-GPU job answers for the groups it owns, whichever slice their members lie in (each pair of
  // identical sequences is seen by exactly one rank); a single GPU honours the slice it was asked about
  const uint32_t dup_first = ctx->owner_world > 1 ? 0u : first, dup_count = ctx->owner_world > 1 ? n : count;
  // streaming build (d1_stream.inc): keys, partition, groups + work lists + identical sequences, all in one go
  swa_t0(ctx, 7);
  SWA_TRY(build_stream_index(ctx, dup_first, dup_count));
  swa_t1(ctx, 7);
  // [0] duplicates [1] order broken [2] a bucket's groups do not fit the table [3] a sequence too short for two windows
  // [4] groups left to the plain kernel (oversized / a member too long) [5] their members [6] 0xFFFFFFFF - shortest sequence
  // (ONE copy of the status block: the flags, and how long the work lists are — the network call then launches no kernel
  // over an empty one)
  // (into the context's pinned mirror: a copy to pageable memory is staged by the runtime, ~15 us more per look)
  auto * status = static_cast<uint32_t *>(ctx->h_status);
  SWA_HIP(ctx, hipMemcpyAsync(status, ctx->d_status.ptr, 512 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t * flags = status;
  *dups = flags[0];
  memcpy(ctx->list_counts, status + 256 + kCounterBase, sizeof(ctx->list_counts));
  ctx->list_counts_ready = true;
  if (oversized_mass != nullptr) { *oversized_mass = flags[5]; }
  if (shortest != nullptr) { *shortest = 0xFFFFFFFFu - flags[6]; }
  if (flags[2] != 0) {                                      // a bucket with more distinct keys than the group kernel's table: finer
    if (ctx->stream_extra_bits >= 8) { return swa_fail_msg(ctx, SWA_E_DEVICE, "streaming index build: partition still too coarse"); }
    ctx->stream_extra_bits += 2;
    SWA_HIP(ctx, hipMemsetAsync(dflags, 0, 16 * sizeof(uint32_t), ctx->stream));
    return build_owned_index(ctx, first, count, needs_table, dups, oversized_mass, shortest);
  }
  if (flags[1] != 0) { ctx->db_unordered = true; }
  *needs_table = flags[1] != 0 || flags[3] != 0 || flags[4] != 0;
  // (groups left to the plain kernel — and, on a single GPU, sequences too short for two windows — are the ONLY reason: a
  // table of their members alone serves it — build_member_index)
  ctx->only_oversized = flags[1] == 0 && (flags[4] != 0 || flags[3] != 0) &&
                        (flags[3] == 0 || (ctx->owner_world == 1u && ctx->route_ids[0] == nullptr && ctx->route_rec[0] == nullptr && member_index_enabled()));
  ctx->over_mass = flags[5];
  return SWA_OK;
}

extern "C" int swa_d1_index_build_range(swa_ctx * ctx, uint32_t first, uint32_t count, int * has_duplicates) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (ctx->db.n == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_index_build: no database"); }
  if (first > ctx->db.n || count > ctx->db.n - first) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_index_build_range: bad range"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->db.n;
  ctx->index_first = first; ctx->index_count = count; ctx->index_routed = ctx->route_ids[0] != nullptr || ctx->route_rec[0] != nullptr;   // (network_run_guarded repeats it)
  ctx->d1_ready = false;
  ctx->csr_ready = false;                                   // (a resident network belongs to the index it was made from)
  ctx->anchor_ready = false;
  ctx->full_index = false;
  ctx->member_index = false;
  ctx->anchor_a = ctx->anchor_b = 0;
  for (int slot : {0, 1, 2, 7, 8, 9, 10, 15}) { ctx->ev_used[slot] = false; }   // phases this build does not run report 0
  ctx->table_size = swa_hashtable_size(n);
  const uint64_t bloom_bytes = ctx->table_size < 8 ? 8 : ctx->table_size;   // bloompat.cc:100-113
  ctx->bloom_words = bloom_bytes >> 3;

  if (!ctx->patterns_resident) {                            // the 1024 Bloom patterns are constants: upload once
    std::vector<uint64_t> pat;
    swa_bloom_patterns(1024, 8, pat);
    SWA_TRY(swa_reserve(ctx, ctx->d_patterns, pat.size() * sizeof(uint64_t)));
    SWA_HIP(ctx, hipMemcpyAsync(ctx->d_patterns.ptr, pat.data(), pat.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));       // pat is a host temporary
    ctx->patterns_resident = true;
  }
  SWA_TRY(swa_reserve(ctx, ctx->d_flags, 16 * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_stats, 16 * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_guard, 24 * sizeof(uint64_t)));
  // (flags, stats and the guard's index / network counters lie together in the status block: one fill)
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_status.ptr, 0, 64 + 128 + 16 * sizeof(uint64_t), ctx->stream));
  ctx->guard_index = false;
  ctx->anchor_w = 32;
  ctx->anchor_usable = anchor_applicable(ctx);

  bool owned_ok = false;                                    // served without the database-wide table
  uint32_t group_dups = 0;
  // The lean build (no database-wide table / Bloom / table-based duplicate check) serves a single GPU as well
  // as a rank of a multi-GPU job: with world = 1 this GPU owns every anchor group.
  if (ctx->anchor_usable && owned_index_enabled()) {
    bool needs_table = false;
    uint32_t mass = 0, shortest = 0;
    swa_lap(ctx, "(index build: before the windows)");
    SWA_TRY(ensure_anchor_windows(ctx));
    swa_lap(ctx, "anchor windows");
    const uint32_t sampled = ctx->anchor_a;
    const bool routed = ctx->route_ids[0] != nullptr || ctx->route_rec[0] != nullptr;        // (the lists were made under these windows: no second thoughts)
    SWA_TRY(build_owned_index(ctx, first, count, &needs_table, &group_dups, &mass, &shortest));
    // Conserved flanks: when a noticeable part of the database sits in groups too large for LDS (everybody shares
    // the first or last 32 nt), the anchor windows move inwards, 32 nt at a time, as far as the shortest sequence
    // allows (every seed needs win_a + win_b + 65 nt), and the setting with the fewest stranded members wins.  Window
    // SWA_D1_WINDOWS=0 switches the search off.
    const char * env_win = getenv("SWA_D1_WINDOWS");
    // (safety net behind the sample: the real build still found too many stranded members — try the next offsets.
    // Single GPU only: under ownership `mass` counts the oversized groups THIS rank owns, the ranks would settle on
    // different windows and divide the pairs differently; there the sampled choice — the same on every rank — stands
    // and oversized groups take the plain kernel)
    if (!routed && ctx->owner_world == 1 && needs_table && mass > n / 64u && !(env_win != nullptr && env_win[0] == '0')) {
      uint32_t best = sampled, best_mass = mass;
      for (uint32_t w = sampled + 32u; 2u * w + kMinAnchoredLen <= shortest && w <= 96u; w += 32u) {
        ctx->anchor_a = ctx->anchor_b = w;
        ctx->anchor_w = 32;                                     // (moved windows: anchor_nwin_for)
        SWA_HIP(ctx, hipMemsetAsync(ctx->d_flags.ptr, 0, 16 * sizeof(uint32_t), ctx->stream));
        uint32_t m = 0;
        SWA_TRY(build_owned_index(ctx, first, count, &needs_table, &group_dups, &m));
        if (m < best_mass) { best_mass = m; best = w; }
        if (m <= n / 64u) { break; }
      }
      if (ctx->anchor_a != best) {                            // (the last one tried was not the best: build that one again)
        ctx->anchor_a = ctx->anchor_b = best;
        ctx->anchor_w = best == sampled ? ctx->windows_w : 32u;
        SWA_HIP(ctx, hipMemsetAsync(ctx->d_flags.ptr, 0, 16 * sizeof(uint32_t), ctx->stream));
        SWA_TRY(build_owned_index(ctx, first, count, &needs_table, &group_dups));
      }
    }
    ctx->windows_chosen = ctx->anchor_a;                     // (what the safety net settled on, for the next build)
    ctx->windows_w = ctx->anchor_w;
    owned_ok = !needs_table;
    if (!owned_ok && ctx->only_oversized && member_index_enabled()) {
      // nothing but groups too large for the pair kernels stands in the way: a table of their members is all the plain kernel needs
      SWA_TRY(build_member_index(ctx, ctx->owner_world > 1 ? 0u : first, ctx->owner_world > 1 ? n : count));
      owned_ok = true;
      // (the member table's own search for identical sequences may have raised the flag since the build looked at it)
      auto * mirror = static_cast<uint32_t *>(ctx->h_status);
      SWA_HIP(ctx, hipMemcpyAsync(mirror, ctx->d_flags.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
      SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      group_dups = mirror[0];
    }
    if (!owned_ok) { SWA_HIP(ctx, hipMemsetAsync(ctx->d_flags.ptr, 0, 16 * sizeof(uint32_t), ctx->stream)); }
  }
  uint32_t flag = group_dups;
  if (!owned_ok) {
    SWA_TRY(ensure_full_index(ctx));
    swa_t0(ctx, 2);
    if (count > 0) {
      hipLaunchKernelGGL(k_dup_check, dim3(grid_for(ctx, count, 256, 8)), dim3(256), 0, ctx->stream, ctx->db.seqs, ctx->db.seq_off,
                         ctx->db.seqlen, static_cast<const uint64_t *>(ctx->d_seqhash.ptr), first, count,
                         static_cast<const swa_slot *>(ctx->d_table.ptr), ctx->table_size - 1,
                         static_cast<uint32_t *>(ctx->d_flags.ptr));
    }
    SWA_HIP(ctx, hipGetLastError());
    swa_t1(ctx, 2);
    if (ctx->anchor_usable) {
      SWA_TRY(launch_abundance_rank(ctx));
      SWA_HIP(ctx, hipGetLastError());
    }
    uint32_t flags[2] = {0, 0};                             // [0] duplicates [1] abundances not in descending order
    SWA_HIP(ctx, hipMemcpyAsync(flags, ctx->d_flags.ptr, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    flag |= flags[0];
    if (flags[1] != 0) { ctx->anchor_usable = false; ctx->db_unordered = true; }   // the anchored passes rely on the db order
  }
  ctx->d1_ready = true;
  if (has_duplicates != nullptr) { *has_duplicates = flag != 0 ? 1 : 0; }
  if (flag != 0) {
    return swa_fail_msg(ctx, SWA_E_DUPLICATES, "some fasta entries have identical sequences");
  }
  return SWA_OK;
}

// width of the anchor windows the last index build chose, in nucleotides (32, 64 or 128)
extern "C" uint32_t swa_d1_anchor_width(const swa_ctx * ctx) { return ctx != nullptr ? ctx->anchor_w : 0; }

extern "C" uint64_t swa_d1_table_size(const swa_ctx * ctx) { return ctx != nullptr ? ctx->table_size : 0; }

// the anchor windows the last index build chose: out2 = {nt from the start, nt from the end} (0, 0 = the first / last 32 nt)
extern "C" int swa_d1_anchor_windows(const swa_ctx * ctx, uint32_t * out2) {
  if (ctx == nullptr || out2 == nullptr) { return SWA_E_ARG; }
  out2[0] = ctx->anchor_a; out2[1] = ctx->anchor_b;
  return SWA_OK;
}

static size_t network_lds_bytes(const swa_ctx * ctx, bool zlds) {
  const uint32_t maxwords = (ctx->db.longest + 31u) >> 5;
  const size_t per_wave = (size_t)(maxwords + 2u) + kQueueCap + kQueueCap / 2;
  return sizeof(uint64_t) * ((zlds ? 4ull * ctx->zobrist_len : 0ull) + 1024ull + kWaves * per_wave);
}

// launches the network kernel for [first, first+count); leaves the edge list, the
// per-amplicon counts and the edge counter on the device
static int launch_network(swa_ctx * ctx, int ncb, uint32_t first, uint32_t count, bool stats) {
  NetArgs a{};
  a.seqs = ctx->db.seqs; a.seq_off = ctx->db.seq_off; a.seqlen = ctx->db.seqlen; a.abundance = ctx->db.abundance;
  a.zobrist = static_cast<const uint64_t *>(ctx->d_zobrist.ptr);
  a.zlen = ctx->zobrist_len;
  a.maxwords = (ctx->db.longest + 31u) >> 5;
  a.table = static_cast<const swa_slot *>(ctx->d_table.ptr);
  a.tmask = ctx->table_size - 1;
  a.bloom = static_cast<const uint64_t *>(ctx->d_bloom.ptr);
  a.bmask = ctx->bloom_words - 1;
  a.patterns = static_cast<const uint64_t *>(ctx->d_patterns.ptr);
  a.no_cluster_breaking = ncb;
  a.first = first; a.count = count;
  a.edges = static_cast<uint64_t *>(ctx->d_edges.ptr);
  a.seg_cap = ctx->seg_cap;
  a.seg_fill = static_cast<uint32_t *>(ctx->d_seg_fill.ptr);
  a.stats = static_cast<unsigned long long *>(ctx->d_stats.ptr);
  a.counts = static_cast<uint32_t *>(ctx->d_counts.ptr);
  a.owner_rank = ctx->owner_rank; a.owner_world = ctx->owner_world;
  a.anchor_w = 32;
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_stats.ptr, 0, 16 * sizeof(uint64_t), ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_counts.ptr, 0, uint64_t(count) * sizeof(uint32_t), ctx->stream));   // skipped seeds keep 0
  const bool zlds = 4ull * ctx->zobrist_len * sizeof(uint64_t) <= kMaxZobristLds;
  const size_t lds = network_lds_bytes(ctx, zlds);
  const int grid = grid_for(ctx, count, kWaves, 8);
  swa_t0(ctx, 3);
  if (zlds && stats) {
    hipLaunchKernelGGL((k_d1_probe<true, true, 0>), dim3(grid), dim3(kThreads), lds, ctx->stream, a);
  } else if (zlds) {
    hipLaunchKernelGGL((k_d1_probe<true, false, 0>), dim3(grid), dim3(kThreads), lds, ctx->stream, a);
  } else if (stats) {
    hipLaunchKernelGGL((k_d1_probe<false, true, 0>), dim3(grid), dim3(kThreads), lds, ctx->stream, a);
  } else {
    hipLaunchKernelGGL((k_d1_probe<false, false, 0>), dim3(grid), dim3(kThreads), lds, ctx->stream, a);
  }
  swa_t1(ctx, 3);
  SWA_HIP(ctx, hipGetLastError());
  return SWA_OK;
}

// The guard (round 4).  The reference's network thread is serialised by a mutex and cannot return a partial
// network (src/algod1.cc:630-670); this pipeline is twenty kernels handing records to one another, and twice in three
// rounds a run lost the links of a few wavefronts' worth of amplicons without any error (DESIGN: the anomaly).  So the
// counts that must balance are kept and compared before a network leaves this file:
//   records keyed per index (k_keys)  =  members of listed + singleton + oversized groups (k_group1)
//   members of listed groups          =  members the pair kernels staged, per pass
//   no member sits in a group whose window key is not its own (pair_misfiled; key collisions excepted, exactly)
//   links in the wave segments        =  links the partition sorted  =  the last CSR offset
// A mismatch is SWA_E_INTERNAL — never a short network.
static int guard_check(swa_ctx * ctx, const uint64_t * g, const uint64_t * got, bool csr_stream, uint64_t csr_end, uint32_t links_sorted,
                       uint64_t n_edges) {
  char msg[256];
  if (g[10] != 0) {
    snprintf(msg, sizeof(msg), "d=1 guard: %llu member(s) filed under an anchor key that is not theirs", (unsigned long long)g[10]);
    return swa_fail_msg(ctx, SWA_E_INTERNAL, msg);
  }
  if (ctx->stream_index && ctx->guard_keys_pending) {
    ctx->guard_keys_pending = false;
    for (int i = 0; i < 2; ++i) {
      if (g[16 + i] != g[19 + i]) {
        snprintf(msg, sizeof(msg), "d=1 guard: the %s derived from the amplicon lines are not the ones the packed database gives (%016llx / %016llx)",
                 i == 0 ? "prefix-side key records" : "suffix-side key records", (unsigned long long)g[19 + i],
                 (unsigned long long)g[16 + i]);
        return swa_fail_msg(ctx, SWA_E_INTERNAL, msg);
      }
    }
    ctx->guard_keys_done = true;
  }
  if (ctx->stream_index && ctx->guard_index) {
    for (int i = 0; i < 2; ++i) {
      if (g[i] != g[2 + i] + g[4 + i] + g[6 + i]) {
        snprintf(msg, sizeof(msg), "d=1 guard: index %d has %llu key records but %llu + %llu + %llu grouped members", i, (unsigned long long)g[i],
                 (unsigned long long)g[2 + i], (unsigned long long)g[4 + i], (unsigned long long)g[6 + i]);
        return swa_fail_msg(ctx, SWA_E_INTERNAL, msg);
      }
      if (got[2 + i] != g[2 + i]) {
        snprintf(msg, sizeof(msg), "d=1 guard: pass %d staged %llu members of %llu listed", i, (unsigned long long)got[2 + i], (unsigned long long)g[2 + i]);
        return swa_fail_msg(ctx, SWA_E_INTERNAL, msg);
      }
    }
  }
  if (csr_stream && (csr_end != n_edges || (uint64_t)links_sorted != (n_edges & 0xFFFFFFFFull))) {
    snprintf(msg, sizeof(msg), "d=1 guard: %llu links found, %u sorted, CSR ends at %llu", (unsigned long long)n_edges, links_sorted, (unsigned long long)csr_end);
    return swa_fail_msg(ctx, SWA_E_INTERNAL, msg);
  }
  return SWA_OK;
}

// The network over [first, first + count), left in HBM either as CSR (d_offsets / d_neighbours)
// or, when d_edge_list != nullptr, as one flat list of (source << 32 | target) links.
static int network_run(swa_ctx * ctx, int no_cluster_breaking, uint32_t first, uint32_t count, uint64_t * d_offsets,
                       uint32_t * d_neighbours, uint64_t * d_edge_list, uint64_t cap, uint64_t * total) {
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const char * env_stats = getenv("SWA_D1_STATS");
  const bool stats = env_stats != nullptr && env_stats[0] == '1';
  SWA_TRY(swa_reserve(ctx, ctx->d_counts, uint64_t(count) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_cursor, uint64_t(count) * sizeof(uint32_t)));
  const uint32_t tiles = (count + kScanTile - 1) / kScanTile;
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_tmp, uint64_t(tiles) * sizeof(uint64_t)));
  // hits leave the kernels through per-wave segments of the edge buffer (no shared counter)
  const uint32_t nseg = (uint32_t)ctx->num_cus * 8u * kWaves;          // >= waves of any launch below
  SWA_TRY(swa_reserve(ctx, ctx->d_seg_fill, 3ull * nseg * sizeof(uint32_t)));   // fills | members staged, pass 0 | pass 1
  SWA_TRY(swa_reserve(ctx, ctx->d_guard, 24 * sizeof(uint64_t)));
  if (ctx->seg_cap == 0) {
    ctx->seg_cap = 512;
    while (ctx->seg_cap < 8ull * count / nseg) { ctx->seg_cap <<= 1; }
    const char * env_cap = getenv("SWA_D1_SEG_CAP");          // test hook: start small, exercise the regrow path
    if (env_cap != nullptr && atoi(env_cap) > 0) { ctx->seg_cap = (uint64_t)atoi(env_cap); }
  }
  constexpr uint32_t kLongRowCap = 1u << 16;
  SWA_TRY(swa_reserve(ctx, ctx->d_long_rows, (kLongRowCap + 1ull) * sizeof(uint32_t)));
  uint64_t n_edges = 0;
  uint64_t stream_link_cap = 0;
  bool clean = false;                                        // the last attempt ran to the end without a retry condition
  bool twins = false;                                        // the prefix pass of the pair kernels met identical sequences
  for (int attempt = 0; attempt < 8 && !clean; ++attempt) {
    SWA_TRY(swa_reserve(ctx, ctx->d_edges, uint64_t(nseg) * ctx->seg_cap * sizeof(uint64_t)));
    // (the segment fills start at zero: the anchored route clears them with its other counters — one launch —, the plain one here)
    if (!(ctx->anchor_usable && !stats)) { SWA_HIP(ctx, hipMemsetAsync(ctx->d_seg_fill.ptr, 0, 3ull * nseg * sizeof(uint32_t), ctx->stream)); }
    // CSR by the streaming route (links sorted by source with the partition primitive) unless a flat list is wanted
    const bool csr_stream = d_edge_list == nullptr && stream_csr_enabled() && ctx->anchor_usable && !stats;
    if (ctx->anchor_usable && !stats) {
      if (!(ctx->anchor_ready && ctx->stream_index)) {
        // (the streaming index serves any query range; it is rebuilt when the owner changed)
        swa_t0(ctx, 7);
        SWA_TRY(launch_abundance_rank(ctx));
        // ([2] duplicates [3] seeds left to the plain kernel [4] oversized groups [5] their members: the previous owner's
        // values must not survive into this owner's build — [3] is only ever set, [4] / [5] accumulate)
        SWA_HIP(ctx, hipMemsetAsync(static_cast<uint32_t *>(ctx->d_flags.ptr) + 2, 0, 5 * sizeof(uint32_t), ctx->stream));
        SWA_TRY(build_stream_index(ctx, 0, 0));
        swa_t1(ctx, 7);
        // The index of another owner (swa_d1_set_ownership after the build): the member table and its Bloom filter hold the
        // previous owner's oversized groups, so they are dropped, and whatever this owner's groups leave to the plain kernel
        // ([3] seeds the anchored passes cannot serve, [4] oversized groups) gets the database-wide table
        if (!ctx->full_index) {
          uint32_t fl[6] = {};
          SWA_HIP(ctx, hipMemcpyAsync(fl, ctx->d_flags.ptr, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
          SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
          ctx->member_index = false;
          if (fl[3] != 0 || fl[4] != 0) { SWA_TRY(ensure_full_index(ctx)); }
        }
      }
      SWA_TRY(launch_network_anchored(ctx, no_cluster_breaking, first, count, !csr_stream));
    }
    else {
      SWA_TRY(ensure_full_index(ctx));
      SWA_TRY(launch_network(ctx, no_cluster_breaking, first, count, stats));
    }
    hipLaunchKernelGGL(k_seg_reduce, dim3(1), dim3(1024), 0, ctx->stream, static_cast<const uint32_t *>(ctx->d_seg_fill.ptr),
                       nseg, static_cast<unsigned long long *>(ctx->d_stats.ptr) + 8);
    const bool check_unserved = ctx->anchor_usable && !stats && !ctx->full_index && !ctx->member_index;
    // CSR assembly is enqueued right behind, before the host looks at the totals: every kernel
    // below guards its writes with `cap` / the segment capacity, so a run that turns out to
    // need bigger segments or a bigger neighbour buffer has only wasted these launches.
    // Offsets are always complete; neighbours only if they fit.
    swa_t0(ctx, 4);
    if (d_edge_list != nullptr) {
      SWA_TRY(swa_reserve(ctx, ctx->d_seg_base, uint64_t(nseg) * sizeof(uint64_t)));
      hipLaunchKernelGGL(k_seg_bases, dim3(1), dim3(256), 0, ctx->stream, static_cast<const uint32_t *>(ctx->d_seg_fill.ptr), nseg,
                         static_cast<unsigned long long *>(ctx->d_seg_base.ptr));
      hipLaunchKernelGGL(k_seg_compact, dim3(std::min<uint32_t>(nseg, (uint32_t)ctx->num_cus * 8u)), dim3(256), 0, ctx->stream,
                         static_cast<const uint64_t *>(ctx->d_edges.ptr), static_cast<const uint32_t *>(ctx->d_seg_fill.ptr),
                         nseg, ctx->seg_cap, static_cast<const unsigned long long *>(ctx->d_seg_base.ptr), d_edge_list, cap);
    } else if (csr_stream) {
      stream_link_cap = std::max<uint64_t>(std::max<uint64_t>(cap, stream_link_cap), 2ull * count + 1024);
      SWA_TRY(launch_csr_stream(ctx, first, count, nseg, stream_link_cap, d_offsets, d_neighbours, cap));
    } else {
      hipLaunchKernelGGL((k_scan_tiles<uint32_t>), dim3(tiles), dim3(kScanBlock), 0, ctx->stream,
                         static_cast<const uint32_t *>(ctx->d_counts.ptr), count, static_cast<uint64_t *>(ctx->d_scan_tmp.ptr));
      hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanBlock), 0, ctx->stream, static_cast<uint64_t *>(ctx->d_scan_tmp.ptr), tiles);
      hipLaunchKernelGGL((k_scan_apply<uint32_t>), dim3(tiles), dim3(kScanBlock), 0, ctx->stream,
                         static_cast<const uint32_t *>(ctx->d_counts.ptr), count,
                         static_cast<const uint64_t *>(ctx->d_scan_tmp.ptr), d_offsets);
      if (d_neighbours != nullptr && cap > 0) {
        SWA_HIP(ctx, hipMemsetAsync(ctx->d_cursor.ptr, 0, uint64_t(count) * sizeof(uint32_t), ctx->stream));
        SWA_HIP(ctx, hipMemsetAsync(ctx->d_long_rows.ptr, 0, sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(k_scatter_edges, dim3(std::min<uint32_t>(nseg, (uint32_t)ctx->num_cus * 8u)), dim3(256), 0, ctx->stream,
                           static_cast<const uint64_t *>(ctx->d_edges.ptr), static_cast<const uint32_t *>(ctx->d_seg_fill.ptr),
                           nseg, ctx->seg_cap, first, d_offsets, static_cast<uint32_t *>(ctx->d_cursor.ptr), d_neighbours, cap);
        hipLaunchKernelGGL(k_sort_rows, dim3(grid_for(ctx, count, 256, 8)), dim3(256), 0, ctx->stream, d_offsets, count,
                           d_neighbours, cap, static_cast<uint32_t *>(ctx->d_long_rows.ptr), kLongRowCap);
        hipLaunchKernelGGL(k_sort_long_rows, dim3(ctx->num_cus), dim3(64), 0, ctx->stream, d_offsets, count, d_neighbours, cap,
                           static_cast<const uint32_t *>(ctx->d_long_rows.ptr), kLongRowCap);
      }
    }
    SWA_HIP(ctx, hipGetLastError());
    swa_t1(ctx, 4);
    // ONE copy of the status block's first 512 bytes: flags | stats (links, fullest segment, members staged) | the guard's
    // counters | last CSR offset, links sorted
    auto * status = static_cast<uint64_t *>(ctx->h_status);   // (the pinned mirror)
    SWA_HIP(ctx, hipMemcpyAsync(status, ctx->d_status.ptr, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    uint32_t * unserved_at = reinterpret_cast<uint32_t *>(status + 64);
    *unserved_at = 0;                                        // fallback seeds listed while there is no table to serve them
    if (check_unserved) {
      SWA_HIP(ctx, hipMemcpyAsync(unserved_at, static_cast<uint32_t *>(ctx->d_acounters.ptr) + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t unserved = *unserved_at;
    const uint64_t * got = status + 8 + 8;                    // d_stats[8 ..]: links, fullest segment, members staged by pass
    const uint64_t * guard = status + 24;
    const uint32_t anchor_overflow = reinterpret_cast<const uint32_t *>(status)[2];
    const uint64_t csr_end = status[48];
    const uint32_t links_sorted = (uint32_t)status[49];
    const bool guarded = ctx->anchor_usable && !stats;
    n_edges = got[0];
    if (ctx->anchor_usable && !stats && anchor_overflow != 0) {
      // a bucket of the index rebuilt above held more distinct keys than the group kernel's table: partition finer, again
      ctx->stream_extra_bits = std::min<uint32_t>(ctx->stream_extra_bits + 2, 8);
      ctx->anchor_ready = false;
      continue;
    }
    if (check_unserved && unserved != 0) {                   // should not happen (build_owned_index looks for such seeds)
      SWA_TRY(ensure_full_index(ctx));
      continue;
    }
    if (csr_stream && got[1] <= ctx->seg_cap && n_edges > stream_link_cap) {
      // more links than the internal link buffers of the CSR stage hold (a first call with a small `cap`): bigger, again
      stream_link_cap = n_edges + 1024;
      continue;
    }
    if (got[1] <= ctx->seg_cap) {
      clean = true;
      if (guarded) { SWA_TRY(guard_check(ctx, guard, got, csr_stream, csr_end, links_sorted, n_edges)); }
      twins = guarded && reinterpret_cast<const uint32_t *>(status)[0] != 0u;     // (flags[0]: the prefix pass met two identical sequences)
      break;
    }
    // one wave found more hits than its segment holds: grow the segments and run again (rare).
    // The small-group passes hand out work dynamically, so a wave's fill differs from run to run:
    // twice the observed maximum, not just the maximum
    while (ctx->seg_cap < 2 * got[1]) { ctx->seg_cap <<= 1; }
  }
  if (!clean) {
    // counts[] / offsets include links the segment guards dropped: never hand that out as a result
    return swa_fail_msg(ctx, SWA_E_DEVICE, "swa_d1_network: per-wave link segments still overflowed after 8 attempts");
  }
  *total = n_edges;
  // Identical sequences (the reference: fatal while it fills its table, src/algod1.cc:1131-1150).  On the pair route they are
  // met here — two members of a prefix group whose sequences agree word for word — not by the index build (round 6).
  if (twins) { return swa_fail_msg(ctx, SWA_E_DUPLICATES, "some fasta entries have identical sequences"); }
  if (n_edges > cap) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_d1_network: neighbour buffer too small"); }
  return SWA_OK;
}

// The reference cannot fail here (src/algod1.cc:630-670), so a drop-in must not exit where the reference would not: when
// the guard finds counts that do not balance, everything derived from the uploaded database (lines, lengths, ranks,
// windows, indexes) is made again and the step repeated — once; which count disagreed goes to stderr.  Only a second
// disagreement is SWA_E_INTERNAL.  (A routed build's id lists belong to the caller: multi.hip repeats its own exchange.)
static int network_run_guarded(swa_ctx * ctx, int no_cluster_breaking, uint32_t first, uint32_t count, uint64_t * d_offsets,
                               uint32_t * d_neighbours, uint64_t * d_edge_list, uint64_t cap, uint64_t * total) {
  const int rc = network_run(ctx, no_cluster_breaking, first, count, d_offsets, d_neighbours, d_edge_list, cap, total);
  if (rc != SWA_E_INTERNAL || ctx->index_routed) { return rc; }
  const std::string first_msg = ctx->err;
  fprintf(stderr, "swarm-amd: %s; rebuilding the index from the packed database and repeating the step\n", first_msg.c_str());
  ++ctx->guard_retries;
  ctx->lines_ready = ctx->props_ready = ctx->windows_ready = ctx->rank_ready = false;
  ctx->anchor_ready = ctx->stream_index = ctx->member_index = ctx->full_index = false;
  ctx->guard_index = ctx->guard_keys_done = ctx->guard_keys_pending = false;
  ctx->stream_extra_bits = 0;
  int dup = 0;
  const int rb = swa_d1_index_build_range(ctx, ctx->index_first, ctx->index_count, &dup);
  if (rb != SWA_OK && rb != SWA_E_DUPLICATES) { return rb; }
  const int again = network_run(ctx, no_cluster_breaking, first, count, d_offsets, d_neighbours, d_edge_list, cap, total);
  if (again == SWA_E_INTERNAL) { ctx->err = first_msg + "; after a rebuild: " + ctx->err; }
  return again;
}

extern "C" int swa_d1_guard_retries(const swa_ctx * ctx) { return ctx != nullptr ? (int)ctx->guard_retries : 0; }

extern "C" int swa_d1_network_device(swa_ctx * ctx, int no_cluster_breaking, uint32_t first, uint32_t count,
                                     uint64_t * d_offsets, uint32_t * d_neighbours, uint64_t cap, uint64_t * total) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (!ctx->d1_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_network: call swa_d1_index_build first"); }
  if (count == 0 || (uint64_t)first + count > ctx->db.n || d_offsets == nullptr || total == nullptr ||
      (d_neighbours == nullptr && cap != 0)) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_network: bad range or null buffer");
  }
  return network_run_guarded(ctx, no_cluster_breaking, first, count, d_offsets, d_neighbours, nullptr, cap, total);
}

extern "C" int swa_d1_network_edges_device(swa_ctx * ctx, int no_cluster_breaking, uint32_t first, uint32_t count,
                                           uint64_t * d_edge_list, uint64_t cap, uint64_t * total) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (!ctx->d1_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_network: call swa_d1_index_build first"); }
  if (count == 0 || (uint64_t)first + count > ctx->db.n || d_edge_list == nullptr || total == nullptr) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_network_edges: bad range or null buffer");
  }
  return network_run_guarded(ctx, no_cluster_breaking, first, count, nullptr, nullptr, d_edge_list, cap, total);
}

// multi.hip: the CSR of the whole database from the ranks' link lists as they lie gathered on this device — `lists` runs of
// (source << 32 | target) links, run r = d_links[starts[r] .. + counts[r]) — by the partition + row kernels of the
// single-GPU step (two levels by source bits, then a wave per 256 sources) instead of a 64-bit radix sort of everything.
// Leaves offsets [n + 1] and, if they fit `cap`, the neighbours (rows ascending) in the caller's device buffers.
int swa_d1_csr_from_lists(swa_ctx * ctx, const unsigned long long * d_links, const uint64_t * starts, const uint64_t * counts, uint32_t lists,
                          uint64_t * d_offsets, uint32_t * d_neighbours, uint64_t cap) {
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->db.n;
  uint64_t all = 0, tiles = 2;
  std::vector<uint64_t> h_start(lists);
  std::vector<uint32_t> h_size(lists);
  for (uint32_t r = 0; r < lists; ++r) {
    if (counts[r] > 0xFFFFFFFFull) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_d1_csr_from_lists: a list of more than 2^32 links"); }
    h_start[r] = starts[r]; h_size[r] = (uint32_t)counts[r];
    all += counts[r]; tiles += (counts[r] + 4095) / 4096;
  }
  SWA_TRY(swa_reserve(ctx, ctx->d_seg_base, ((uint64_t)lists * 3 / 2 + 2) * sizeof(uint64_t)));
  auto * d_start = static_cast<uint64_t *>(ctx->d_seg_base.ptr);
  auto * d_size = reinterpret_cast<uint32_t *>(d_start + lists);
  SWA_HIP(ctx, hipMemcpyAsync(d_start, h_start.data(), lists * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  SWA_HIP(ctx, hipMemcpyAsync(d_size, h_size.data(), lists * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));           // (the host vectors are temporaries)
  return csr_from_chunks(ctx, 0, n, d_links, d_start, 0, d_size, lists, 0xFFFFFFFFu, tiles, all + 1, d_offsets, d_neighbours, cap);
}

extern "C" int swa_d1_route_slice(swa_ctx * ctx, uint32_t first, uint32_t count, uint32_t world, uint32_t * d_ids, uint64_t cap,
                                  uint32_t * d_counts) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (ctx->db.n == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_route_slice: no database"); }
  if (first > ctx->db.n || count > ctx->db.n - first) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_route_slice: bad range"); }
  if (world == 0 || world > kRouteMaxWorld || d_ids == nullptr || d_counts == nullptr || cap == 0) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_route_slice: bad argument (1..64 ranks)");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  SWA_TRY(swa_reserve(ctx, ctx->d_flags, 16 * sizeof(uint32_t)));
  SWA_TRY(ensure_anchor_windows(ctx));
  SWA_HIP(ctx, hipMemsetAsync(d_counts, 0, (2ull * world + 1) * sizeof(uint32_t), ctx->stream));
  if (count != 0) {
    hipLaunchKernelGGL(k_anchor_route, dim3(grid_for(ctx, (count + kRoutePerThread - 1) / kRoutePerThread, 256, 8)), dim3(256), 0, ctx->stream,
                       ctx->db.seqs, ctx->db.seq_off, ctx->db.seqlen, first, count, world, ctx->anchor_a, ctx->anchor_b, ctx->anchor_w / 32u, d_ids, cap,
                       d_counts);
  }
  SWA_HIP(ctx, hipGetLastError());
  // the caller reads the counts next (and may do so from another stream: a context without a caller's stream works on
  // its own): they are final when this returns
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}

extern "C" int swa_d1_index_build_routed(swa_ctx * ctx, const uint32_t * d_ids_prefix, uint32_t n_prefix, const uint32_t * d_ids_suffix,
                                         uint32_t n_suffix, int * has_duplicates) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (ctx->db.n == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_index_build_routed: no database"); }
  if (ctx->owner_world <= 1) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_index_build_routed: call swa_d1_set_ownership(rank, world > 1) first"); }
  if ((n_prefix != 0 && d_ids_prefix == nullptr) || (n_suffix != 0 && d_ids_suffix == nullptr)) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_index_build_routed: null list"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  SWA_TRY(ensure_db_lengths(ctx));
  static const uint32_t nothing = 0;                         // (an empty list still marks the build as routed)
  ctx->route_ids[0] = n_prefix != 0 ? d_ids_prefix : &nothing; ctx->route_m[0] = n_prefix;
  ctx->route_ids[1] = n_suffix != 0 ? d_ids_suffix : &nothing; ctx->route_m[1] = n_suffix;
  const int rc = swa_d1_index_build_range(ctx, 0, ctx->db.n, has_duplicates);
  ctx->route_ids[0] = ctx->route_ids[1] = nullptr;
  ctx->route_m[0] = ctx->route_m[1] = 0;
  return rc;
}

// The same exchange with the KEY RECORDS travelling (d1_stream.inc: k_anchor_route_records): 8 bytes per amplicon and index;
// the owner's build starts at the partition.
extern "C" int swa_d1_route_slice_records(swa_ctx * ctx, uint32_t first, uint32_t count, uint32_t world, uint64_t * d_records, uint64_t cap,
                                          uint32_t * d_counts) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (ctx->db.n == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_route_slice_records: no database"); }
  if (first > ctx->db.n || count > ctx->db.n - first) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_route_slice_records: bad range"); }
  if (world == 0 || world > kRouteMaxWorld || d_records == nullptr || d_counts == nullptr || cap == 0) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_route_slice_records: bad argument (1..64 ranks)");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  SWA_TRY(swa_reserve(ctx, ctx->d_flags, 16 * sizeof(uint32_t)));
  SWA_TRY(ensure_anchor_windows(ctx));
  SWA_HIP(ctx, hipMemsetAsync(d_counts, 0, (2ull * world + 1) * sizeof(uint32_t), ctx->stream));
  if (count != 0) {
    hipLaunchKernelGGL(k_anchor_route_records, dim3(grid_for(ctx, (count + 3) / 4, 256, 8)), dim3(256), 0, ctx->stream,
                       ctx->db.seqs, ctx->db.seq_off, ctx->db.seqlen, first, count, world, ctx->anchor_a, ctx->anchor_b, ctx->anchor_w / 32u,
                       reinterpret_cast<unsigned long long *>(d_records), cap, d_counts);
  }
  SWA_HIP(ctx, hipGetLastError());
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));           // (the caller reads the counts next, maybe from another stream)
  return SWA_OK;
}

extern "C" int swa_d1_index_build_records(swa_ctx * ctx, const uint64_t * d_rec_prefix, uint32_t n_prefix, const uint64_t * d_rec_suffix, uint32_t n_suffix,
                                          int * has_duplicates) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (ctx->db.n == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_index_build_records: no database"); }
  if (ctx->owner_world <= 1) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_index_build_records: call swa_d1_set_ownership(rank, world > 1) first"); }
  if ((n_prefix != 0 && d_rec_prefix == nullptr) || (n_suffix != 0 && d_rec_suffix == nullptr)) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_index_build_records: null list");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  SWA_TRY(ensure_db_lengths(ctx));
  static const unsigned long long nothing = 0;               // (an empty list still marks the build as routed)
  ctx->route_rec[0] = n_prefix != 0 ? reinterpret_cast<const unsigned long long *>(d_rec_prefix) : &nothing; ctx->route_m[0] = n_prefix;
  ctx->route_rec[1] = n_suffix != 0 ? reinterpret_cast<const unsigned long long *>(d_rec_suffix) : &nothing; ctx->route_m[1] = n_suffix;
  const int rc = swa_d1_index_build_range(ctx, 0, ctx->db.n, has_duplicates);
  ctx->route_rec[0] = ctx->route_rec[1] = nullptr;
  ctx->route_m[0] = ctx->route_m[1] = 0;
  return rc;
}

extern "C" int swa_d1_set_ownership(swa_ctx * ctx, uint32_t rank, uint32_t world) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (world == 0 || rank >= world) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_set_ownership: bad rank / world"); }
  if (rank != ctx->owner_rank || world != ctx->owner_world) {
    ctx->owner_rank = rank;
    ctx->owner_world = world;
    ctx->csr_ready = false;
    ctx->anchor_ready = false;                               // the next network call indexes this rank's groups
    ctx->member_index = false;                               // (the table of the previous owner's oversized groups is not this one's)
  }
  return SWA_OK;
}

extern "C" int swa_d1_network(swa_ctx * ctx, int no_cluster_breaking, uint32_t first, uint32_t count,
                              uint64_t * offsets, uint32_t * neighbours, uint64_t cap, uint64_t * total) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (offsets == nullptr || total == nullptr) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_network: null buffer"); }
  ctx->csr_ready = false;                                   // (d_offsets_tmp / d_nb_tmp are about to hold this call's rows)
  SWA_TRY(swa_reserve(ctx, ctx->d_offsets_tmp, (uint64_t(count) + 1) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_nb_tmp, (cap > 0 ? cap : 1) * sizeof(uint32_t)));
  const int rc = swa_d1_network_device(ctx, no_cluster_breaking, first, count,
                                       static_cast<uint64_t *>(ctx->d_offsets_tmp.ptr),
                                       static_cast<uint32_t *>(ctx->d_nb_tmp.ptr), cap, total);
  if (rc != SWA_OK && rc != SWA_E_CAPACITY) { return rc; }
  swa_touch_pages(offsets, (uint64_t(count) + 1) * sizeof(uint64_t));
  if (rc == SWA_OK) { swa_touch_pages(neighbours, *total * sizeof(uint32_t)); }
  SWA_HIP(ctx, hipMemcpyAsync(offsets, ctx->d_offsets_tmp.ptr, (uint64_t(count) + 1) * sizeof(uint64_t),
                              hipMemcpyDeviceToHost, ctx->stream));
  if (rc == SWA_OK && *total > 0) {
    SWA_HIP(ctx, hipMemcpyAsync(neighbours, ctx->d_nb_tmp.ptr, *total * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                ctx->stream));
  }
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return rc;
}

extern "C" int swa_d1_debug_read(swa_ctx * ctx, int what, void * out, size_t out_bytes) {
  if (ctx == nullptr || out == nullptr) { return SWA_E_ARG; }
  if (!ctx->d1_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_debug_read: no index"); }
  if (what >= 10) {
    // the streaming index as it lies in HBM (tools/check_stream.py validates it on the host): 10 + i: ids in group order of
    // index i, u32[n] · 12 + i: its item buffer, swa_item[regions' end] · 14: the counters, u32[kCounterWords] · 15: the amplicon lines,
    // lines_quads x 16 bytes each · 16: u64[kWidthClasses][kListKinds + 1] first items of the lists (ListRegions) + [.][0] = line quads behind
    if (!ctx->stream_index) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_debug_read: no streaming index in place"); }
    const void * from = nullptr;
    size_t need = 0;
    if (what == 10 || what == 11) { from = ctx->d_stream[kSbMembers + (what - 10)].ptr; need = uint64_t(ctx->db.n) * sizeof(uint32_t); }
    else if (what == 12 || what == 13) { from = ctx->d_aitems[what - 12].ptr; need = ctx->list_regions_items * sizeof(swa_item); }
    else if (what == 14) { from = ctx->d_acounters.ptr; need = kCounterWords * sizeof(uint32_t); }
    else if (what == 15) { from = ctx->d_stream[kSbLines].ptr; need = uint64_t(ctx->db.n) * ctx->lines_quads * 16u; }
    else if (what == 16) {                                    // (host-side facts: where the lists lie, how wide a line is)
      uint64_t total = 0;
      const ListRegions r = list_regions(ctx, &total);
      const size_t bytes = sizeof(r) + 2 * sizeof(uint64_t);
      if (out_bytes < bytes) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_d1_debug_read: buffer too small"); }
      memcpy(out, &r, sizeof(r));
      const uint64_t tail[2] = {total, ctx->lines_quads};
      memcpy(static_cast<char *>(out) + sizeof(r), tail, sizeof(tail));
      return SWA_OK;
    }
    else { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_debug_read: unknown selector"); }
    if (out_bytes < need) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_d1_debug_read: buffer too small"); }
    SWA_HIP(ctx, hipSetDevice(ctx->device));
    SWA_HIP(ctx, hipMemcpyAsync(out, from, need, hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SWA_OK;
  }
  SWA_TRY(ensure_full_index(ctx));
  const void * src = nullptr;
  size_t bytes = 0;
  switch (what) {
    case 0: src = ctx->d_seqhash.ptr; bytes = uint64_t(ctx->db.n) * sizeof(uint64_t); break;
    case 1: src = ctx->d_bloom.ptr; bytes = ctx->bloom_words * sizeof(uint64_t); break;
    case 2: src = ctx->d_zobrist.ptr; bytes = 4ull * ctx->zobrist_len * sizeof(uint64_t); break;
    case 3: src = ctx->d_stats.ptr; bytes = 8 * sizeof(uint64_t); break;
    default: return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_debug_read: unknown selector");
  }
  if (out_bytes < bytes) { return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_d1_debug_read: buffer too small"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  SWA_HIP(ctx, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}

// ---- seam B2: the fastidious second pass --------------------------------------------------
// algo_d1_run's fastidious branch (src/algod1.cc:1337-1467).  Two routes, chosen per PAIR of
// (heavy, light) amplicons by the shorter of the two lengths:
//   pair route  (both >= kFastMinLen; d1_fast.inc): groups by shared 32-nt windows, exact
//               "within two edits" test per pair, exact |V1(h) ∩ V1(x)| per surviving pair;
//   Bloom route (a sequence shorter than that; the reference's own scheme): light-only table + Bloom,
//               every microvariant of every light amplicon clears its k pattern bits in the flexible
//               Bloom (atomicAnd: the reference's plain `&=` from many threads can lose a bit; -t 1 is
//               the specification), every microvariant of every heavy amplicon is tested, survivors
//               become 16-byte tasks, one wave per task re-runs the d=1 probe (k_d1_probe<MODE 1>).
// Both end in atomicMin on graft_cand = add_graft_candidate (src/algod1.cc:244-258) and one counter.
extern "C" int swa_d1_fastidious(swa_ctx * ctx, const uint8_t * is_light, uint64_t light_nt, uint32_t bloom_bits,
                                 uint32_t * graft_cand, uint64_t * counters) {
  return swa_d1_fastidious_shard(ctx, is_light, light_nt, bloom_bits, 0, 1, graft_cand, counters);
}

// fc layout (d_fcounters, u64[16]): [0] light variants [1] heavy variants [2] candidates [3] tasks of the
// current batch [4] variants of the current batch [5] pairs found [6] short light amplicons [7] short heavy
// amplicons [8] nucleotides of the short light amplicons [9] scratch
static int fastidious_bloom_route(swa_ctx * ctx, uint32_t n_light, uint32_t n_heavy, uint64_t light_nt, uint32_t bloom_bits,
                                  uint32_t pair_route_min_len) {
  SWA_TRY(ensure_full_index(ctx));                           // hashes of every amplicon, room for the table
  uint32_t k = static_cast<uint32_t>(0.4 * static_cast<double>(bloom_bits));
  if (k < 1) { k = 1; }
  uint64_t m = light_nt * 7ull * bloom_bits;
  if (m < 64) { m = 64; }
  const uint64_t n_bytes = ((m - 1) / 8) + 1;
  const uint64_t fsize = n_bytes >> 3;
  std::vector<uint64_t> fpat;
  swa_bloom_patterns(65536, k, fpat);
  SWA_TRY(swa_reserve(ctx, ctx->d_bloomflex, fsize * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_fpatterns, fpat.size() * sizeof(uint64_t)));
  const uint64_t maxv = 7ull * ctx->db.longest + 4ull;
  uint64_t task_cap = uint64_t(n_heavy) * maxv;
  const uint64_t cap_limit = (1ull << 30) / sizeof(swa_task);            // 1 GiB of tasks
  if (task_cap > cap_limit) { task_cap = cap_limit; }
  if (task_cap < maxv) { task_cap = maxv; }
  SWA_TRY(swa_reserve(ctx, ctx->d_queue, task_cap * sizeof(swa_task)));
  SWA_HIP(ctx, hipMemcpyAsync(ctx->d_fpatterns.ptr, fpat.data(), fpat.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_bloomflex.ptr, 0xFF, fsize * sizeof(uint64_t), ctx->stream));
  ctx->d1_ready = false;                                     // the table now holds (short) light amplicons only
  ctx->full_index = false;
  SWA_TRY(swa_d1_rebuild_table(ctx, static_cast<const uint8_t *>(ctx->d_light.ptr)));

  auto * fc = static_cast<unsigned long long *>(ctx->d_fcounters.ptr);
  FlexArgs f{};
  f.seqs = ctx->db.seqs; f.seq_off = ctx->db.seq_off; f.seqlen = ctx->db.seqlen;
  f.zobrist = static_cast<const uint64_t *>(ctx->d_zobrist.ptr);
  f.zlen = ctx->zobrist_len;
  f.maxwords = (ctx->db.longest + 1u + 31u) >> 5;
  f.fbits = static_cast<unsigned long long *>(ctx->d_bloomflex.ptr);
  f.fsize = fsize;
  f.finv = 1.0 / static_cast<double>(fsize);
  f.fpat = static_cast<const uint64_t *>(ctx->d_fpatterns.ptr);
  f.tasks = static_cast<swa_task *>(ctx->d_queue.ptr);
  f.task_counter = fc + 3;
  f.task_cap = task_cap;
  const bool zlds = 4ull * ctx->zobrist_len * sizeof(uint64_t) <= kMaxZobristLds;
  const size_t flex_lds = sizeof(uint64_t) * ((zlds ? 4ull * ctx->zobrist_len : 0ull) + kWaves * (size_t)(f.maxwords + 2u)) +
                          kWaves * 256 * sizeof(swa_task);
  // light side
  f.list = static_cast<const uint32_t *>(ctx->d_list_a.ptr);
  f.count = n_light;
  f.variant_counter = fc + 9;                                // (the reported totals are computed for all amplicons, not here)
  {
    const int grid = grid_for(ctx, f.count, kWaves, 8);
    if (zlds) { hipLaunchKernelGGL((k_d1_flex<true, 0>), dim3(grid), dim3(kThreads), flex_lds, ctx->stream, f); }
    else { hipLaunchKernelGGL((k_d1_flex<false, 0>), dim3(grid), dim3(kThreads), flex_lds, ctx->stream, f); }
    SWA_HIP(ctx, hipGetLastError());
  }
  // heavy side
  NetArgs a{};
  a.seqs = ctx->db.seqs; a.seq_off = ctx->db.seq_off; a.seqlen = ctx->db.seqlen; a.abundance = ctx->db.abundance;
  a.zobrist = f.zobrist; a.zlen = ctx->zobrist_len;
  a.maxwords = f.maxwords;                                   // a microvariant can be one nt longer
  a.table = static_cast<const swa_slot *>(ctx->d_table.ptr);
  a.tmask = ctx->table_size - 1;
  a.bloom = static_cast<const uint64_t *>(ctx->d_bloom.ptr);
  a.bmask = ctx->bloom_words - 1;
  a.patterns = static_cast<const uint64_t *>(ctx->d_patterns.ptr);
  a.stats = static_cast<unsigned long long *>(ctx->d_stats.ptr);
  a.tasks = f.tasks;
  a.graft = static_cast<uint32_t *>(ctx->d_graft.ptr);
  a.cand_counter = fc + 2;
  a.fast_min_len = pair_route_min_len;
  const size_t probe_lds = sizeof(uint64_t) * ((zlds ? 4ull * ctx->zobrist_len : 0ull) + 1024ull +
                                               kWaves * ((size_t)(a.maxwords + 2u) + kQueueCap + kQueueCap / 2));
  uint64_t done = 0;
  uint64_t batch = task_cap / maxv;                          // cannot overflow
  double rate = -1.0;                                        // observed tasks per heavy amplicon
  while (done < n_heavy) {
    uint64_t want = batch;
    if (rate >= 0.0) {                                       // optimistic sizing, overflow is detected below
      const double est = static_cast<double>(task_cap) / (4.0 * rate + 1.0);
      want = est > 4.0e9 ? 4000000000ull : static_cast<uint64_t>(est);
      if (want < batch) { want = batch; }
    }
    if (want > n_heavy - done) { want = n_heavy - done; }
    SWA_HIP(ctx, hipMemsetAsync(fc + 3, 0, 2 * sizeof(uint64_t), ctx->stream));    // tasks + batch variants
    f.list = static_cast<const uint32_t *>(ctx->d_list_b.ptr) + done;
    f.count = static_cast<uint32_t>(want);
    f.variant_counter = fc + 4;
    const int grid = grid_for(ctx, f.count, kWaves, 8);
    if (zlds) { hipLaunchKernelGGL((k_d1_flex<true, 1>), dim3(grid), dim3(kThreads), flex_lds, ctx->stream, f); }
    else { hipLaunchKernelGGL((k_d1_flex<false, 1>), dim3(grid), dim3(kThreads), flex_lds, ctx->stream, f); }
    SWA_HIP(ctx, hipGetLastError());
    uint64_t got[2] = {0, 0};
    SWA_HIP(ctx, hipMemcpyAsync(got, fc + 3, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rate = static_cast<double>(got[0]) / static_cast<double>(want);
    if (got[0] > task_cap) { continue; }                     // optimistic batch did not fit: redo it smaller
    if (got[0] > 0) {
      a.count = static_cast<uint32_t>(got[0]);
      const int pgrid = grid_for(ctx, a.count, kWaves, 8);
      if (zlds) { hipLaunchKernelGGL((k_d1_probe<true, false, 1>), dim3(pgrid), dim3(kThreads), probe_lds, ctx->stream, a); }
      else { hipLaunchKernelGGL((k_d1_probe<false, false, 1>), dim3(pgrid), dim3(kThreads), probe_lds, ctx->stream, a); }
      SWA_HIP(ctx, hipGetLastError());
    }
    done += want;
  }
  return SWA_OK;
}

// LDS of k_fast_count for `waves` waves per block; 0 = does not fit
static size_t fast_count_lds(const swa_ctx * ctx, uint32_t slots, int waves) {
  const uint32_t maxwords = (ctx->db.longest + 31u) >> 5;
  const size_t bytes = sizeof(uint64_t) * (4ull * ctx->zobrist_len + (size_t)waves * (2ull * (maxwords + 3u) + slots));
  return bytes <= 160u * 1024u ? bytes : 0;
}

static int fastidious_pair_route(swa_ctx * ctx, uint32_t n_light, uint32_t n_heavy, uint32_t slots, int count_waves) {
  const uint32_t n = ctx->db.n;
  auto * fc = static_cast<unsigned long long *>(ctx->d_fcounters.ptr);
  if (n_light == 0 || n_heavy == 0) { return SWA_OK; }
  uint64_t asize_max = 64, asize_ends = 64;
  while (asize_max < 6ull * n_light) { asize_max <<= 1; }    // load <= 0.5 with three memberships per light amplicon (middle windows)
  while (asize_ends < 2ull * n_light) { asize_ends <<= 1; }  // one membership (prefix / suffix groups): a smaller table to clear and scan
  uint64_t asize = asize_max;
  const uint64_t member_cap = 3ull * n_light + n_heavy;
  const uint64_t item_cap64 = 3ull * n_light + (3ull * n_light + n_heavy) / 2 + 64;
  const uint32_t item_cap = item_cap64 > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)item_cap64;
  SWA_TRY(swa_reserve(ctx, ctx->d_fkeys, asize * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_fcnt, asize * 5 * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_foff, (asize + 1) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_fslot, uint64_t(n) * 4 * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_fmembers, member_cap * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_fitems, uint64_t(item_cap) * sizeof(swa_fitem)));
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_tmp, ((asize_max + kScanTile - 1) / kScanTile) * sizeof(uint64_t)));
  if (ctx->fast_pair_cap == 0) { ctx->fast_pair_cap = 4ull * n_light + (1ull << 20); }
  auto * keys = static_cast<unsigned long long *>(ctx->d_fkeys.ptr);
  auto * cnt_l = static_cast<uint32_t *>(ctx->d_fcnt.ptr);
  uint32_t * cnt_h = cnt_l + asize, * cur_l = cnt_h + asize, * cur_h = cur_l + asize, * tot = cur_h + asize;
  auto * offsets = static_cast<uint64_t *>(ctx->d_foff.ptr);
  auto * lslot = static_cast<uint32_t *>(ctx->d_fslot.ptr);
  auto * hslot = lslot + 3ull * n;
  auto * members = static_cast<uint32_t *>(ctx->d_fmembers.ptr);
  auto * items = static_cast<swa_fitem *>(ctx->d_fitems.ptr);
  auto * dflags = static_cast<uint32_t *>(ctx->d_flags.ptr);
  auto * item_counter = dflags + 8;                          // [8] item counter, [9] key table overflow
  uint64_t npairs = 0;
  // pairs on the amplicon lines, sequences in registers (up to 416 nt; SWA_FAST_PAIRS=words: the round-2 kernel, which
  // walks the packed sequences — comparison switch)
  const char * env_fp = getenv("SWA_FAST_PAIRS");
  // (the register kernels exist for 5, 8 and 13 words: sequences up to 416 nt, which 128-byte lines hold)
  const int pair_w = (env_fp != nullptr && env_fp[0] == 'w') ? 0 : (ctx->db.longest <= 160u ? 5 : (ctx->db.longest <= 256u ? 8 : (ctx->db.longest <= 416u ? 13 : 0)));
  if (pair_w != 0) {
    SWA_TRY(launch_abundance_rank(ctx));
    SWA_TRY(ensure_lines(ctx));
  }
  swa_t0(ctx, 5);
  for (int attempt = 0; attempt < 6; ++attempt) {
    SWA_TRY(swa_reserve(ctx, ctx->d_fpairs, ctx->fast_pair_cap * sizeof(uint64_t)));
    SWA_HIP(ctx, hipMemsetAsync(fc + 5, 0, sizeof(uint64_t), ctx->stream));
    SWA_HIP(ctx, hipMemsetAsync(dflags + 9, 0, sizeof(uint32_t), ctx->stream));
    for (int type = 0; type < 3; ++type) {
      asize = type == 2 ? asize_max : asize_ends;
      cnt_h = cnt_l + asize; cur_l = cnt_h + asize; cur_h = cur_l + asize; tot = cur_h + asize;
      const uint32_t tiles = (uint32_t)((asize + kScanTile - 1) / kScanTile);
      FastGroupArgs g{};
      g.seqs = ctx->db.seqs; g.seq_off = ctx->db.seq_off; g.seqlen = ctx->db.seqlen;
      g.role = static_cast<const uint8_t *>(ctx->d_frole.ptr); g.n = n;
      g.keys = keys; g.cnt_l = cnt_l; g.cnt_h = cnt_h; g.amask = asize - 1; g.lslot = lslot; g.hslot = hslot; g.overflow = dflags + 9;
      const dim3 gn(grid_for(ctx, n, 256, 8)), ga(grid_for(ctx, asize, 256, 8)), b(256);
      hipLaunchKernelGGL(k_fg_clear, ga, b, 0, ctx->stream, keys, cnt_l, cnt_h, cur_l, cur_h, asize);
      if (type == 0) { hipLaunchKernelGGL(k_fg_light<0>, gn, b, 0, ctx->stream, g); hipLaunchKernelGGL(k_fg_heavy<0>, gn, b, 0, ctx->stream, g); }
      else if (type == 1) { hipLaunchKernelGGL(k_fg_light<1>, gn, b, 0, ctx->stream, g); hipLaunchKernelGGL(k_fg_heavy<1>, gn, b, 0, ctx->stream, g); }
      else { hipLaunchKernelGGL(k_fg_light<2>, gn, b, 0, ctx->stream, g); hipLaunchKernelGGL(k_fg_heavy<2>, gn, b, 0, ctx->stream, g); }
      hipLaunchKernelGGL(k_fg_totals, ga, b, 0, ctx->stream, cnt_l, cnt_h, asize, tot);
      hipLaunchKernelGGL((k_scan_tiles<uint32_t>), dim3(tiles), dim3(kScanBlock), 0, ctx->stream, tot, (uint32_t)asize, static_cast<uint64_t *>(ctx->d_scan_tmp.ptr));
      hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanBlock), 0, ctx->stream, static_cast<uint64_t *>(ctx->d_scan_tmp.ptr), tiles);
      hipLaunchKernelGGL((k_scan_apply<uint32_t>), dim3(tiles), dim3(kScanBlock), 0, ctx->stream, tot, (uint32_t)asize,
                         static_cast<const uint64_t *>(ctx->d_scan_tmp.ptr), offsets);
      if (type == 0) { hipLaunchKernelGGL(k_fg_scatter<0>, gn, b, 0, ctx->stream, g, offsets, cur_l, cur_h, members); }
      else if (type == 1) { hipLaunchKernelGGL(k_fg_scatter<1>, gn, b, 0, ctx->stream, g, offsets, cur_l, cur_h, members); }
      else { hipLaunchKernelGGL(k_fg_scatter<2>, gn, b, 0, ctx->stream, g, offsets, cur_l, cur_h, members); }
      SWA_HIP(ctx, hipMemsetAsync(item_counter, 0, sizeof(uint32_t), ctx->stream));
      hipLaunchKernelGGL(k_fg_items, ga, b, 0, ctx->stream, cnt_l, cnt_h, offsets, asize, items, item_counter, item_cap);
      FastPairArgs p{};
      p.seqs = ctx->db.seqs; p.seq_off = ctx->db.seq_off; p.seqlen = ctx->db.seqlen; p.members = members;
      p.items = items; p.item_count = item_counter; p.item_cap = item_cap;
      p.pairs = static_cast<unsigned long long *>(ctx->d_fpairs.ptr); p.pair_counter = fc + 5; p.pair_cap = ctx->fast_pair_cap;
      const dim3 gp(ctx->num_cus * 8);
      p.lines = static_cast<const uint4 *>(ctx->d_stream[kSbLines].ptr);
      p.line_quads = ctx->lines_quads;
      if (pair_w == 5) {
        if (type == 0) { hipLaunchKernelGGL((k_fast_pairs_lines<0, 5>), gp, dim3(kThreads), 0, ctx->stream, p); }
        else if (type == 1) { hipLaunchKernelGGL((k_fast_pairs_lines<1, 5>), gp, dim3(kThreads), 0, ctx->stream, p); }
        else { hipLaunchKernelGGL((k_fast_pairs_lines<2, 5>), gp, dim3(kThreads), 0, ctx->stream, p); }
      } else if (pair_w == 8) {
        if (type == 0) { hipLaunchKernelGGL((k_fast_pairs_lines<0, 8>), gp, dim3(kThreads), 0, ctx->stream, p); }
        else if (type == 1) { hipLaunchKernelGGL((k_fast_pairs_lines<1, 8>), gp, dim3(kThreads), 0, ctx->stream, p); }
        else { hipLaunchKernelGGL((k_fast_pairs_lines<2, 8>), gp, dim3(kThreads), 0, ctx->stream, p); }
      } else if (pair_w == 13) {
        if (type == 0) { hipLaunchKernelGGL((k_fast_pairs_lines<0, 13>), gp, dim3(kThreads), 0, ctx->stream, p); }
        else if (type == 1) { hipLaunchKernelGGL((k_fast_pairs_lines<1, 13>), gp, dim3(kThreads), 0, ctx->stream, p); }
        else { hipLaunchKernelGGL((k_fast_pairs_lines<2, 13>), gp, dim3(kThreads), 0, ctx->stream, p); }
      }
      else if (type == 0) { hipLaunchKernelGGL(k_fast_pairs<0>, gp, dim3(kThreads), 0, ctx->stream, p); }
      else if (type == 1) { hipLaunchKernelGGL(k_fast_pairs<1>, gp, dim3(kThreads), 0, ctx->stream, p); }
      else { hipLaunchKernelGGL(k_fast_pairs<2>, gp, dim3(kThreads), 0, ctx->stream, p); }
      SWA_HIP(ctx, hipGetLastError());
    }
    uint64_t got = 0;
    uint32_t fl[2] = {0, 0};
    SWA_HIP(ctx, hipMemcpyAsync(&got, fc + 5, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipMemcpyAsync(fl, dflags + 8, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (fl[1] != 0) { return swa_fail_msg(ctx, SWA_E_DEVICE, "swa_d1_fastidious: group key table overflow"); }   // cannot happen: load <= 0.5
    if (got <= ctx->fast_pair_cap) { npairs = got; break; }
    if (attempt == 5) { return swa_fail_msg(ctx, SWA_E_NOMEM, "swa_d1_fastidious: pair list keeps overflowing"); }
    ctx->fast_pair_cap = got + got / 8 + 1024;               // the count of a complete run: size for it (+ slack: none needed, it is exact)
  }
  swa_t1(ctx, 5);
  swa_t0(ctx, 6);
  if (npairs != 0) {
    FastCountArgs c{};
    c.seqs = ctx->db.seqs; c.seq_off = ctx->db.seq_off; c.seqlen = ctx->db.seqlen;
    c.zobrist = static_cast<const uint64_t *>(ctx->d_zobrist.ptr);
    c.zlen = ctx->zobrist_len; c.maxwords = (ctx->db.longest + 31u) >> 5; c.slots = slots;
    c.pairs = static_cast<const unsigned long long *>(ctx->d_fpairs.ptr); c.npairs = npairs;
    c.graft = static_cast<uint32_t *>(ctx->d_graft.ptr); c.cand_counter = fc + 2;
    const char * env_sets = getenv("SWA_FAST_COUNT_SETS");     // test switch: the LDS-set kernel whatever the length
    const bool by_sites = ctx->db.longest <= 255u && !(env_sets != nullptr && env_sets[0] == '1');
    if (by_sites) {
      uint64_t blocks = (npairs + kWaves - 1) / kWaves;
      const uint64_t max_blocks = (uint64_t)ctx->num_cus * 8;
      if (blocks > max_blocks) { blocks = max_blocks; }
      if (ctx->db.longest <= 159u) { hipLaunchKernelGGL(k_fast_count_sites<5>, dim3((uint32_t)blocks), dim3(kThreads), 0, ctx->stream, c); }
      else { hipLaunchKernelGGL(k_fast_count_sites<8>, dim3((uint32_t)blocks), dim3(kThreads), 0, ctx->stream, c); }
    } else {
    const size_t lds = fast_count_lds(ctx, slots, count_waves);
    uint64_t blocks = (npairs + count_waves - 1) / count_waves;
    const uint64_t max_blocks = (uint64_t)ctx->num_cus * 8;
    if (blocks > max_blocks) { blocks = max_blocks; }
    hipLaunchKernelGGL(k_fast_count, dim3((uint32_t)blocks), dim3(64 * count_waves), lds, ctx->stream, c);
    }
    SWA_HIP(ctx, hipGetLastError());
  }
  swa_t1(ctx, 6);
  return SWA_OK;
}

extern "C" int swa_d1_fastidious_shard(swa_ctx * ctx, const uint8_t * is_light, uint64_t light_nt, uint32_t bloom_bits,
                                       uint32_t shard, uint32_t nshards, uint32_t * graft_cand, uint64_t * counters) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (nshards == 0 || shard >= nshards) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_fastidious_shard: bad shard"); }
  if (!ctx->d1_ready) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_fastidious: call swa_d1_index_build first"); }
  if (is_light == nullptr || graft_cand == nullptr || counters == nullptr || bloom_bits < 2 || bloom_bits > 64) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_d1_fastidious: bad argument");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  SWA_TRY(prepare_hashing(ctx));                             // the Zobrist table
  const uint32_t n = ctx->db.n;
  // the numbers of the reference's log line (src/algod1.cc:1337-1357, 1383-1403; bloomflex.cc:91-115)
  uint32_t k = static_cast<uint32_t>(0.4 * static_cast<double>(bloom_bits));
  if (k < 1) { k = 1; }
  uint64_t m = light_nt * 7ull * bloom_bits;
  if (m < 64) { m = 64; }

  // role of every amplicon: 0 light, 1 heavy and in this shard's contiguous slice of the heavy ones, 2 neither
  // (by blocks on a few threads: two serial passes over 10 M flags were 15 ms of this call, lease r6j)
  std::unique_ptr<uint8_t[]> role(new uint8_t[n]);
  uint64_t n_light = 0, n_heavy_all = 0;
  const unsigned blocks = n >= (1u << 20) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
  std::vector<uint64_t> heavy_before(blocks + 1, 0);
  auto in_blocks = [&](auto && body) {
    std::vector<std::thread> team;
    for (unsigned b = 1; b < blocks; ++b) { team.emplace_back([&, b] { body(b); }); }
    body(0u);
    for (auto & t : team) { t.join(); }
  };
  in_blocks([&](unsigned b) {
    uint64_t heavy = 0;
    for (uint64_t i = (uint64_t)n * b / blocks; i < (uint64_t)n * (b + 1) / blocks; ++i) { heavy += is_light[i] == 0 ? 1u : 0u; }
    heavy_before[b + 1] = heavy;
  });
  for (unsigned b = 0; b < blocks; ++b) { heavy_before[b + 1] += heavy_before[b]; }
  n_heavy_all = heavy_before[blocks];
  n_light = n - n_heavy_all;
  const uint64_t lo = n_heavy_all * shard / nshards, hi = n_heavy_all * (shard + 1ull) / nshards;
  in_blocks([&](unsigned b) {
    uint64_t seen = heavy_before[b];
    for (uint64_t i = (uint64_t)n * b / blocks; i < (uint64_t)n * (b + 1) / blocks; ++i) {
      if (is_light[i] != 0) { role[i] = 0; }
      else { role[i] = (seen >= lo && seen < hi) ? 1 : 2; ++seen; }
    }
  });
  const uint64_t n_heavy = hi - lo;

  // which pairs the pair route can take: k_fast_count's LDS set must hold the microvariants of the longest sequence
  uint32_t slots = 1024;
  { const uint64_t v = 7ull * ctx->db.longest + 4ull; while (slots < v + v / 2) { slots <<= 1; } }
  int count_waves = 0;
  for (int w : {4, 2, 1}) { if (count_waves == 0 && fast_count_lds(ctx, slots, w) != 0) { count_waves = w; } }
  const char * env_route = getenv("SWA_FAST_BLOOM");          // test hook: the reference's scheme for every pair
  const bool pair_route = count_waves != 0 && ctx->db.longest >= kFastMinLen && !(env_route != nullptr && env_route[0] == '1');
  const uint32_t min_len = pair_route ? kFastMinLen : 0xFFFFFFFFu;

  SWA_TRY(swa_reserve(ctx, ctx->d_frole, n));
  SWA_TRY(swa_reserve(ctx, ctx->d_light, n));
  SWA_TRY(swa_reserve(ctx, ctx->d_graft, uint64_t(n) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_list_a, (uint64_t(n) + 1) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_list_b, (uint64_t(n) + 1) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_fcounters, 16 * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_flags, 16 * sizeof(uint32_t)));
  SWA_HIP(ctx, hipMemcpyAsync(ctx->d_frole.ptr, role.get(), n, hipMemcpyHostToDevice, ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_graft.ptr, 0xFF, uint64_t(n) * sizeof(uint32_t), ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_fcounters.ptr, 0, 16 * sizeof(uint64_t), ctx->stream));
  auto * fc = static_cast<unsigned long long *>(ctx->d_fcounters.ptr);
  for (int slot : {5, 6}) { ctx->ev_used[slot] = false; }

  // the log's variant totals, and the amplicons that can be part of a pair with a short sequence
  hipLaunchKernelGGL(k_fast_variant_totals, dim3(grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, ctx->db.seqs, ctx->db.seq_off,
                     ctx->db.seqlen, static_cast<const uint8_t *>(ctx->d_frole.ptr), n, fc);
  const uint32_t short_cut = pair_route ? kFastMinLen + 1u : 0xFFFFFFFFu;   // len <= cut: may pair with a sequence < kFastMinLen
  hipLaunchKernelGGL(k_fast_short_lists, dim3(grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, ctx->db.seqlen,
                     static_cast<const uint8_t *>(ctx->d_frole.ptr), n, short_cut, static_cast<uint8_t *>(ctx->d_light.ptr),
                     static_cast<uint32_t *>(ctx->d_list_a.ptr), static_cast<uint32_t *>(ctx->d_list_b.ptr), fc + 6);
  SWA_HIP(ctx, hipGetLastError());
  uint64_t shorts[3] = {0, 0, 0};
  SWA_HIP(ctx, hipMemcpyAsync(shorts, fc + 6, sizeof(shorts), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));           // (role is a host temporary, too)

  if (pair_route) { SWA_TRY(fastidious_pair_route(ctx, (uint32_t)n_light, (uint32_t)n_heavy, slots, count_waves)); }
  if (shorts[0] != 0 && shorts[1] != 0) {
    SWA_TRY(fastidious_bloom_route(ctx, (uint32_t)shorts[0], (uint32_t)shorts[1], shorts[2], bloom_bits, min_len));
  }

  uint64_t host_fc[8] = {};
  SWA_HIP(ctx, hipMemcpyAsync(host_fc, fc, sizeof(host_fc), hipMemcpyDeviceToHost, ctx->stream));
  swa_touch_pages(graft_cand, uint64_t(n) * sizeof(uint32_t));
  SWA_HIP(ctx, hipMemcpyAsync(graft_cand, ctx->d_graft.ptr, uint64_t(n) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  counters[0] = host_fc[0];
  counters[1] = host_fc[1];
  counters[2] = host_fc[2];
  counters[3] = m;
  counters[4] = k;
  return SWA_OK;
}
