// dn_graph.hip — d >= 2 in BULK: every ordered pair (query, target) of amplicons whose alignment has at
// most d differences, as one CSR, without the greedy loop.
//
// The reference (src/algo.cc:384-602) interleaves the search with the clustering: one q-gram scan +
// alignment call per seed and sub-seed, against the shrinking pool — ~120 k dependent steps for 1 M
// amplicons, each a few kernels and a host round trip, which is what bounded the fused scan of
// scan.hip at 3 % of the HBM roofline.  But which pairs are within d differences does not depend on
// the pool (scan.hip's batching already relies on that), the abundance rule is a property of the
// pair, and the triangle-inequality prune (src/algo.cc:521-522) only skips pairs that cannot match.
// So the whole search is a function of the database:  G = { (q, t, diff(q, t)) : diff <= d }  —
// the same object as the d = 1 network, and the clustering becomes the host walk over it
// (cluster_dn.cpp), exactly like d = 1.
//
// Finding the pairs without comparing everything with everything: an alignment with <= d
// non-identical columns is <= d edits, so of d + 1 disjoint windows of the query (W nucleotides at
// offsets 0, W, 2W, ...) at least one is untouched, and it reappears in the target shifted by the net
// indels before it, -d .. +d.  Every amplicon is filed as a TARGET under its windows at all those
// shifts and looks itself up as a QUERY under its unshifted windows ("groups", one window at a
// time); inside a group every (query < target) pair goes through: length difference, "not already
// found through an earlier window", the q-gram bound (src/qgram.cc:68-96, B3) — survivors are
// aligned by the B4 kernels (align.hip) in the direction(s) the abundance rule allows, and the
// accepted (query, target, diff) triples are sorted into a CSR (rocPRIM radix sort).
//
// Applies when every sequence holds d + 1 windows of 32 (or 16) nucleotides; otherwise the fused
// scan of scan.hip serves (swa_dn_graph_supported).  HBM traffic: group bookkeeping (a few tens of
// bytes per amplicon and window) + the members' sequences and signatures, mostly from L2.
#include "swa_internal.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <vector>

int swa_align_launch(swa_ctx * ctx, uint32_t query, const uint32_t * d_queries, const uint32_t * d_targets,
                     const uint32_t * d_count, uint32_t max_count, uint32_t * d_diffs, uint32_t * d_scores,
                     uint32_t * d_alnlens);

namespace {

constexpr uint32_t kEmpty = SWA_NO_AMPLICON;
constexpr uint64_t kKeyEmpty = ~0ull;
constexpr uint32_t kTile = 4096;          // (query, target) pairs per turn of a wave
constexpr uint32_t kStride = 64;          // a group's tiles are dealt round-robin to at most this many items
constexpr uint32_t kStage = 256;          // per-wave staging of found pairs
constexpr int kMaxShifts = 17;            // 2 d + 1 for d <= 8

struct dg_item { uint32_t begin, nt, nq, tile; };   // members[begin, begin+nt) targets, then nq queries

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

// `wlen` (<= 32) nucleotides from position pos on (may read the following word: the database ends in zero words)
__device__ __forceinline__ uint64_t window(const uint64_t * seq, uint32_t pos, uint32_t wlen) {
  const uint32_t w = pos >> 5, sh = (pos & 31u) << 1;
  uint64_t v = seq[w] >> sh;
  if (sh != 0u) { v |= seq[w + 1] << (64u - sh); }
  return wlen >= 32u ? v : (v & ((1ull << (2u * wlen)) - 1ull));
}

__device__ __forceinline__ uint64_t window_key(uint64_t w, uint32_t k) {
  return mix64(w ^ (0xA24BAED4963EE407ull * (uint64_t)(k + 1u))) & 0x7FFFFFFFFFFFFFFFull;
}

struct GroupArgs {
  const uint64_t * seqs;
  const uint64_t * seq_off;
  const uint32_t * seqlen;
  uint32_t n, d, k, wlen;        // window k at offset k * wlen
  unsigned long long * keys;
  uint32_t * cnt_t, * cnt_q;
  uint64_t amask;
  uint32_t * tslot;              // [n * (2 d + 1)]
  uint32_t * qslot;              // [n]
  uint32_t * overflow;
  uint32_t owner_rank, owner_world;   // world > 1: only the window keys this rank owns make groups here (swa_dn_set_ownership)
};

__global__ __launch_bounds__(256) void k_dg_clear(unsigned long long * keys, uint32_t * c0, uint32_t * c1, uint32_t * c2, uint32_t * c3,
                                                  uint64_t asize) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < asize; i += (uint64_t)gridDim.x * blockDim.x) {
    keys[i] = kKeyEmpty; c0[i] = 0u; c1[i] = 0u; c2[i] = 0u; c3[i] = 0u;
  }
}

// every amplicon is a target under its window k at the shifts -d .. +d
__global__ __launch_bounds__(256) void k_dg_targets(const GroupArgs a) {
  const uint32_t ns = 2u * a.d + 1u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
    const uint32_t len = a.seqlen[i];
    const uint64_t * s = a.seqs + a.seq_off[i];
    uint64_t seen[kMaxShifts];
    for (uint32_t j = 0; j < ns; ++j) {
      uint32_t slot = kEmpty;
      const int64_t pos = (int64_t)a.k * a.wlen + (int64_t)j - (int64_t)a.d;
      seen[j] = kKeyEmpty;
      if (pos >= 0 && (uint64_t)pos + a.wlen <= len) {
        const uint64_t key = window_key(window(s, (uint32_t)pos, a.wlen), a.k);
        bool repeat = false;                                  // (low complexity: the same window at two shifts — one membership)
        for (uint32_t q = 0; q < j; ++q) { repeat = repeat || seen[q] == key; }
        seen[j] = key;
        // (ownership by bits of the mixed key that the table index — its low bits — does not use alone)
        const bool owned = a.owner_world == 1u || (uint32_t)((((mix64(key) >> 40) & 0xFFFFFFull) * a.owner_world) >> 24) == a.owner_rank;
        if (!repeat && owned) {
          uint64_t idx = mix64(key) & a.amask;
          bool placed = false;
          for (uint64_t probes = 0; probes <= a.amask; ++probes) {
            const unsigned long long old = atomicCAS(&a.keys[idx], kKeyEmpty, (unsigned long long)key);
            if (old == kKeyEmpty || old == key) { placed = true; break; }
            idx = (idx + 1) & a.amask;
          }
          if (placed) { atomicAdd(&a.cnt_t[idx], 1u); slot = (uint32_t)idx; }
          else { *a.overflow = 1u; }
        }
      }
      a.tslot[(uint64_t)i * ns + j] = slot;
    }
  }
}

// ... and a query under the unshifted window (its own target entry at shift 0 made the group)
__global__ __launch_bounds__(256) void k_dg_queries(const GroupArgs a) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
    const uint64_t key = window_key(window(a.seqs + a.seq_off[i], a.k * a.wlen, a.wlen), a.k);
    uint64_t idx = mix64(key) & a.amask;
    uint32_t slot = kEmpty;
    for (uint64_t probes = 0; probes <= a.amask; ++probes) {
      const unsigned long long have = a.keys[idx];
      if (have == key) { atomicAdd(&a.cnt_q[idx], 1u); slot = (uint32_t)idx; break; }
      if (have == kKeyEmpty) { break; }
      idx = (idx + 1) & a.amask;
    }
    a.qslot[i] = slot;
  }
}

// room in the member list only for groups with a pair of DIFFERENT amplicons (an amplicon alone is its own target)
__global__ __launch_bounds__(256) void k_dg_totals(const uint32_t * __restrict__ cnt_t, const uint32_t * __restrict__ cnt_q,
                                                   uint64_t asize, uint32_t * __restrict__ tot) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < asize; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t t = cnt_t[i], q = cnt_q[i];
    tot[i] = t * q >= 2 ? (uint32_t)(t + q) : 0u;
  }
}

__global__ __launch_bounds__(256) void k_dg_scatter(const GroupArgs a, const uint32_t * __restrict__ tot,
                                                    const uint64_t * __restrict__ offsets, uint32_t * cur_t, uint32_t * cur_q,
                                                    uint32_t * __restrict__ members) {
  const uint32_t ns = 2u * a.d + 1u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
    for (uint32_t j = 0; j < ns; ++j) {
      const uint32_t s = a.tslot[(uint64_t)i * ns + j];
      if (s != kEmpty && tot[s] != 0u) { members[offsets[s] + atomicAdd(&cur_t[s], 1u)] = i; }
    }
    const uint32_t s = a.qslot[i];
    if (s != kEmpty && tot[s] != 0u) { members[offsets[s] + a.cnt_t[s] + atomicAdd(&cur_q[s], 1u)] = i; }
  }
}

// BLOCKED = false: the tiles of k_dg_pairs (4096 pairs, at most kStride items a group); true: the blocks of k_dg_pairs_lds
// (kBlockT targets x kBlockQ queries each, item.tile = target block + target blocks x query chunk)
constexpr uint32_t kBlockT = 64;          // targets of a block: one per lane
constexpr uint32_t kBlockQ = 256;         // queries of a block, staged kStageQ at a time
constexpr uint32_t kStageQ = 16;
template <bool BLOCKED>
__global__ __launch_bounds__(256) void k_dg_items(const uint32_t * __restrict__ cnt_t, const uint32_t * __restrict__ cnt_q,
                                                  const uint32_t * __restrict__ tot, const uint64_t * __restrict__ offsets,
                                                  uint64_t asize, dg_item * items, uint32_t * counter, uint32_t cap) {
  __shared__ uint32_t n_items, base;
  if (threadIdx.x == 0) { n_items = 0u; }
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t start = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto items_of = [&](uint64_t s) -> uint32_t {
    if (tot[s] == 0u) { return 0u; }
    if (BLOCKED) {
      const uint64_t blocks = (uint64_t)((cnt_t[s] + kBlockT - 1u) / kBlockT) * ((cnt_q[s] + kBlockQ - 1u) / kBlockQ);
      return (uint32_t)(blocks < 0x7FFFFFFFull ? blocks : 0x7FFFFFFFull);
    }
    const uint64_t tiles = ((uint64_t)cnt_t[s] * cnt_q[s] + kTile - 1) / kTile;
    return (uint32_t)(tiles < kStride ? tiles : kStride);
  };
  for (uint64_t s = start; s < asize; s += stride) {
    const uint32_t k = items_of(s);
    if (k != 0u) { atomicAdd(&n_items, k); }
  }
  __syncthreads();
  if (threadIdx.x == 0) { base = n_items != 0u ? atomicAdd(counter, n_items) : 0u; n_items = 0u; }
  __syncthreads();
  for (uint64_t s = start; s < asize; s += stride) {
    const uint32_t k = items_of(s);
    if (k == 0u) { continue; }
    const uint32_t at = base + atomicAdd(&n_items, k);
    dg_item it;
    it.begin = (uint32_t)offsets[s]; it.nt = cnt_t[s]; it.nq = cnt_q[s];
    for (uint32_t t = 0; t < k; ++t) { it.tile = t; if (at + t < cap) { items[at + t] = it; } }
  }
}

struct PairArgs {
  const uint64_t * seqs;
  const uint64_t * seq_off;
  const uint32_t * seqlen;
  const ulonglong2 * sigs;       // q-gram signatures, 8 x 16 bytes per amplicon
  const uint32_t * members;
  const dg_item * items;
  const uint32_t * item_count;
  uint32_t item_cap;
  uint32_t d, k, wlen;
  unsigned long long * pairs;    // (query << 32) | target, query < target
  unsigned long long * counters; // [0] pairs [1] q-gram comparisons
  uint64_t pair_cap;
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  const int lo = __shfl((int)(uint32_t)v, src, 64);
  const int hi = __shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// all (query, target) pairs of a group with query < target, one pair per lane and turn
__global__ __launch_bounds__(256) void k_dg_pairs(const PairArgs a) {
  __shared__ unsigned long long stage_all[4][kStage];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long * stage = stage_all[wave];
  uint32_t nstage = 0;
  unsigned long long compared = 0;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  auto flush = [&]() {
    wave_lds_sync();
    unsigned long long base = 0;
    if (lane == 0) { base = atomicAdd(&a.counters[0], (unsigned long long)nstage); }
    base = shfl_u64(base, 0);
    for (uint32_t i = lane; i < nstage; i += 64u) { if (base + i < a.pair_cap) { a.pairs[base + i] = stage[i]; } }
    nstage = 0;
    wave_lds_sync();
  };
  const uint32_t nitems = min(*a.item_count, a.item_cap);
  const uint32_t nwaves = gridDim.x * 4u;
  const int d = (int)a.d;
  for (uint32_t it = blockIdx.x * 4u + wave; it < nitems; it += nwaves) {
    const dg_item item = a.items[it];
    const uint64_t npairs = (uint64_t)item.nt * item.nq;
    const uint64_t ntiles = (npairs + kTile - 1) / kTile;
    const uint32_t * targets = a.members + item.begin;
    const uint32_t * queries = targets + item.nt;
    for (uint64_t tile = item.tile; tile < ntiles; tile += kStride) {
      const uint64_t q0 = tile * kTile;
      const uint64_t q1 = min(q0 + (uint64_t)kTile, npairs);
      for (uint64_t qb = q0; qb < q1; qb += 64u) {
        const uint64_t p = qb + (uint64_t)lane;
        bool take = false;
        uint32_t q = 0, t = 0;
        if (p < q1) {
          const uint32_t iq = (uint32_t)(p / item.nt);
          const uint32_t itg = (uint32_t)(p - (uint64_t)iq * item.nt);
          q = queries[iq];
          t = targets[itg];
          if (q < t) {
            const int lq = (int)a.seqlen[q], lt = (int)a.seqlen[t];
            const int dl = lq - lt;
            if (dl >= -d && dl <= d) {
              const uint64_t * sq = a.seqs + a.seq_off[q];
              const uint64_t * st = a.seqs + a.seq_off[t];
              // the pair belongs to the FIRST window of the query that reappears (shifted) in the target
              bool earlier = false;
              for (uint32_t k2 = 0; k2 < a.k && !earlier; ++k2) {
                const uint64_t wq = window(sq, k2 * a.wlen, a.wlen);
                for (int s = -d; s <= d && !earlier; ++s) {
                  const int pos = (int)(k2 * a.wlen) + s;
                  earlier = pos >= 0 && pos + (int)a.wlen <= lt && window(st, (uint32_t)pos, a.wlen) == wq;
                }
              }
              if (!earlier) {
                // (the group key may collide: make sure window k really reappears)
                bool here = false;
                const uint64_t wq = window(sq, a.k * a.wlen, a.wlen);
                for (int s = -d; s <= d && !here; ++s) {
                  const int pos = (int)(a.k * a.wlen) + s;
                  here = pos >= 0 && pos + (int)a.wlen <= lt && window(st, (uint32_t)pos, a.wlen) == wq;
                }
                if (here) {
                  // q-gram bound (qgram_diff, src/qgram.cc:68-96): ceil(popcount(sig_q ^ sig_t) / 10) <= d
                  const ulonglong2 * gq = a.sigs + (uint64_t)q * 8u;
                  const ulonglong2 * gt = a.sigs + (uint64_t)t * 8u;
                  uint32_t pop = 0;
#pragma unroll
                  for (int w = 0; w < 8; ++w) {
                    const ulonglong2 x = gq[w], y = gt[w];
                    pop += (uint32_t)__popcll(x.x ^ y.x) + (uint32_t)__popcll(x.y ^ y.y);
                  }
                  ++compared;
                  take = (pop + 9u) / 10u <= a.d;
                }
              }
            }
          }
        }
        const uint64_t m = __ballot(take);
        if (m != 0ull) {
          if (take) { stage[nstage + (uint32_t)__popcll(m & lane_lt)] = ((unsigned long long)q << 32) | t; }
          nstage += (uint32_t)__popcll(m);
          if (nstage > kStage - 64u) { flush(); }
        }
      }
    }
  }
  if (nstage != 0u) { flush(); }
  for (int o = 32; o > 0; o >>= 1) { compared += shfl_u64(compared, lane ^ o); }
  if (lane == 0 && compared != 0ull) { atomicAdd(&a.counters[1], compared); }
}

// The same pairs, block by block (r05).  k_dg_pairs computes every pair from global memory: two lengths, two offsets, up
// to (k + 1)(2 d + 1) windows cut out of the target's words and two 128-byte signatures PER PAIR — 16 ms of 57 at
// 1 M x 400, d = 3, 0.07 of the HBM roofline on bytes that mostly came from L2.  Here a wave takes a
// block of a group: 64 targets, one per lane, each with its windows at every shift (cut out once), its validity bits and
// its signature in registers; the queries pass by in LDS, kStageQ at a time — their windows, lengths, ids and
// signatures are read by all lanes at the same address (a broadcast) — so a pair costs ~2 instructions of a wave, and
// the global reads are one signature and a few words per MEMBER of the block.  D = the template's d (<= 3: 28 window
// registers); other d: k_dg_pairs.
template <int D>
__global__ __launch_bounds__(256) void k_dg_pairs_lds(const PairArgs a) {
  constexpr int NS = 2 * D + 1;
  struct QRec { uint64_t sig[16]; uint64_t win[D + 1]; uint32_t id, len; };
  __shared__ unsigned long long stage_all[4][kStage];
  __shared__ QRec qrec_all[4][kStageQ];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long * stage = stage_all[wave];
  QRec * qrec = qrec_all[wave];
  uint32_t nstage = 0;
  unsigned long long compared = 0;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  auto flush = [&]() {
    wave_lds_sync();
    unsigned long long base = 0;
    if (lane == 0) { base = atomicAdd(&a.counters[0], (unsigned long long)nstage); }
    base = shfl_u64(base, 0);
    for (uint32_t i = lane; i < nstage; i += 64u) { if (base + i < a.pair_cap) { a.pairs[base + i] = stage[i]; } }
    nstage = 0;
    wave_lds_sync();
  };
  const uint32_t nitems = min(*a.item_count, a.item_cap);
  const uint32_t nwaves = gridDim.x * 4u;
  const uint32_t K = a.k;                                     // this launch's window (<= D), wave-uniform
  for (uint32_t it = blockIdx.x * 4u + wave; it < nitems; it += nwaves) {
    const dg_item item = a.items[it];
    const uint32_t * targets = a.members + item.begin;
    const uint32_t * queries = targets + item.nt;
    const uint32_t ntb = (item.nt + kBlockT - 1u) / kBlockT;
    const uint32_t tb = item.tile % ntb, qc = item.tile / ntb;
    // ---- my target: windows at every shift of the windows 0 .. K, validity, signature
    const uint32_t ti = tb * kBlockT + (uint32_t)lane;
    const bool have_t = ti < item.nt;
    const uint32_t t = have_t ? targets[ti] : 0u;
    const int lt = have_t ? (int)a.seqlen[t] : 0;
    const uint64_t * st = a.seqs + a.seq_off[t];
    uint64_t twin[D + 1][NS];
    uint32_t valid = 0u;
#pragma unroll
    for (int k2 = 0; k2 <= D; ++k2) {
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const int pos = k2 * (int)a.wlen + s2 - D;
        const bool ok = have_t && (uint32_t)k2 <= K && pos >= 0 && pos + (int)a.wlen <= lt;
        twin[k2][s2] = ok ? window(st, (uint32_t)pos, a.wlen) : 0ull;
        valid |= ok ? 1u << (k2 * NS + s2) : 0u;
      }
    }
    uint64_t tsig[16];
    {
      const ulonglong2 * gt = a.sigs + (uint64_t)t * 8u;
#pragma unroll
      for (int w = 0; w < 8; ++w) { const ulonglong2 y = have_t ? gt[w] : make_ulonglong2(0ull, 0ull); tsig[2 * w] = y.x; tsig[2 * w + 1] = y.y; }
    }
    const uint32_t q_begin = qc * kBlockQ, q_end = min(item.nq, q_begin + kBlockQ);
    for (uint32_t qs = q_begin; qs < q_end; qs += kStageQ) {
      const uint32_t nq_here = min(kStageQ, q_end - qs);
      wave_lds_sync();
      // ---- the next queries into LDS: lanes [16 j', 16 j' + 16) carry the signature words of query 4 r + j'
      for (uint32_t r = 0; r < kStageQ / 4u; ++r) {
        const uint32_t j = 4u * r + ((uint32_t)lane >> 4), w = (uint32_t)lane & 15u;
        if (j < nq_here) {
          const uint32_t q = queries[qs + j];
          qrec[j].sig[w] = reinterpret_cast<const uint64_t *>(a.sigs + (uint64_t)q * 8u)[w];
        }
      }
      if ((uint32_t)lane < nq_here) {
        const uint32_t q = queries[qs + (uint32_t)lane];
        const uint64_t * sq = a.seqs + a.seq_off[q];
        qrec[lane].id = q; qrec[lane].len = a.seqlen[q];
#pragma unroll
        for (int k2 = 0; k2 <= D; ++k2) { qrec[lane].win[k2] = (uint32_t)k2 <= K ? window(sq, (uint32_t)k2 * a.wlen, a.wlen) : 0ull; }
      }
      wave_lds_sync();
      for (uint32_t j = 0; j < nq_here; ++j) {
        const uint32_t q = qrec[j].id;
        const int dl = (int)qrec[j].len - lt;
        bool take = have_t && q < t && dl >= -D && dl <= D;
        // the pair belongs to the FIRST window of the query that reappears (shifted) in the target
        bool earlier = false, here = false;
#pragma unroll
        for (int k2 = 0; k2 <= D; ++k2) {
          if ((uint32_t)k2 > K) { continue; }                  // (wave-uniform: the windows behind this launch's are nobody's business here)
          const uint64_t wq = qrec[j].win[k2];
          bool hit = false;
#pragma unroll
          for (int s2 = 0; s2 < NS; ++s2) { hit = hit || (((valid >> (k2 * NS + s2)) & 1u) != 0u && twin[k2][s2] == wq); }
          earlier = earlier || ((uint32_t)k2 < K && hit);
          here = here || ((uint32_t)k2 == K && hit);        // (the group key may collide: window K must really reappear)
        }
        take = take && !earlier && here;
        if (__ballot(take) != 0ull) {
          // q-gram bound (qgram_diff, src/qgram.cc:68-96): ceil(popcount(sig_q ^ sig_t) / 10) <= d
          uint32_t pop = 0;
#pragma unroll
          for (int w = 0; w < 16; ++w) { pop += (uint32_t)__popcll(qrec[j].sig[w] ^ tsig[w]); }
          if (take) { ++compared; }
          take = take && (pop + 9u) / 10u <= (uint32_t)D;
        }
        const uint64_t m = __ballot(take);
        if (m != 0ull) {
          if (take) { stage[nstage + (uint32_t)__popcll(m & lane_lt)] = ((unsigned long long)q << 32) | t; }
          nstage += (uint32_t)__popcll(m);
          if (nstage > kStage - 64u) { flush(); }
        }
      }
    }
  }
  if (nstage != 0u) { flush(); }
  for (int o = 32; o > 0; o >>= 1) { compared += shfl_u64(compared, lane ^ o); }
  if (lane == 0 && compared != 0ull) { atomicAdd(&a.counters[1], compared); }
}

// the alignments a pair needs: (a, b) with a < b always (db order is abundance-descending, so b's
// abundance is not the larger one); (b, a) as well when the abundance rule allows it: equal abundances,
// or no rule (-n).  Queries / targets of the second kind are appended behind the first npairs entries.
__global__ __launch_bounds__(256) void k_dg_worklist(const unsigned long long * __restrict__ pairs, uint64_t npairs,
                                                     const uint64_t * __restrict__ abundance, int ncb, uint32_t * __restrict__ wq,
                                                     uint32_t * __restrict__ wt, unsigned long long * extra_counter) {
  // (2048 pairs a workgroup and turn, ONE atomic for the second directions it adds — see k_dg_edges)
  constexpr uint32_t kPer = 8;
  __shared__ uint32_t wave_sum[4];
  __shared__ unsigned long long block_base;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint64_t p0 = (uint64_t)blockIdx.x * (256u * kPer); p0 < npairs; p0 += (uint64_t)gridDim.x * (256u * kPer)) {
    uint32_t qa[kPer], tb[kPer];
    uint32_t both = 0u, mine = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      const uint64_t p = p0 + (uint64_t)k * 256u + threadIdx.x;
      qa[k] = 0u; tb[k] = 0u;
      if (p < npairs) {
        const unsigned long long pr = pairs[p];
        qa[k] = (uint32_t)(pr >> 32); tb[k] = (uint32_t)pr;
        wq[p] = qa[k]; wt[p] = tb[k];
        if (ncb != 0 || abundance[qa[k]] == abundance[tb[k]]) { both |= 1u << k; ++mine; }
      }
    }
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, (unsigned)o, 64); if ((int)lane >= o) { incl += up; } }
    if (lane == 63u) { wave_sum[wave] = incl; }
    __syncthreads();
    uint32_t before = incl - mine, total = 0u;
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) { before += w < wave ? wave_sum[w] : 0u; total += wave_sum[w]; }
    if (threadIdx.x == 0 && total != 0u) { block_base = atomicAdd(extra_counter, (unsigned long long)total); }
    __syncthreads();
    if (both != 0u) {
      uint64_t at = npairs + block_base + before;
#pragma unroll
      for (uint32_t k = 0; k < kPer; ++k) {
        if ((both >> k & 1u) != 0u) { wq[at] = tb[k]; wt[at] = qa[k]; ++at; }
      }
    }
    __syncthreads();
  }
}

// Round 6: the work list in the order of the alignments' expected length.  k_align_wfa runs a wave — four pairs — until its
// LAST pair is done: a pair within d stops at its score, a pair beyond d takes every step; in the order the pair kernels
// found them nearly every wave holds one of the latter.  Ordered by the q-gram distance of the two signatures (what the
// filter measured: 0..30 differing bits for the pairs it let through) plus ten a nucleotide of length difference (a gap
// costs more than a mismatch), pairs that stop early share their waves.
// key[i] for the work list's item i (both directions carry their pair's key).
__global__ __launch_bounds__(256) void k_dg_work_keys(const uint32_t * __restrict__ wq, const uint32_t * __restrict__ wt, uint64_t nwork,
                                                      const ulonglong2 * __restrict__ sigs, const uint32_t * __restrict__ seqlen, uint32_t len_weight,
                                                      unsigned char * __restrict__ key, unsigned long long * __restrict__ packed) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwork; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t q = wq[i], t = wt[i];
    const ulonglong2 * sq = sigs + (uint64_t)q * 8u;
    const ulonglong2 * st = sigs + (uint64_t)t * 8u;
    uint32_t pop = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) { const ulonglong2 x = sq[w], y = st[w]; pop += (uint32_t)__popcll(x.x ^ y.x) + (uint32_t)__popcll(x.y ^ y.y); }
    const uint32_t lq = seqlen[q], lt = seqlen[t];
    key[i] = (unsigned char)min(pop + len_weight * (max(lq, lt) - min(lq, lt)), 63u);
    packed[i] = ((unsigned long long)q << 32) | t;
  }
}
__global__ __launch_bounds__(256) void k_dg_work_unpack(const unsigned long long * __restrict__ packed, uint64_t nwork, uint32_t * __restrict__ wq,
                                                        uint32_t * __restrict__ wt) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwork; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long v = packed[i];
    wq[i] = (uint32_t)(v >> 32); wt[i] = (uint32_t)v;
  }
}

// accepted alignments -> (query << 32 | target) keys + diffs, compacted.  A workgroup takes 2048 work items a turn — eight a
// thread — and appends what it keeps with ONE atomic on the counter (a single address takes ~90 atomics a microsecond: one
// per wave of 64 items was 50 000 of them, 0.55 of this kernel's 0.60 ms at 3.2 M items; round 6)
__global__ __launch_bounds__(256) void k_dg_edges(const uint32_t * __restrict__ wq, const uint32_t * __restrict__ wt,
                                                  const uint32_t * __restrict__ diffs, uint64_t count, uint32_t d,
                                                  unsigned long long * __restrict__ keys, uint32_t * __restrict__ vals,
                                                  unsigned long long * edge_counter) {
  constexpr uint32_t kPer = 8;
  __shared__ uint32_t wave_sum[4];
  __shared__ unsigned long long block_base;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint64_t e0 = (uint64_t)blockIdx.x * (256u * kPer); e0 < count; e0 += (uint64_t)gridDim.x * (256u * kPer)) {
    uint32_t dv[kPer];
    uint32_t keep = 0u, mine = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      const uint64_t e = e0 + (uint64_t)k * 256u + threadIdx.x;
      dv[k] = e < count ? diffs[e] : 0xFFFFFFFFu;
      if (e < count && dv[k] <= d) { keep |= 1u << k; ++mine; }
    }
    // the thread's place among the workgroup's kept items: a wave scan + the waves before
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, (unsigned)o, 64); if ((int)lane >= o) { incl += up; } }
    if (lane == 63u) { wave_sum[wave] = incl; }
    __syncthreads();
    uint32_t before = incl - mine, total = 0u;
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) { before += w < wave ? wave_sum[w] : 0u; total += wave_sum[w]; }
    if (threadIdx.x == 0 && total != 0u) { block_base = atomicAdd(edge_counter, (unsigned long long)total); }
    __syncthreads();
    if (keep != 0u) {
      uint64_t at = block_base + before;
#pragma unroll
      for (uint32_t k = 0; k < kPer; ++k) {
        if ((keep >> k & 1u) != 0u) {
          const uint64_t e = e0 + (uint64_t)k * 256u + threadIdx.x;
          keys[at] = ((unsigned long long)wq[e] << 32) | wt[e];
          vals[at] = dv[k];
          ++at;
        }
      }
    }
    __syncthreads();
  }
}

// sorted keys -> CSR: offsets[i] = first key >= i << 32; neighbours / diffs split out of the sorted arrays
__global__ __launch_bounds__(256) void k_dg_offsets(const unsigned long long * __restrict__ keys, uint64_t count, uint32_t n,
                                                    uint64_t * __restrict__ offsets) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long want = (unsigned long long)i << 32;
    uint64_t lo = 0, hi = count;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (keys[mid] < want) { lo = mid + 1; } else { hi = mid; } }
    offsets[i] = lo;
  }
}

__global__ __launch_bounds__(256) void k_dg_split(const unsigned long long * __restrict__ keys, const uint32_t * __restrict__ vals,
                                                  uint64_t count, uint32_t * __restrict__ neighbours, uint8_t * __restrict__ diffs) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
    neighbours[i] = (uint32_t)keys[i];
    diffs[i] = (uint8_t)vals[i];
  }
}

// diff of the link parent[v] -> v (rows ascending: binary search), 0 for a seed
__global__ __launch_bounds__(256) void k_dg_parent_diffs(const uint64_t * __restrict__ offsets, const uint32_t * __restrict__ nb,
                                                         const uint8_t * __restrict__ df, const uint32_t * __restrict__ parent, uint32_t n,
                                                         uint8_t * __restrict__ out) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const uint32_t p = parent[v];
    uint8_t d = 0;
    if (p != SWA_NO_AMPLICON) {
      uint64_t lo = offsets[p], hi = offsets[p + 1];
      while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (nb[mid] < v) { lo = mid + 1; } else { hi = mid; }
      }
      d = (lo < offsets[p + 1] && nb[lo] == v) ? df[lo] : (uint8_t)0xFF;
    }
    out[v] = d;
  }
}

__global__ __launch_bounds__(256) void k_dg_shortest(const uint32_t * __restrict__ seqlen, uint32_t n, uint32_t * out) {
  uint32_t mn = 0xFFFFFFFFu;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { mn = min(mn, seqlen[i]); }
  for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64)); }
  if ((threadIdx.x & 63u) == 0u) { atomicMin(out, mn); }
}

struct widen_u32 {
  __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

int grid_for(const swa_ctx * ctx, uint64_t items) {
  uint64_t blocks = (items + 255) / 256;
  const uint64_t cap = (uint64_t)ctx->num_cus * 8;
  return (int)std::max<uint64_t>(1, std::min(blocks, cap));
}

// window length for this database and d: 32, 16, or 0 (no room for d + 1 windows in the shortest sequence)
int window_length(swa_ctx * ctx, uint32_t d, uint32_t * out) {
  if (ctx->dn_shortest == 0) {
    SWA_TRY(swa_reserve(ctx, ctx->d_flags, 16 * sizeof(uint32_t)));
    auto * slot = static_cast<uint32_t *>(ctx->d_flags.ptr) + 10;
    SWA_HIP(ctx, hipMemsetAsync(slot, 0xFF, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_dg_shortest, dim3(grid_for(ctx, ctx->db.n)), dim3(256), 0, ctx->stream, ctx->db.seqlen, ctx->db.n, slot);
    uint32_t mn = 0;
    SWA_HIP(ctx, hipMemcpyAsync(&mn, slot, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->dn_shortest = mn;
  }
  *out = 0;
  if (2u * d + 1u > (uint32_t)kMaxShifts) { return SWA_OK; }
  if (ctx->dn_shortest >= 32u * (d + 1u)) { *out = 32; }
  else if (ctx->dn_shortest >= 16u * (d + 1u)) { *out = 16; }
  return SWA_OK;
}

}  // namespace

extern "C" int swa_dn_graph_supported(swa_ctx * ctx) {
  if (ctx == nullptr || !ctx->search_ready || ctx->db.n == 0) { return 0; }
  uint32_t wlen = 0;
  if (window_length(ctx, (uint32_t)ctx->resolution, &wlen) != SWA_OK) { return 0; }
  return wlen != 0 ? 1 : 0;
}

// the search itself: the sorted (query << 32 | target, diff) list of this context's share of the graph, left in HBM
int swa_dn_graph_compute(swa_ctx * ctx, int no_cluster_breaking) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (!ctx->qgram_ready || !ctx->search_ready) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_dn_graph: call swa_qgram_build and swa_search_begin first");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->db.n;
  const uint32_t d = (uint32_t)ctx->resolution;
  uint32_t wlen = 0;
  SWA_TRY(window_length(ctx, d, &wlen));
  if (wlen == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_dn_graph: a sequence is too short for d + 1 windows (use the scan)"); }

  if (!ctx->dn_graph_ready || ctx->dn_graph_ncb != (no_cluster_breaking != 0)) {
    const uint32_t ns = 2u * d + 1u;
    uint64_t asize = 64;
    while (asize < 2ull * n * ns) { asize <<= 1; }
    const uint64_t member_cap = (uint64_t)n * (ns + 1u);
    const uint64_t item_cap64 = (uint64_t)n + member_cap / 2 + 64;
    const uint32_t item_cap = item_cap64 > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)item_cap64;
    SWA_TRY(swa_reserve(ctx, ctx->d_fkeys, asize * sizeof(uint64_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_fcnt, (asize * 5 + 4) * sizeof(uint32_t)));              // (the scan reads one entry past `tot`)
    SWA_TRY(swa_reserve(ctx, ctx->d_foff, (asize + 2) * sizeof(uint64_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_fslot, (uint64_t)n * (ns + 1u) * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_fmembers, member_cap * sizeof(uint32_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_fitems, (uint64_t)item_cap * sizeof(dg_item)));
    SWA_TRY(swa_reserve(ctx, ctx->d_fcounters, 16 * sizeof(uint64_t)));
    SWA_TRY(swa_reserve(ctx, ctx->d_flags, 16 * sizeof(uint32_t)));
    auto * keys = static_cast<unsigned long long *>(ctx->d_fkeys.ptr);
    auto * cnt_t = static_cast<uint32_t *>(ctx->d_fcnt.ptr);
    auto * cnt_q = cnt_t + asize, * cur_t = cnt_q + asize, * cur_q = cur_t + asize, * tot = cur_q + asize;
    auto * goff = static_cast<uint64_t *>(ctx->d_foff.ptr);
    auto * tslot = static_cast<uint32_t *>(ctx->d_fslot.ptr);
    auto * qslot = tslot + (uint64_t)n * ns;
    auto * members = static_cast<uint32_t *>(ctx->d_fmembers.ptr);
    auto * items = static_cast<dg_item *>(ctx->d_fitems.ptr);
    auto * fc = static_cast<unsigned long long *>(ctx->d_fcounters.ptr);   // [0] pairs [1] comparisons [2] extra directions [3] edges
    auto * dflags = static_cast<uint32_t *>(ctx->d_flags.ptr);             // [8] item counter [9] key table overflow
    size_t scan_bytes = 0;
    auto tot64 = rocprim::make_transform_iterator(tot, widen_u32());
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, tot64, goff, (uint64_t)0, asize + 1, rocprim::plus<uint64_t>(), ctx->stream);
    SWA_TRY(swa_reserve(ctx, ctx->d_scan_tmp, scan_bytes + 16));
    if (ctx->dn_pair_cap == 0) { ctx->dn_pair_cap = 16ull * n + (1ull << 20); }
    uint64_t npairs = 0;
    uint64_t launches = 0;
    for (int slot : {5, 6}) { ctx->ev_used[slot] = false; }
    swa_t0(ctx, 5);                                            // timing slot 5: groups + pairs, slot 6: alignments + CSR
    for (int attempt = 0; attempt < 6; ++attempt) {
      SWA_TRY(swa_reserve(ctx, ctx->d_fpairs, ctx->dn_pair_cap * sizeof(uint64_t)));
      SWA_HIP(ctx, hipMemsetAsync(fc, 0, 8 * sizeof(uint64_t), ctx->stream));
      SWA_HIP(ctx, hipMemsetAsync(dflags + 9, 0, sizeof(uint32_t), ctx->stream));
      for (uint32_t k = 0; k <= d; ++k) {
        GroupArgs g{};
        g.seqs = ctx->db.seqs; g.seq_off = ctx->db.seq_off; g.seqlen = ctx->db.seqlen; g.n = n; g.d = d; g.k = k; g.wlen = wlen;
        g.keys = keys; g.cnt_t = cnt_t; g.cnt_q = cnt_q; g.amask = asize - 1; g.tslot = tslot; g.qslot = qslot; g.overflow = dflags + 9;
        g.owner_rank = ctx->dn_owner_rank; g.owner_world = ctx->dn_owner_world;
        const dim3 gn(grid_for(ctx, n)), ga(grid_for(ctx, asize)), b(256);
        hipLaunchKernelGGL(k_dg_clear, ga, b, 0, ctx->stream, keys, cnt_t, cnt_q, cur_t, cur_q, asize);
        hipLaunchKernelGGL(k_dg_targets, gn, b, 0, ctx->stream, g);
        hipLaunchKernelGGL(k_dg_queries, gn, b, 0, ctx->stream, g);
        hipLaunchKernelGGL(k_dg_totals, ga, b, 0, ctx->stream, cnt_t, cnt_q, asize, tot);
        SWA_HIP(ctx, rocprim::exclusive_scan(ctx->d_scan_tmp.ptr, scan_bytes, tot64, goff, (uint64_t)0, asize + 1, rocprim::plus<uint64_t>(),
                                             ctx->stream));
        hipLaunchKernelGGL(k_dg_scatter, gn, b, 0, ctx->stream, g, tot, goff, cur_t, cur_q, members);
        SWA_HIP(ctx, hipMemsetAsync(dflags + 8, 0, sizeof(uint32_t), ctx->stream));
        // blocks of 64 targets x 256 queries for the LDS kernel (d = 2, 3; SWA_DN_PAIRS=plain: the per-pair kernel, as for other d)
        static const bool plain_pairs = [] { const char * e = getenv("SWA_DN_PAIRS"); return e != nullptr && e[0] == 'p'; }();
        const bool blocked = !plain_pairs && (d == 2u || d == 3u);
        if (blocked) { hipLaunchKernelGGL(k_dg_items<true>, ga, b, 0, ctx->stream, cnt_t, cnt_q, tot, goff, asize, items, dflags + 8, item_cap); }
        else { hipLaunchKernelGGL(k_dg_items<false>, ga, b, 0, ctx->stream, cnt_t, cnt_q, tot, goff, asize, items, dflags + 8, item_cap); }
        PairArgs p{};
        p.seqs = ctx->db.seqs; p.seq_off = ctx->db.seq_off; p.seqlen = ctx->db.seqlen;
        p.sigs = static_cast<const ulonglong2 *>(ctx->d_qgrams.ptr);
        p.members = members; p.items = items; p.item_count = dflags + 8; p.item_cap = item_cap; p.d = d; p.k = k; p.wlen = wlen;
        p.pairs = static_cast<unsigned long long *>(ctx->d_fpairs.ptr); p.counters = fc; p.pair_cap = ctx->dn_pair_cap;
        if (blocked && d == 2u) { hipLaunchKernelGGL(k_dg_pairs_lds<2>, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, p); }
        else if (blocked) { hipLaunchKernelGGL(k_dg_pairs_lds<3>, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, p); }
        else { hipLaunchKernelGGL(k_dg_pairs, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, p); }
        SWA_HIP(ctx, hipGetLastError());
        launches += 9;
      }
      uint64_t got[2] = {0, 0};
      uint32_t fl[2] = {0, 0};
      SWA_HIP(ctx, hipMemcpyAsync(got, fc, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
      SWA_HIP(ctx, hipMemcpyAsync(fl, dflags + 8, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
      SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (fl[1] != 0) { return swa_fail_msg(ctx, SWA_E_DEVICE, "swa_dn_graph: group key table overflow"); }
      ctx->dn_comparisons = got[1];
      if (got[0] <= ctx->dn_pair_cap) { npairs = got[0]; break; }
      if (attempt == 5) { return swa_fail_msg(ctx, SWA_E_NOMEM, "swa_dn_graph: pair list keeps overflowing"); }
      ctx->dn_pair_cap = got[0] + 1024;
    }
    swa_t1(ctx, 5);
    // alignments in the direction(s) the abundance rule allows
    swa_t0(ctx, 6);
    uint64_t nedges = 0;
    if (npairs != 0) {
      SWA_TRY(swa_reserve(ctx, ctx->d_scan_targets, 4 * npairs * sizeof(uint32_t)));        // queries | targets, both directions
      SWA_TRY(swa_reserve(ctx, ctx->d_scan_diffs, 2 * npairs * sizeof(uint32_t)));
      auto * wq = static_cast<uint32_t *>(ctx->d_scan_targets.ptr);
      auto * wt = wq + 2 * npairs;
      auto * wd = static_cast<uint32_t *>(ctx->d_scan_diffs.ptr);
      hipLaunchKernelGGL(k_dg_worklist, dim3(grid_for(ctx, npairs)), dim3(256), 0, ctx->stream,
                         static_cast<const unsigned long long *>(ctx->d_fpairs.ptr), npairs, ctx->db.abundance, no_cluster_breaking,
                         wq, wt, fc + 2);
      uint64_t extra = 0;
      SWA_HIP(ctx, hipMemcpyAsync(&extra, fc + 2, sizeof(extra), hipMemcpyDeviceToHost, ctx->stream));
      SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      const uint64_t nwork = npairs + extra;
      ctx->dn_aligned = nwork;
      // the wavefront kernel's work in the order of its expected length (k_dg_work_keys; SWA_DN_ALIGN_ORDER=0: as found)
      static const bool ordered = [] { const char * e = getenv("SWA_DN_ALIGN_ORDER"); return !(e != nullptr && e[0] == '0'); }();
      static const uint32_t align_len_weight = [] { const char * e = getenv("SWA_DN_ALIGN_LEN_WEIGHT"); return e != nullptr ? (uint32_t)atoi(e) : 10u; }();   // (a nucleotide of length difference counts as ten signature bits: alignments + CSR 6.9-7.1 -> 6.5-6.6 ms against 0; 5 and 20: 7.0 / 6.8)
      if (ordered && nwork > 1 && ctx->d_qgrams.ptr != nullptr) {
        SWA_TRY(swa_reserve(ctx, ctx->d_dn_keys, 2 * nwork * sizeof(uint64_t)));          // packed items: in | out (the edges' keys later)
        SWA_TRY(swa_reserve(ctx, ctx->d_dn_vals, 2 * nwork * sizeof(uint32_t)));          // keys: in | out (bytes; the edges' values later)
        auto * packed = static_cast<unsigned long long *>(ctx->d_dn_keys.ptr);
        auto * wkey = static_cast<unsigned char *>(ctx->d_dn_vals.ptr);
        hipLaunchKernelGGL(k_dg_work_keys, dim3(grid_for(ctx, nwork)), dim3(256), 0, ctx->stream, wq, wt, nwork,
                           static_cast<const ulonglong2 *>(ctx->d_qgrams.ptr), ctx->db.seqlen, align_len_weight, wkey, packed);
        size_t order_bytes = 0;
        (void)rocprim::radix_sort_pairs(nullptr, order_bytes, wkey, wkey + nwork, packed, packed + nwork, nwork, 0, 6, ctx->stream);
        SWA_TRY(swa_reserve(ctx, ctx->d_scan_hits, order_bytes + 16));
        SWA_HIP(ctx, rocprim::radix_sort_pairs(ctx->d_scan_hits.ptr, order_bytes, wkey, wkey + nwork, packed, packed + nwork, nwork, 0, 6, ctx->stream));
        hipLaunchKernelGGL(k_dg_work_unpack, dim3(grid_for(ctx, nwork)), dim3(256), 0, ctx->stream, packed + nwork, nwork, wq, wt);
        launches += 4;
      }
      // (the launcher takes 32-bit counts: in slices)
      for (uint64_t at = 0; at < nwork; at += 0x40000000ull) {
        const uint32_t cnt = (uint32_t)std::min<uint64_t>(0x40000000ull, nwork - at);
        SWA_TRY(swa_align_launch(ctx, 0, wq + at, wt + at, nullptr, cnt, wd + at, nullptr, nullptr));
        ++launches;
      }
      SWA_TRY(swa_reserve(ctx, ctx->d_dn_keys, 2 * nwork * sizeof(uint64_t)));
      SWA_TRY(swa_reserve(ctx, ctx->d_dn_vals, 2 * nwork * sizeof(uint32_t)));
      auto * ekeys = static_cast<unsigned long long *>(ctx->d_dn_keys.ptr);
      auto * evals = static_cast<uint32_t *>(ctx->d_dn_vals.ptr);
      hipLaunchKernelGGL(k_dg_edges, dim3(grid_for(ctx, nwork)), dim3(256), 0, ctx->stream, wq, wt, wd, nwork, d, ekeys, evals, fc + 3);
      SWA_HIP(ctx, hipMemcpyAsync(&nedges, fc + 3, sizeof(nedges), hipMemcpyDeviceToHost, ctx->stream));
      SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (nedges != 0) {
        size_t sort_bytes = 0;
        (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, ekeys, ekeys + nwork, evals, evals + nwork, nedges, 0, 64, ctx->stream);
        SWA_TRY(swa_reserve(ctx, ctx->d_scan_hits, sort_bytes + 16));
        SWA_HIP(ctx, rocprim::radix_sort_pairs(ctx->d_scan_hits.ptr, sort_bytes, ekeys, ekeys + nwork, evals, evals + nwork, nedges, 0, 64,
                                               ctx->stream));
        launches += 6;
      }
    }
    swa_t1(ctx, 6);
    ctx->dn_edges = nedges;
    ctx->dn_work = npairs != 0 ? ctx->dn_aligned : 0;
    ctx->dn_launches = launches + 4;
    ctx->dn_graph_ready = true;
    ctx->dn_graph_ncb = no_cluster_breaking != 0;
  }
  return SWA_OK;
}

// a sorted link list -> offsets / neighbours / diffs on the host (buffers and capacity protocol of swa_dn_graph)
int swa_dn_graph_emit(swa_ctx * ctx, const unsigned long long * sorted, const uint32_t * svals, uint64_t nedges, uint64_t * offsets,
                      uint32_t * neighbours, uint8_t * diffs, uint64_t cap, uint64_t * total) {
  const uint32_t n = ctx->db.n;
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  *total = nedges;
  ctx->csr_ready = false;                                   // (d_offsets_tmp / d_nb_tmp now hold this graph)
  SWA_TRY(swa_reserve(ctx, ctx->d_offsets_tmp, ((uint64_t)n + 1) * sizeof(uint64_t)));
  hipLaunchKernelGGL(k_dg_offsets, dim3(grid_for(ctx, (uint64_t)n + 1)), dim3(256), 0, ctx->stream, sorted, nedges, n,
                     static_cast<uint64_t *>(ctx->d_offsets_tmp.ptr));
  SWA_HIP(ctx, hipMemcpyAsync(offsets, ctx->d_offsets_tmp.ptr, ((uint64_t)n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  if (nedges > cap) {
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return swa_fail_msg(ctx, SWA_E_CAPACITY, "swa_dn_graph: neighbour buffer too small");
  }
  if (nedges != 0) {
    SWA_TRY(swa_reserve(ctx, ctx->d_nb_tmp, nedges * (sizeof(uint32_t) + 1)));
    auto * nb32 = static_cast<uint32_t *>(ctx->d_nb_tmp.ptr);
    auto * df8 = reinterpret_cast<uint8_t *>(nb32 + nedges);
    hipLaunchKernelGGL(k_dg_split, dim3(grid_for(ctx, nedges)), dim3(256), 0, ctx->stream, sorted, svals, nedges, nb32, df8);
    SWA_HIP(ctx, hipMemcpyAsync(neighbours, nb32, nedges * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipMemcpyAsync(diffs, df8, nedges, hipMemcpyDeviceToHost, ctx->stream));
  }
  SWA_HIP(ctx, hipGetLastError());
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}

extern "C" int swa_dn_graph(swa_ctx * ctx, int no_cluster_breaking, uint64_t * offsets, uint32_t * neighbours, uint8_t * diffs,
                            uint64_t cap, uint64_t * total) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (offsets == nullptr || total == nullptr || (cap != 0 && (neighbours == nullptr || diffs == nullptr))) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_dn_graph: null buffer");
  }
  SWA_TRY(swa_dn_graph_compute(ctx, no_cluster_breaking));
  const unsigned long long * sorted = ctx->dn_work != 0 ? static_cast<const unsigned long long *>(ctx->d_dn_keys.ptr) + ctx->dn_work : nullptr;
  const uint32_t * svals = ctx->dn_work != 0 ? static_cast<const uint32_t *>(ctx->d_dn_vals.ptr) + ctx->dn_work : nullptr;
  return swa_dn_graph_emit(ctx, sorted, svals, ctx->dn_edges, offsets, neighbours, diffs, cap, total);
}

// The graph left in HBM as the resident network (offsets | neighbours | diffs in the buffers swa_d1_network_resident uses):
// swa_d1_cluster_device then evaluates the agglomeration of src/algo.cc:384-602 on it where it lies — the same pure
// function of the directed graph as for d = 1 (seeds by lowest id, generations by distance, members by generation then
// id: find_correct_position_in_list, src/algo.cc:205-219) — and swa_dn_parent_diffs adds what the radius needs.
extern "C" int swa_dn_graph_resident(swa_ctx * ctx, int no_cluster_breaking, uint64_t * total) {
  if (ctx == nullptr || total == nullptr) { return SWA_E_ARG; }
  SWA_TRY(swa_dn_graph_compute(ctx, no_cluster_breaking));
  const uint32_t n = ctx->db.n;
  const uint64_t nedges = ctx->dn_edges;
  const unsigned long long * sorted = ctx->dn_work != 0 ? static_cast<const unsigned long long *>(ctx->d_dn_keys.ptr) + ctx->dn_work : nullptr;
  const uint32_t * svals = ctx->dn_work != 0 ? static_cast<const uint32_t *>(ctx->d_dn_vals.ptr) + ctx->dn_work : nullptr;
  ctx->csr_ready = false;
  SWA_TRY(swa_reserve(ctx, ctx->d_offsets_tmp, ((uint64_t)n + 1) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_nb_tmp, (nedges + 1) * (sizeof(uint32_t) + 1)));
  hipLaunchKernelGGL(k_dg_offsets, dim3(grid_for(ctx, (uint64_t)n + 1)), dim3(256), 0, ctx->stream, sorted, nedges, n,
                     static_cast<uint64_t *>(ctx->d_offsets_tmp.ptr));
  if (nedges != 0) {
    auto * nb32 = static_cast<uint32_t *>(ctx->d_nb_tmp.ptr);
    hipLaunchKernelGGL(k_dg_split, dim3(grid_for(ctx, nedges)), dim3(256), 0, ctx->stream, sorted, svals, nedges, nb32,
                       reinterpret_cast<uint8_t *>(nb32 + nedges));
  }
  SWA_HIP(ctx, hipGetLastError());
  ctx->csr_ready = true;
  ctx->csr_total = nedges;
  ctx->csr_has_diffs = true;
  *total = nedges;
  return SWA_OK;
}

// pdiff[v] = differences between v and its parent in the clustering swa_d1_cluster_device has just made of the graph
// (0 for seeds); n bytes on the host.  radius(v) = radius(parent) + pdiff[v] (src/algo.cc:560-574).
extern "C" int swa_dn_parent_diffs(swa_ctx * ctx, uint8_t * pdiff) {
  if (ctx == nullptr || pdiff == nullptr) { return SWA_E_ARG; }
  if (!ctx->csr_ready || !ctx->csr_has_diffs || ctx->d_cluster.ptr == nullptr) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_dn_parent_diffs: call swa_dn_graph_resident and swa_d1_cluster_device first");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->db.n;
  const auto * nb32 = static_cast<const uint32_t *>(ctx->d_nb_tmp.ptr);
  const auto * parent = static_cast<const uint32_t *>(ctx->d_cluster.ptr) + 2ull * n;        // (layout of swa_d1_cluster_device: label | gen | parent)
  SWA_TRY(swa_reserve(ctx, ctx->d_scan_diffs, (uint64_t)n + 16));
  auto * out = static_cast<uint8_t *>(ctx->d_scan_diffs.ptr);
  hipLaunchKernelGGL(k_dg_parent_diffs, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, static_cast<const uint64_t *>(ctx->d_offsets_tmp.ptr),
                     nb32, reinterpret_cast<const uint8_t *>(nb32 + ctx->csr_total), parent, n, out);
  SWA_HIP(ctx, hipGetLastError());
  SWA_HIP(ctx, hipMemcpyAsync(pdiff, out, n, hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}

// Multi-GPU by ownership of window groups (as swa_d1_set_ownership): with world > 1 this context makes only the groups
// whose window key maps to `rank`.  A pair is reported through the FIRST window it shares ("not already found through
// an earlier window" is decided on the two sequences), and that window's group lives on one rank: over all ranks every
// pair of the graph is found exactly once.  world = 1 restores the complete graph.
extern "C" int swa_dn_set_ownership(swa_ctx * ctx, uint32_t rank, uint32_t world) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (world == 0 || rank >= world) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_dn_set_ownership: bad rank / world"); }
  if (rank != ctx->dn_owner_rank || world != ctx->dn_owner_world) {
    ctx->dn_owner_rank = rank; ctx->dn_owner_world = world;
    ctx->dn_graph_ready = false;
  }
  return SWA_OK;
}

// out3 = {q-gram comparisons, aligned pairs, kernel launches} of the last swa_dn_graph
extern "C" int swa_dn_graph_totals(swa_ctx * ctx, uint64_t * out3) {
  if (ctx == nullptr || out3 == nullptr) { return SWA_E_ARG; }
  out3[0] = ctx->dn_comparisons; out3[1] = ctx->dn_aligned; out3[2] = ctx->dn_launches;
  return SWA_OK;
}

// loads this translation unit's code object (see swa_ctx_warmup): an empty launch
__global__ void k_warm_dn() {}
void swa_warm_dn(swa_ctx * ctx) { hipLaunchKernelGGL(k_warm_dn, dim3(1), dim3(64), 0, ctx->stream); }
