// align.hip — seam B4: one-vs-many alignment scan on gfx950.
//
// Replaces search_do -> search8/search16 -> backtrack<> (src/scan.cc:221-256,
// src/search8.cc:629-903, src/search16.cc:385-678, src/utils/backtrack.h:51-138): for a
// query amplicon and a list of targets, the number of non-identical columns ("diff") of the
// reference's tie-broken minimum-cost global alignment with affine gaps.
//
// The reference fills a full qlen x dlen matrix in 16 (or 8) SIMD channels, stores four
// direction bits per cell and walks them backwards.  This kernel is built differently:
//
//   * forward-only: besides H/E/F each cell carries three small counters A_M, A_I, A_D = the
//     number of non-identical columns the reference's backtrack WOULD collect from that cell
//     to the origin when it arrives in state match / continuing-insertion / continuing-
//     deletion.  They obey local recurrences driven by the very comparisons the reference
//     records as direction bits (nw.cc:91-103, backtrack priority nw.cc:139-172), so no
//     direction matrix and no serial backtrack exist; diff = A_M(last cell).
//   * banded: a pair can only be accepted if diff <= d, hence cost <= T = d*max(mismatch,
//     gapopen+gapextend); every value that can influence the accepted path is < T + gapopen,
//     which confines the computation to |row - column| <= W = floor(T / gapextend) + 1.
//     Out-of-band inputs read as the saturation value.  Values saturate at 255 / 65535 like
//     the reference's v_add8 / v_add16 (src/utils/intrinsics_to_functions_x86_64.cc:103-125).
//     For pairs the reference would accept the result is bit-identical; any other pair gets
//     some value > d (proof sketch in DESIGN.md, exhaustive parity in tests/).
//   * anti-diagonal wavefront: G = 32 or 64 lanes per (query, target) pair, lane <-> band
//     offset (column - row); a cell needs its own lane's value from two steps ago (diagonal)
//     and its two neighbour lanes' values from the previous step, so each step is one
//     shuffle up + one shuffle down and pure integer VALU work.  Sequences sit 2-bit packed
//     in LDS.
#include "swa_internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

struct AlignArgs {
  const uint64_t * seqs;
  const uint64_t * seq_off;
  const uint32_t * seqlen;
  uint32_t query;             // the query amplicon, unless `queries` gives one per pair
  const uint32_t * queries;
  uint32_t ntargets;          // upper bound; the real count is *ntargets_dev when that is set
  const uint32_t * ntargets_dev;
  const uint32_t * targets;
  uint32_t * scores;        // may be null
  uint32_t * diffs;
  uint32_t * alnlens;       // may be null
  uint32_t mismatch, gapopen, gapextend;
  uint32_t sat;             // 255 or 65535
  int W;                    // band half-width
  uint32_t maxwords;        // words per staged sequence
};

__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b, uint32_t sat) {
  const uint32_t s = a + b;
  return s < sat ? s : sat;
}

// c ? x : y as a bit-field insert on an all-ones / all-zeros mask: never turned into a branch
__device__ __forceinline__ uint32_t select_u32(bool c, uint32_t x, uint32_t y) {
  const uint32_t m = 0u - (uint32_t)c;
  return (x & m) | (y & ~m);
}

// value of `v` in the neighbouring lane (full-wave DPP shifts, GFX9 family): a few cycles
// instead of an LDS-crossbar ds_bpermute on the critical path of every anti-diagonal step
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v) {   // lane i <- lane i-1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t from_lane_above(uint32_t v) {   // lane i <- lane i+1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

// Lane layout of one pair's group of G lanes: lane 0 and the lanes above 2W+1 are GUARD lanes,
// lanes 1 .. 2W+1 carry the band offsets -W .. +W.  A guard lane never computes a cell, so what
// its neighbours read from it is forever the "outside the band" value — no per-step select is
// needed at the band edges, and the first / last lane of the wave (whose DPP shift has no
// source) are guards as well.  Needs 2W + 3 <= G.
template <int G, bool WANT_LEN>
__global__ __launch_bounds__(256) void k_align(const AlignArgs a) {
  extern __shared__ uint64_t lds[];
  constexpr int kGroups = 256 / G;
  const int group = threadIdx.x / G;
  const int t = threadIdx.x % G;
  uint64_t * qw = lds + (size_t)(2 * group) * a.maxwords;     // this group's query words
  uint64_t * dw = qw + a.maxwords;                            // this group's target words

  const uint32_t SAT = a.sat;
  const int W = a.W;
  const int o = t - 1 - W;                                    // band offset = column - row
  const bool lane_in_band = t >= 1 && t <= 2 * W + 1;
  const uint32_t go = a.gapopen, ge = a.gapextend, mm = a.mismatch;
  constexpr uint32_t kBigCount = 0xFFFFu;
  const uint32_t kOutside = SAT | (kBigCount << 16);          // what an out-of-band neighbour provides

  const uint32_t ntargets = a.ntargets_dev != nullptr ? min(*a.ntargets_dev, a.ntargets) : a.ntargets;
  for (uint32_t pair = blockIdx.x * kGroups + group; pair < ntargets; pair += gridDim.x * kGroups) {
    const uint32_t target = a.targets[pair];
    const uint32_t query = a.queries != nullptr ? a.queries[pair] : a.query;
    const uint32_t dl = a.seqlen[target];
    const uint32_t ql = a.seqlen[query];
    {
      const uint64_t * gd = a.seqs + a.seq_off[target];
      const uint64_t * gq = a.seqs + a.seq_off[query];
      const uint32_t dn = (dl + 31u) >> 5;
      const uint32_t qn = (ql + 31u) >> 5;
      for (uint32_t w = t; w < dn; w += G) { dw[w] = gd[w]; }
      for (uint32_t w = t; w < qn; w += G) { qw[w] = gq[w]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const int delta = (int)ql - (int)dl;
    const bool feasible = (delta <= W) && (-delta <= W);      // the end cell lies inside the band
    // Per-lane state = the last cell this lane computed.  The values handed to the neighbour
    // lanes travel packed: low 16 bits the (saturated) score, high 16 bits the diff counter.
    // The loop body is branch-free, and the nucleotide comparison of the NEXT step is fetched
    // from LDS while the current step computes.
    uint32_t Hown = 0, AMown = 0, LMown = 0;                  // H, A_M, alignment length (diagonal input)
    uint32_t dn_pk = kOutside, dn_len = 0;                    // E | A_I  passed down  to (r+1, c)
    uint32_t rt_pk = kOutside, rt_len = 0;                    // F | A_D  passed right to (r, c+1)
    if (feasible) {
      const int last = (int)dl + (int)ql - 2;
      const int rmax = (int)dl - 1, cmax = (int)ql - 1;
      auto mismatch_at = [&](int s) -> uint32_t {             // d[r] != q[c] for this lane's cell of step s
        const int rs = s - o;
        int r = rs >> 1;
        int c = s - r;
        r = r < 0 ? 0 : (r > rmax ? rmax : r);                // clamped: inactive lanes read something valid
        c = c < 0 ? 0 : (c > cmax ? cmax : c);
        const uint32_t dnt = (uint32_t)(dw[r >> 5] >> ((r & 31) << 1));
        const uint32_t qnt = (uint32_t)(qw[c >> 5] >> ((c & 31) << 1));
        return ((dnt ^ qnt) & 3u) != 0u ? 1u : 0u;
      };
      // interior steps: no clamping needed — in-band lanes stay inside the matrix (one word of
      // slack past the last row / column is staged for the prefetch of the first EDGE step after
      // the interior), guard lanes borrow offset 0 so that they read something valid too
      const int o_safe = lane_in_band ? o : 0;
      // A lane computes a cell every OTHER step, so inside the interior one comparison serves a
      // pair of steps (s even, s + 1): the lane's own active step of the pair is s + ((o ^ s) & 1).
      auto mismatch_pair = [&](int s_even) -> uint32_t {
        const int r = (s_even + (o_safe & 1) - o_safe) >> 1;
        const int c = s_even + (o_safe & 1) - r;
        const uint32_t dnt = (uint32_t)(dw[r >> 5] >> ((r & 31) << 1));
        const uint32_t qnt = (uint32_t)(qw[c >> 5] >> ((c & 31) << 1));
        return ((dnt ^ qnt) & 3u) != 0u ? 1u : 0u;
      };
      const bool act_even = lane_in_band && ((o & 1) == 0);  // active on even steps
      const bool act_odd = lane_in_band && ((o & 1) != 0);
      uint32_t mis_next = mismatch_at(0);
      // One anti-diagonal step.  EDGE = the step may touch row 0 / column 0 (nw.cc:66-79) or the
      // last row / column; interior steps (the bulk) need neither the boundary inputs, which are
      // wave-uniform functions of the step, nor the range tests: a lane is active iff it is in
      // the band and (s - o) is even.
      auto step = [&](int s, auto edge_tag, uint32_t mis_pair, bool act_inner) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        uint32_t mis = mis_pair;
        if (EDGE) {
          mis = mis_next;
          mis_next = mismatch_at(s + 1);                      // LDS latency hides behind this step's math
        }
        // neighbour values from the previous step (every lane of the group takes part)
        const uint32_t above_pk = from_lane_above(dn_pk);     // cell (r-1, c) lives in lane t+1
        const uint32_t below_pk = from_lane_below(rt_pk);     // cell (r, c-1) lives in lane t-1
        uint32_t above_len = 0, below_len = 0;
        if (WANT_LEN) { above_len = from_lane_above(dn_len); below_len = from_lane_below(rt_len); }
        const int rs = s - o;
        bool act = EDGE ? (lane_in_band && ((rs & 1) == 0)) : act_inner;   // in the band and (s - o) even
        uint32_t hd = Hown, amd = AMown, left_pk = above_pk, top_pk = below_pk;   // "left" / "top" as in nw.cc
        bool row0 = false, col0 = false;
        uint32_t su = (uint32_t)s;
        if (EDGE) {
          const int r = rs >> 1;
          const int c = s - r;
          act = act && rs >= 0 && r <= rmax && c >= 0 && c <= cmax;
          row0 = rs == 0;                                      // r == 0  (then c == s)
          col0 = s + o == 0;                                   // c == 0  (then r == s)
          const uint32_t edge_h = s == 0 ? 0u : sat_add(go, su * ge, SAT);
          const uint32_t edge_pk = sat_add(2u * go, (su + 2u) * ge, SAT) | ((su + 1u) << 16);
          hd = (row0 || col0) ? edge_h : Hown;
          amd = (row0 || col0) ? su : AMown;
          left_pk = row0 ? edge_pk : above_pk;
          top_pk = col0 ? edge_pk : below_pk;
        }
        const uint32_t left = left_pk & 0xFFFFu, aiv = left_pk >> 16;
        const uint32_t top = top_pk & 0xFFFFu, adh = top_pk >> 16;

        const uint32_t dp = sat_add(hd, mis ? mm : 0u, SAT);
        const bool up = top < dp;                              // nw.cc:91
        uint32_t h = dp < top ? dp : top;
        h = h < left ? h : left;
        const bool lb = left == h;                             // nw.cc:94
        const uint32_t d2 = sat_add(h, go + ge, SAT);
        const uint32_t l2 = sat_add(left, ge, SAT);
        const uint32_t t2 = sat_add(top, ge, SAT);
        const bool eu = t2 < d2;                               // nw.cc:102
        const bool el = l2 < d2;                               // nw.cc:103
        // what the backtrack would count from here (priority: nw.cc:139-172)
        // (all candidates are computed, then selected: no divergent control flow in the loop)
        const uint32_t via_l = aiv + 1u, via_t = adh + 1u, via_d = amd + mis;
        uint32_t am = select_u32(up, via_t, via_d);
        am = select_u32(lb, via_l, am);
        am = am < kBigCount ? am : kBigCount;
        uint32_t ai = select_u32(el, via_l, am);
        uint32_t ad = select_u32(eu, via_t, am);
        ai = ai < kBigCount ? ai : kBigCount;
        ad = ad < kBigCount ? ad : kBigCount;
        Hown = act ? h : Hown;
        AMown = act ? am : AMown;
        dn_pk = act ? ((d2 < l2 ? d2 : l2) | (ai << 16)) : dn_pk;
        rt_pk = act ? ((d2 < t2 ? d2 : t2) | (ad << 16)) : rt_pk;
        if (WANT_LEN) {
          const uint32_t lmd = (row0 || col0) ? su : LMown;
          const uint32_t liv = row0 ? su + 1u : above_len;
          const uint32_t ldh = col0 ? su + 1u : below_len;
          const uint32_t lm = lb ? liv + 1u : (up ? ldh + 1u : lmd + 1u);
          LMown = act ? lm : LMown;
          dn_len = act ? (el ? liv + 1u : lm) : dn_len;
          rt_len = act ? (eu ? ldh + 1u : lm) : rt_len;
        }
      };
      // interior steps: W < s <= 2 * min(rmax, cmax) - W  (every in-band cell of the step is
      // inside the matrix and off its first row / column)
      const int in_lo = W + 1;
      const int in_hi = 2 * (rmax < cmax ? rmax : cmax) - W;
      int s = 0;
      for (; s <= last && (s < in_lo || (s & 1) != 0); ++s) { step(s, std::true_type{}, 0u, false); }
      if (s + 1 <= in_hi && s + 1 <= last) {                 // interior, two steps per turn, s even
        uint32_t mis_pair = mismatch_pair(s);
        for (; s + 1 <= in_hi && s + 1 <= last; s += 2) {
          const uint32_t mis_now = mis_pair;
          mis_pair = mismatch_pair(s + 2);                    // (reads at most one word past the sequences)
          step(s, std::false_type{}, mis_now, act_even);
          step(s + 1, std::false_type{}, mis_now, act_odd);
        }
        mis_next = mismatch_at(s);                            // hand over to the generic prefetch chain
      }
      for (; s <= last; ++s) { step(s, std::true_type{}, 0u, false); }
    }
    // the lane whose offset equals ql - dl holds the end cell
    if (!feasible) {
      if (t == 0) {
        a.diffs[pair] = SAT;
        if (a.scores != nullptr) { a.scores[pair] = SAT; }
        if (a.alnlens != nullptr) { a.alnlens[pair] = 0; }
      }
    } else if (lane_in_band && o == delta) {
      // like the reference: a saturated score means "no alignment", diff = SAT (search8.cc:776-810)
      const bool overflow = Hown >= SAT;
      a.diffs[pair] = overflow ? SAT : AMown;
      if (a.scores != nullptr) { a.scores[pair] = Hown; }
      if (a.alnlens != nullptr) { a.alnlens[pair] = overflow ? 0 : LMown; }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- wavefront formulation (the fast path) -------------------------------------------------
// For near-identical pairs — the only ones the caller keeps — filling even a banded matrix is
// wasteful: ~(Lq + Lt) dependent anti-diagonal steps.  The wavefront recurrences of gap-affine
// alignment (furthest-reaching point per diagonal and per score; Marco-Sola et al. 2021) reach
// the optimal cost in one step per REACHABLE cost value <= T plus ~L/32 word comparisons.
// They give the optimal cost, not the reference's tie-broken path.  That is enough whenever
// every cost <= T can be written in exactly one way as a*mismatch + g*gapopen + e*gapextend
// up to (a + e, e): then all optimal alignments of a pair have the same number of
// non-identical columns (a + e = diff) and of columns ((Lq + Lt + e) / 2), whichever one the
// reference's tie-breaking picks.  swa_search_begin checks this for the penalties and d in use
// (true for the default 18/24/13 up to d = 5) and the banded kernel above remains the
// general path.
struct swa_wfa_step {
  uint32_t score;           // a reachable cost value, ascending
  int32_t from_x;           // index of score - mismatch in the list, or -1
  int32_t from_oe;          //          score - gapopen - gapextend
  int32_t from_e;           //          score - gapextend
  uint32_t diff;            // a + e of the (unique) decomposition
  uint32_t gapcols;         // e
};

constexpr int kWfaMaxSteps = 64;

struct WfaArgs {
  AlignArgs a;
  const swa_wfa_step * steps;
  uint32_t nsteps;
  uint32_t ring;                  // steps of history kept: a step reads at most ring - 1 steps back
};

// 32 nucleotides of `seq` starting at position p (2 bits each, LSB first)
__device__ __forceinline__ uint64_t window_at(const uint64_t * seq, uint32_t p) {
  const uint32_t wi = p >> 5, sh = (p & 31u) << 1;
  const uint64_t lo = seq[wi], hi = seq[wi + 1u];
  return sh != 0u ? (lo >> sh) | (hi << (64u - sh)) : lo;
}

// Lanes: as in k_align, lane t of a 32-lane group owns diagonal k = t - 1 - W (column - row),
// lanes 0 and > 2W+1 are guards that stay invalid.  Offsets are columns (query positions).
// G = lanes of a pair's group: 32, or — round 6 — 16 where the band fits (2 W + 3 <= 16: the default penalties up to d = 3
// with the band cut to the diagonals a score <= T can reach): four pairs a wave instead of two.  The kernel waits on
// dependent LDS round trips of a few active lanes; what a wave gets done per round trip is its pairs.
template <int G>
__global__ __launch_bounds__(128) void k_align_wfa(const WfaArgs w) {
  extern __shared__ uint64_t lds[];
  constexpr int kGroups = 128 / G;
  const AlignArgs & a = w.a;
  const int group = threadIdx.x / G;
  const int t = threadIdx.x % G;
  uint64_t * qw = lds + (size_t)(2 * group) * a.maxwords;
  uint64_t * dw = qw + a.maxwords;
  // wavefront history: [step][M | I | D][lane], offset + 1 as u16 (0 = invalid)
  // (a RING of the steps a step can look back to — score - mismatch, - gap extension, - gap opening - extension are a few
  // list entries away —, not room for the most steps a scoring may have: 49 KB a workgroup and three workgroups a CU
  // became a few KB and as many waves as a CU holds; the kernel waits on dependent LDS round trips, so the waves in flight
  // are its throughput: 21.2 -> 11.7 ms with the history sized by the steps in use, r05)
  uint16_t * hist = reinterpret_cast<uint16_t *>(lds + (size_t)(2 * kGroups) * a.maxwords) +
                    (size_t)group * ((size_t)w.ring * 3 * G);
  const uint32_t ring = w.ring;
  const int W = a.W;
  const int k = t - 1 - W;
  const bool lane_in_band = t >= 1 && t <= 2 * W + 1;
  const uint32_t ntargets = a.ntargets_dev != nullptr ? min(*a.ntargets_dev, a.ntargets) : a.ntargets;
  for (uint32_t base = blockIdx.x * kGroups; base < ntargets; base += gridDim.x * kGroups) {
    const uint32_t pair = base + group;
    const bool have = pair < ntargets;
    const uint32_t target = have ? a.targets[pair] : 0u;
    const uint32_t query = have ? (a.queries != nullptr ? a.queries[pair] : a.query) : 0u;
    const int dl = have ? (int)a.seqlen[target] : 0;         // rows (pattern)
    const int ql = have ? (int)a.seqlen[query] : 0;          // columns (text)
    if (have) {
      const uint64_t * gd = a.seqs + a.seq_off[target];
      const uint64_t * gq = a.seqs + a.seq_off[query];
      const uint32_t dn = ((uint32_t)dl + 31u) >> 5;
      const uint32_t qn = ((uint32_t)ql + 31u) >> 5;
      for (uint32_t x = t; x < dn + 1u; x += G) { dw[x] = x < dn ? gd[x] : 0ull; }
      for (uint32_t x = t; x < qn + 1u; x += G) { qw[x] = x < qn ? gq[x] : 0ull; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const int kend = ql - dl;
    const bool feasible = have && kend <= W && -kend <= W;
    int found = -1;                                          // step index at which the end was reached
    bool running = feasible;
    for (uint32_t i = 0; i < w.nsteps; ++i) {
      if (__ballot(running) == 0ull) { break; }
      const swa_wfa_step st = w.steps[i];
      auto load = [&](int from, int which) -> int {
        return from >= 0 ? (int)hist[((size_t)((uint32_t)from % ring) * 3 + which) * G + t] - 1 : -1;
      };
      const int m_x = load(st.from_x, 0);                     // own diagonal
      const int m_oe = load(st.from_oe, 0);
      const int i_e = load(st.from_e, 1);
      const int d_e = load(st.from_e, 2);
      // insertion: from diagonal k-1 (lane t-1), consumes one query column
      const int m_oe_below = (int)from_lane_below((uint32_t)m_oe), i_e_below = (int)from_lane_below((uint32_t)i_e);
      // deletion: from diagonal k+1 (lane t+1), consumes one target row
      const int m_oe_above = (int)from_lane_above((uint32_t)m_oe), d_e_above = (int)from_lane_above((uint32_t)d_e);
      int I = (m_oe_below > i_e_below ? m_oe_below : i_e_below);
      I = (I >= 0 && I + 1 <= ql) ? I + 1 : -1;               // row = I - k stays what it was: valid
      int D = (m_oe_above > d_e_above ? m_oe_above : d_e_above);
      D = (D >= 0 && D - k <= dl) ? D : -1;                   // same column, one more row
      int M = (m_x >= 0 && m_x + 1 <= ql && m_x + 1 - k <= dl) ? m_x + 1 : -1;
      M = M > I ? M : I;
      M = M > D ? M : D;
      if (i == 0u) { M = (k == 0) ? 0 : -1; }                // score 0 starts at the origin
      if (!lane_in_band || !running) { M = -1; I = -1; D = -1; }
      // extend M along the diagonal while the sequences agree, 32 nucleotides per turn
      bool ext = M >= 0;
      while (__ballot(ext) != 0ull) {
        if (ext) {
          const int r = M - k;
          const int rem = min(ql - M, dl - r);
          if (rem <= 0) { ext = false; }
          else {
            const uint64_t x = window_at(qw, (uint32_t)M) ^ window_at(dw, (uint32_t)r);
            int n = x != 0ull ? (__ffsll((unsigned long long)x) - 1) >> 1 : 32;
            n = n < rem ? n : rem;
            M += n;
            ext = n == 32;
          }
        }
      }
      hist[((size_t)(i % ring) * 3 + 0) * G + t] = (uint16_t)(M + 1);
      hist[((size_t)(i % ring) * 3 + 1) * G + t] = (uint16_t)(I + 1);
      hist[((size_t)(i % ring) * 3 + 2) * G + t] = (uint16_t)(D + 1);
      // finished when the end diagonal's furthest point is the last column (then row = dl too)
      const bool at_end = running && lane_in_band && k == kend && M == ql;
      const uint64_t endmask = __ballot(at_end);
      // (a 64-lane wave holds 64 / G groups: look at this group's lanes only)
      const uint64_t mine = (endmask >> ((threadIdx.x & 63u) & ~(uint32_t)(G - 1))) & ((1ull << G) - 1ull);
      if (running && mine != 0ull) { found = (int)i; running = false; }
      // (a lane only ever reads back the history it wrote itself: no synchronisation needed)
    }
    if (have && t == 0) {
      uint32_t diff = a.sat, score = a.sat, alen = 0;
      if (found >= 0) {
        const swa_wfa_step st = w.steps[found];
        if (st.score < a.sat) {                               // a saturated score means "no alignment" (search8.cc:776-810)
          diff = st.diff;
          score = st.score;
          alen = ((uint32_t)(ql + dl) + st.gapcols) >> 1;
        }
      }
      a.diffs[pair] = diff;
      if (a.scores != nullptr) { a.scores[pair] = score; }
      if (a.alnlens != nullptr) { a.alnlens[pair] = alen; }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Generic fallback for bands wider than 64 lanes (large d or unusual penalties): one thread
// per pair walks the band row by row with the previous row kept in an interleaved global
// scratch (column-major across threads, so a wave's accesses coalesce).  Same recurrences,
// same saturation, same result; only slower.
__global__ __launch_bounds__(256) void k_align_generic(const AlignArgs a, uint32_t * __restrict__ scratch,
                                                       uint32_t nthreads, uint32_t qcap) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= nthreads) { return; }
  const uint32_t SAT = a.sat;
  const int W = a.W;
  const uint32_t go = a.gapopen, ge = a.gapextend, mm = a.mismatch;
  constexpr uint32_t kBigCount = 0xFFFFu;
  // six planes of qcap columns each: H, E, A_M, A_I, L_M, L_I of the previous row
  uint32_t * pH = scratch;
  uint32_t * pE = pH + (size_t)qcap * nthreads;
  uint32_t * pAM = pE + (size_t)qcap * nthreads;
  uint32_t * pAI = pAM + (size_t)qcap * nthreads;
  uint32_t * pLM = pAI + (size_t)qcap * nthreads;
  uint32_t * pLI = pLM + (size_t)qcap * nthreads;
  const uint32_t ntargets = a.ntargets_dev != nullptr ? min(*a.ntargets_dev, a.ntargets) : a.ntargets;
  for (uint32_t pair = tid; pair < ntargets; pair += nthreads) {
    const uint32_t target = a.targets[pair];
    const uint32_t query = a.queries != nullptr ? a.queries[pair] : a.query;
    const uint32_t ql = a.seqlen[query];
    const uint64_t * qw = a.seqs + a.seq_off[query];
    const uint32_t dl = a.seqlen[target];
    const uint64_t * dw = a.seqs + a.seq_off[target];
    const int delta = (int)ql - (int)dl;
    uint32_t endH = SAT, endAM = kBigCount, endLM = 0;
    const bool feasible = (delta <= W) && (-delta <= W);
    if (feasible) {
      for (int r = 0; r < (int)dl; ++r) {
        const uint32_t dnt = (uint32_t)((dw[r >> 5] >> ((r & 31) << 1)) & 3u);
        const int c0 = r - W > 0 ? r - W : 0;
        const int c1 = r + W < (int)ql - 1 ? r + W : (int)ql - 1;
        // running horizontal state and the diagonal inputs for column c0
        uint32_t top, adh, ldh, hd, amd, lmd;
        if (c0 == 0) {
          top = sat_add(2u * go, (uint32_t)(r + 2) * ge, SAT); adh = (uint32_t)r + 1u; ldh = (uint32_t)r + 1u;
          hd = (r == 0) ? 0u : sat_add(go, (uint32_t)r * ge, SAT); amd = (uint32_t)r; lmd = (uint32_t)r;
        } else {
          top = SAT; adh = kBigCount; ldh = 0u;                  // (r, c0-1) lies outside the band
          const size_t at = (size_t)(c0 - 1) * nthreads + tid;   // (r-1, c0-1): same offset, inside
          hd = pH[at]; amd = pAM[at]; lmd = pLM[at];
        }
        for (int c = c0; c <= c1; ++c) {
          const size_t at = (size_t)c * nthreads + tid;
          uint32_t left, aiv, liv;
          if (r == 0) { left = sat_add(2u * go, (uint32_t)(c + 2) * ge, SAT); aiv = (uint32_t)c + 1u; liv = (uint32_t)c + 1u; }
          else if (c <= r - 1 + W) { left = pE[at]; aiv = pAI[at]; liv = pLI[at]; }
          else { left = SAT; aiv = kBigCount; liv = 0u; }
          // the diagonal input of the NEXT column is this column's previous-row value
          uint32_t nhd, namd, nlmd;
          if (r == 0) { nhd = sat_add(go, (uint32_t)(c + 1) * ge, SAT); namd = (uint32_t)c + 1u; nlmd = (uint32_t)c + 1u; }
          else { nhd = pH[at]; namd = pAM[at]; nlmd = pLM[at]; }
          const uint32_t qnt = (uint32_t)((qw[c >> 5] >> ((c & 31) << 1)) & 3u);
          const uint32_t mis = dnt != qnt ? 1u : 0u;
          const uint32_t dp = sat_add(hd, mis ? mm : 0u, SAT);
          const bool up = top < dp;
          uint32_t h = dp < top ? dp : top;
          h = h < left ? h : left;
          const bool lb = left == h;
          const uint32_t d2 = sat_add(h, go + ge, SAT);
          const uint32_t l2 = sat_add(left, ge, SAT);
          const uint32_t t2 = sat_add(top, ge, SAT);
          const bool eu = t2 < d2;
          const bool el = l2 < d2;
          uint32_t am, lm;
          if (lb) { am = aiv + 1u; lm = liv + 1u; }
          else if (up) { am = adh + 1u; lm = ldh + 1u; }
          else { am = amd + mis; lm = lmd + 1u; }
          if (am > kBigCount) { am = kBigCount; }
          uint32_t ai = el ? aiv + 1u : am;
          uint32_t ad = eu ? adh + 1u : am;
          if (ai > kBigCount) { ai = kBigCount; }
          if (ad > kBigCount) { ad = kBigCount; }
          pH[at] = h; pAM[at] = am; pLM[at] = lm;
          pE[at] = d2 < l2 ? d2 : l2; pAI[at] = ai; pLI[at] = el ? liv + 1u : lm;
          top = d2 < t2 ? d2 : t2; adh = ad; ldh = eu ? ldh + 1u : lm;
          hd = nhd; amd = namd; lmd = nlmd;
          if (r == (int)dl - 1 && c == (int)ql - 1) { endH = h; endAM = am; endLM = lm; }
        }
      }
    }
    const bool overflow = !feasible || endH >= SAT;
    a.diffs[pair] = overflow ? SAT : endAM;
    if (a.scores != nullptr) { a.scores[pair] = feasible ? endH : SAT; }
    if (a.alnlens != nullptr) { a.alnlens[pair] = overflow ? 0 : endLM; }
  }
}

}  // namespace

extern "C" int swa_search_begin(swa_ctx * ctx, uint64_t mismatch, uint64_t gapopen, uint64_t gapextend, uint64_t d) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (mismatch == 0 || gapextend == 0 || mismatch > 65535 || gapopen > 65535 || gapextend > 65535 || d == 0 || d > 255) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_search_begin: penalties must be positive and d in 1..255");
  }
  ctx->pen_mismatch = mismatch;
  ctx->pen_gapopen = gapopen;
  ctx->pen_gapextend = gapextend;
  ctx->resolution = d;
  ctx->search_ready = true;
  ctx->dn_graph_ready = false;                               // (penalties / d define the graph of dn_graph.hip)
  // Wavefront fast path (k_align_wfa): usable when every cost <= T has one decomposition only
  // into (non-identical columns, gap columns) — see the comment above the kernel.
  ctx->wfa_steps = 0;
  {
    const uint64_t T = d * std::max<uint64_t>(mismatch, gapopen + gapextend);
    struct Way { uint64_t cost; uint32_t diff, gapcols; };
    std::vector<Way> ways;
    bool unique = T < (1u << 20);
    for (uint64_t a = 0; unique && a * mismatch <= T; ++a) {
      for (uint64_t g = 0; a * mismatch + g * (gapopen + gapextend) <= T; ++g) {
        for (uint64_t e = g; a * mismatch + g * gapopen + e * gapextend <= T; ++e) {
          ways.push_back({a * mismatch + g * gapopen + e * gapextend, (uint32_t)(a + e), (uint32_t)e});
          if (g == 0) { break; }                          // no gap: no gap columns
        }
        if (ways.size() > 100000) { unique = false; break; }
      }
    }
    std::sort(ways.begin(), ways.end(), [](const Way & x, const Way & y) {
      return x.cost != y.cost ? x.cost < y.cost : (x.diff != y.diff ? x.diff < y.diff : x.gapcols < y.gapcols);
    });
    std::vector<swa_wfa_step> steps;
    for (size_t i = 0; unique && i < ways.size(); ++i) {
      if (i > 0 && ways[i].cost == ways[i - 1].cost) {
        if (ways[i].diff != ways[i - 1].diff || ways[i].gapcols != ways[i - 1].gapcols) { unique = false; }
        continue;
      }
      swa_wfa_step st{};
      st.score = (uint32_t)ways[i].cost; st.diff = ways[i].diff; st.gapcols = ways[i].gapcols;
      steps.push_back(st);
    }
    if (unique && !steps.empty() && steps.size() <= (size_t)kWfaMaxSteps) {
      auto index_of = [&](int64_t cost) -> int32_t {
        if (cost < 0) { return -1; }
        for (size_t j = 0; j < steps.size(); ++j) { if (steps[j].score == (uint64_t)cost) { return (int32_t)j; } }
        return -1;
      };
      for (auto & st : steps) {
        st.from_x = index_of((int64_t)st.score - (int64_t)mismatch);
        st.from_oe = index_of((int64_t)st.score - (int64_t)(gapopen + gapextend));
        st.from_e = index_of((int64_t)st.score - (int64_t)gapextend);
      }
      SWA_HIP(ctx, hipSetDevice(ctx->device));
      SWA_TRY(swa_reserve(ctx, ctx->d_wfa, steps.size() * sizeof(swa_wfa_step)));
      SWA_HIP(ctx, hipMemcpyAsync(ctx->d_wfa.ptr, steps.data(), steps.size() * sizeof(swa_wfa_step), hipMemcpyHostToDevice, ctx->stream));
      SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      ctx->wfa_steps = (uint32_t)steps.size();
      // how far back (in steps) a step looks: its history is a ring of that many steps (k_align_wfa)
      uint32_t back = 1;
      for (size_t j = 0; j < steps.size(); ++j) {
        for (const int32_t from : {steps[j].from_x, steps[j].from_oe, steps[j].from_e}) { if (from >= 0) { back = std::max<uint32_t>(back, (uint32_t)(j - (size_t)from)); } }
      }
      ctx->wfa_ring = back + 1;
    }
  }
  return SWA_OK;
}

extern "C" int swa_search_uses_wavefront(const swa_ctx * ctx) {
  return ctx != nullptr && ctx->search_ready && ctx->wfa_steps != 0 ? 1 : 0;
}

namespace {

__global__ __launch_bounds__(256) void k_narrow(const uint64_t * __restrict__ in, uint32_t * __restrict__ out, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    out[i] = (uint32_t)in[i];
  }
}
__global__ __launch_bounds__(256) void k_widen(const uint32_t * __restrict__ in, uint64_t * __restrict__ out, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    out[i] = in[i];
  }
}

}  // namespace

// Enqueue the alignment of `query` against d_targets[0 .. count) on the context's stream.
// count = *d_count (device) when d_count != nullptr, bounded by max_count; d_queries != nullptr
// gives one query per pair (batched sub-seeds) instead of the single `query`; all pointers
// are device memory.  Shared by swa_search_do and the fused scan (scan.hip).
int swa_align_launch(swa_ctx * ctx, uint32_t query, const uint32_t * d_queries, const uint32_t * d_targets,
                     const uint32_t * d_count, uint32_t max_count, uint32_t * d_diffs, uint32_t * d_scores,
                     uint32_t * d_alnlens) {
  const uint64_t mm = ctx->pen_mismatch, go = ctx->pen_gapopen, ge = ctx->pen_gapextend, d = ctx->resolution;
  // 8- or 16-bit arithmetic exactly as set_bit_mode decides (src/algo.cc:96-120)
  const uint64_t diff_saturation = std::min<uint64_t>(255 / mm, 255 / (go + ge));
  const uint32_t sat = d > diff_saturation ? 65535u : 255u;
  const uint64_t T = d * std::max<uint64_t>(mm, go + ge);
  const uint64_t W64 = T / ge + 1;
  const bool generic = 2 * W64 + 3 > 64;                     // band + the two guard lanes must fit a group
  AlignArgs a{};
  a.seqs = ctx->db.seqs; a.seq_off = ctx->db.seq_off; a.seqlen = ctx->db.seqlen;
  a.query = query;
  a.queries = d_queries;
  a.ntargets = max_count;
  a.ntargets_dev = d_count;
  a.targets = d_targets;
  a.diffs = d_diffs; a.scores = d_scores; a.alnlens = d_alnlens;
  a.mismatch = (uint32_t)mm; a.gapopen = (uint32_t)go; a.gapextend = (uint32_t)ge;
  a.sat = sat;
  a.W = (int)W64;
  a.maxwords = ((ctx->db.longest + 31u) >> 5) + 1u;
  const bool wide = 2 * W64 + 3 > 32;
  const int groups = wide ? 4 : 8;
  uint64_t blocks = ((uint64_t)max_count + groups - 1) / groups;
  const uint64_t cap = uint64_t(ctx->num_cus) * 8;
  if (blocks > cap) { blocks = cap; }
  if (blocks < 1) { blocks = 1; }
  const size_t lds = sizeof(uint64_t) * (size_t)a.maxwords * (2 * groups);
  const char * no_wfa = std::getenv("SWA_ALIGN_BANDED");         // test / comparison switch: force the banded kernel
  if (ctx->wfa_steps != 0 && !wide && ctx->db.longest < 65000u && (no_wfa == nullptr || no_wfa[0] == '0')) {
    WfaArgs w{};
    w.a = a;
    w.steps = static_cast<const swa_wfa_step *>(ctx->d_wfa.ptr);
    w.nsteps = ctx->wfa_steps;
    w.ring = std::min<uint32_t>(ctx->wfa_ring != 0 ? ctx->wfa_ring : ctx->wfa_steps, ctx->wfa_steps);
    // the diagonals a score <= T can reach: a gap of g columns costs open + g extend, so |k| <= (T - open) / extend — the band
    // of the wavefront kernel (the banded kernels keep the wider T / extend + 1).  Where band + guards fit 16 lanes, a wave
    // takes four pairs (SWA_ALIGN_WFA_LANES=32: two, as until round 5)
    const uint64_t wfa_w = T >= go + ge ? (T - go) / ge : 0;
    static const bool lanes32 = [] { const char * e = std::getenv("SWA_ALIGN_WFA_LANES"); return e != nullptr && std::atoi(e) == 32; }();
    const bool narrow = !lanes32 && 2 * wfa_w + 3 <= 16;
    const uint32_t per_block = narrow ? 8u : 4u;
    if (narrow) { w.a.W = (int)wfa_w; }
    uint64_t wblocks = ((uint64_t)max_count + per_block - 1) / per_block;
    if (wblocks > 2 * cap) { wblocks = 2 * cap; }
    if (wblocks < 1) { wblocks = 1; }
    const size_t wlds = sizeof(uint64_t) * (size_t)a.maxwords * 2 * per_block + sizeof(uint16_t) * per_block * (size_t)w.ring * 3 * (narrow ? 16 : 32);
    if (narrow) { hipLaunchKernelGGL(k_align_wfa<16>, dim3((unsigned)wblocks), dim3(128), wlds, ctx->stream, w); }
    else { hipLaunchKernelGGL(k_align_wfa<32>, dim3((unsigned)wblocks), dim3(128), wlds, ctx->stream, w); }
  } else if (generic) {
    a.W = W64 > 0x3FFFFFFF ? 0x3FFFFFFF : (int)W64;
    const uint32_t qcap = ctx->db.longest + 1u;
    uint64_t nthreads = max_count < 65536 ? (((uint64_t)max_count + 63) / 64) * 64 : 65536;
    if (nthreads < 64) { nthreads = 64; }
    while (nthreads > 64 && nthreads * 6ull * qcap * sizeof(uint32_t) > (2ull << 30)) { nthreads /= 2; }
    SWA_TRY(swa_reserve(ctx, ctx->d_queue, nthreads * 6ull * qcap * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_align_generic, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, ctx->stream, a,
                       static_cast<uint32_t *>(ctx->d_queue.ptr), (uint32_t)nthreads, qcap);
  } else if (d_alnlens != nullptr) {
    if (wide) { hipLaunchKernelGGL((k_align<64, true>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); }
    else { hipLaunchKernelGGL((k_align<32, true>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); }
  } else {
    if (wide) { hipLaunchKernelGGL((k_align<64, false>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); }
    else { hipLaunchKernelGGL((k_align<32, false>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); }
  }
  SWA_HIP(ctx, hipGetLastError());
  return SWA_OK;
}

extern "C" int swa_search_do(swa_ctx * ctx, uint64_t query_no, uint64_t listlength, const uint64_t * targets,
                             uint64_t * scores, uint64_t * diffs, uint64_t * alignlengths) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (!ctx->search_ready || ctx->db.n == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_search_do: call swa_search_begin first"); }
  if (listlength == 0) { return SWA_OK; }
  if (targets == nullptr || diffs == nullptr || query_no >= ctx->db.n || listlength > 0xFFFFFFFFull) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_search_do: bad argument");
  }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  // staging: [a] u64 in/out, [b] u32 targets, [c] u32 diffs | scores | alnlens
  SWA_TRY(swa_reserve(ctx, ctx->d_list_a, listlength * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_list_b, listlength * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_list_c, 3 * listlength * sizeof(uint32_t)));
  auto * d64 = static_cast<uint64_t *>(ctx->d_list_a.ptr);
  auto * t32 = static_cast<uint32_t *>(ctx->d_list_b.ptr);
  auto * r32 = static_cast<uint32_t *>(ctx->d_list_c.ptr);
  unsigned cb = (unsigned)std::min<uint64_t>((listlength + 255) / 256, uint64_t(ctx->num_cus) * 8);
  SWA_HIP(ctx, hipMemcpyAsync(d64, targets, listlength * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_narrow, dim3(cb), dim3(256), 0, ctx->stream, d64, t32, listlength);
  SWA_TRY(swa_align_launch(ctx, (uint32_t)query_no, nullptr, t32, nullptr, (uint32_t)listlength, r32,
                           scores != nullptr ? r32 + listlength : nullptr,
                           alignlengths != nullptr ? r32 + 2 * listlength : nullptr));
  uint64_t * outs[3] = {diffs, scores, alignlengths};
  for (int k = 0; k < 3; ++k) {
    if (outs[k] == nullptr) { continue; }
    hipLaunchKernelGGL(k_widen, dim3(cb), dim3(256), 0, ctx->stream, r32 + k * listlength, d64, listlength);
    SWA_HIP(ctx, hipMemcpyAsync(outs[k], d64, listlength * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  SWA_HIP(ctx, hipGetLastError());
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}

// loads this translation unit's code object (see swa_ctx_warmup): an empty launch
__global__ void k_warm_align() {}
void swa_warm_align(swa_ctx * ctx) { hipLaunchKernelGGL(k_warm_align, dim3(1), dim3(64), 0, ctx->stream); }
