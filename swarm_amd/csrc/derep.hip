// derep.hip — d = 0: dereplication on the GPU.
//
// The reference walks the amplicons in db order and, for each, finds the bucket of the
// identical sequence in an open-addressing table keyed by the Zobrist hash, comparing the
// packed sequences on a hash match (src/derep.cc:276-354).  What the clustering needs from
// that loop is, per amplicon, the FIRST amplicon (smallest db index) with the identical
// sequence; cluster order, mass and the member chains follow from it on the host.
//
// Here every amplicon claims the table slot of its key with one atomicCAS (identical
// sequences share a key, hence a slot — a sequence with 10^6 copies costs 10^6 atomics on one
// slot, not a 10^6-long probe chain), the slot keeps the minimum claimant index, and a second
// kernel verifies each amplicon against its slot's minimum.  Two different sequences with the
// same 64-bit key (astronomically rare; forced in the tests with a narrowed key) share a slot:
// the group that does not own the minimum fails verification and goes round again with a
// re-mixed key until nobody is left.
#include "swa_internal.h"

namespace {

// key of hash h in round `round`: the hash itself first, a splitmix64 re-mix of it afterwards.
// keybits = 64 in production; the tests narrow it (6 more bits every round, so the rounds end)
__device__ __forceinline__ uint64_t derep_key(uint64_t h, uint32_t keybits, uint32_t round) {
  uint64_t k = h;
  if (round != 0) {
    k += 0x9E3779B97F4A7C15ull * round;
    k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
    k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
    k ^= k >> 31;
  }
  const uint32_t bits = keybits + 6u * round;
  if (bits < 64u) { k &= (1ull << bits) - 1ull; }
  return k == 0 ? 1ull : k;                          // 0 marks a free slot
}

__global__ __launch_bounds__(256) void k_derep_claim(const uint64_t * __restrict__ seqhash,
                                                     const uint32_t * __restrict__ list, uint32_t count,
                                                     uint32_t keybits, uint32_t round, unsigned long long * keys,
                                                     uint32_t * rep, uint64_t tmask, uint32_t * slot_of) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
    const uint32_t i = list != nullptr ? list[t] : t;
    const uint64_t key = derep_key(seqhash[i], keybits, round);
    uint64_t idx = (key ^ (key >> 32)) & tmask;
    for (;;) {
      const unsigned long long old = atomicCAS(&keys[idx], 0ull, (unsigned long long)key);
      if (old == 0ull || old == key) { break; }
      idx = (idx + 1) & tmask;
    }
    atomicMin(&rep[idx], i);
    slot_of[i] = (uint32_t)idx;
  }
}

__global__ __launch_bounds__(256) void k_derep_resolve(const uint64_t * __restrict__ seqs,
                                                       const uint64_t * __restrict__ seq_off,
                                                       const uint32_t * __restrict__ seqlen,
                                                       const uint32_t * __restrict__ list, uint32_t count,
                                                       const uint32_t * __restrict__ rep,
                                                       const uint32_t * __restrict__ slot_of, uint32_t * first_identical,
                                                       uint32_t * next_list, uint32_t * next_count) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
    const uint32_t i = list != nullptr ? list[t] : t;
    const uint32_t r = rep[slot_of[i]];
    bool same = r == i;
    if (!same && seqlen[r] == seqlen[i]) {
      const uint64_t * x = seqs + seq_off[i];
      const uint64_t * y = seqs + seq_off[r];
      same = true;
      for (uint32_t w = 0; w < ((seqlen[i] + 31u) >> 5); ++w) { same = same && (x[w] == y[w]); }
    }
    if (same) { first_identical[i] = r; }
    else { next_list[atomicAdd(next_count, 1u)] = i; }
  }
}

}  // namespace

extern "C" int swa_derep(swa_ctx * ctx, uint32_t * first_identical) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (ctx->db.n == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_derep: no database"); }
  if (first_identical == nullptr) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_derep: bad argument"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->db.n;
  SWA_TRY(swa_hash_sequences(ctx));
  // test hook: narrow the key so that distinct sequences collide (exercises the re-key rounds)
  uint32_t keybits = 64;
  if (const char * bits = std::getenv("SWA_DEREP_KEY_BITS")) {
    const int b = std::atoi(bits);
    if (b >= 1 && b < 64) { keybits = (uint32_t)b; }
  }
  const uint64_t tsize = swa_hashtable_size(n);
  SWA_TRY(swa_reserve(ctx, ctx->d_table, tsize * sizeof(unsigned long long)));        // keys
  SWA_TRY(swa_reserve(ctx, ctx->d_counts, tsize * sizeof(uint32_t)));                 // minimum claimant per slot
  SWA_TRY(swa_reserve(ctx, ctx->d_cursor, uint64_t(n) * sizeof(uint32_t)));           // slot of every amplicon
  SWA_TRY(swa_reserve(ctx, ctx->d_graft, uint64_t(n) * sizeof(uint32_t)));            // result
  SWA_TRY(swa_reserve(ctx, ctx->d_list_a, uint64_t(n) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_list_b, uint64_t(n) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_flags, 16 * sizeof(uint32_t)));
  ctx->d1_ready = false;                                   // d_table is re-purposed
  ctx->full_index = false;
  ctx->anchor_ready = false;
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_table.ptr, 0, tsize * sizeof(unsigned long long), ctx->stream));
  SWA_HIP(ctx, hipMemsetAsync(ctx->d_counts.ptr, 0xFF, tsize * sizeof(uint32_t), ctx->stream));

  auto * keys = static_cast<unsigned long long *>(ctx->d_table.ptr);
  auto * rep = static_cast<uint32_t *>(ctx->d_counts.ptr);
  auto * slot_of = static_cast<uint32_t *>(ctx->d_cursor.ptr);
  auto * result = static_cast<uint32_t *>(ctx->d_graft.ptr);
  auto * next_count = static_cast<uint32_t *>(ctx->d_flags.ptr) + 4;
  uint32_t * lists[2] = {static_cast<uint32_t *>(ctx->d_list_a.ptr), static_cast<uint32_t *>(ctx->d_list_b.ptr)};
  const uint32_t * active = nullptr;                       // round 0: every amplicon
  uint32_t count = n;
  for (uint32_t round = 0; count > 0; ++round) {
    if (round > 64) { return swa_fail_msg(ctx, SWA_E_DEVICE, "swa_derep: key collisions did not resolve"); }
    uint32_t * next = lists[round & 1u];
    const int grid = (int)std::min<uint64_t>((uint64_t(count) + 255) / 256, (uint64_t)ctx->num_cus * 16);
    SWA_HIP(ctx, hipMemsetAsync(next_count, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_derep_claim, dim3(grid), dim3(256), 0, ctx->stream,
                       static_cast<const uint64_t *>(ctx->d_seqhash.ptr), active, count, keybits, round, keys, rep,
                       tsize - 1, slot_of);
    hipLaunchKernelGGL(k_derep_resolve, dim3(grid), dim3(256), 0, ctx->stream, ctx->db.seqs, ctx->db.seq_off,
                       ctx->db.seqlen, active, count, rep, slot_of, result, next, next_count);
    SWA_HIP(ctx, hipGetLastError());
    uint32_t left = 0;
    SWA_HIP(ctx, hipMemcpyAsync(&left, next_count, sizeof(left), hipMemcpyDeviceToHost, ctx->stream));
    SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    active = next;
    count = left;
  }
  SWA_HIP(ctx, hipMemcpyAsync(first_identical, result, uint64_t(n) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}
