// ctx.hip — context, error reporting, database residency (seam L2 of include/swarm_amd.h).
#include "swa_internal.h"

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <cstdio>

int swa_fail(swa_ctx * ctx, int code, const char * what, hipError_t e) {
  if (ctx != nullptr) {
    ctx->err = std::string(what) + ": " + hipGetErrorString(e);
  }
  return code;
}

int swa_fail_msg(swa_ctx * ctx, int code, const std::string & msg) {
  if (ctx != nullptr) { ctx->err = msg; }
  return code;
}

int swa_reserve(swa_ctx * ctx, swa_dbuf & buf, size_t bytes) {
  if (bytes <= buf.bytes && buf.ptr != nullptr) { return SWA_OK; }
  if (buf.ptr != nullptr) {
    (void)hipFree(buf.ptr);
    buf.ptr = nullptr;
    buf.bytes = 0;
  }
  if (bytes == 0) { bytes = 16; }
  static const bool timing = getenv("SWARM_AMD_ALLOC_TIMING") != nullptr;      // (development aid: what every allocation costs)
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = hipMalloc(&buf.ptr, bytes);
  if (timing) {
    fprintf(stderr, "[alloc] %12zu bytes  %8.1f us\n", bytes, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  if (e != hipSuccess) {
    buf.ptr = nullptr;
    return swa_fail(ctx, SWA_E_NOMEM, "hipMalloc", e);
  }
  buf.bytes = bytes;
  return SWA_OK;
}

void swa_release(swa_dbuf & buf) {
  if (buf.ptr != nullptr) { (void)hipFree(buf.ptr); }
  buf.ptr = nullptr;
  buf.bytes = 0;
}

extern "C" int swa_abi_version(void) { return SWA_ABI_VERSION; }

extern "C" int swa_ctx_create(int device, void * stream, swa_ctx ** out) {
  if (out == nullptr) { return SWA_E_ARG; }
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
    return SWA_E_DEVICE;   // no CPU fallback, ever
  }
  if (hipSetDevice(device) != hipSuccess) { return SWA_E_DEVICE; }
  hipDeviceProp_t prop{};
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { return SWA_E_DEVICE; }
  auto * ctx = new swa_ctx();
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (stream != nullptr) {
    ctx->stream = static_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return SWA_E_DEVICE;
    }
    ctx->own_stream = true;
  }
  // (the anomaly hunt: SWARM_AMD_POISON_MB=N fills N MB of HBM with non-zero bytes and frees them again first, so that
  // what the context allocates next has been 0xA5 / 0xFF / 0x01 / 0x5A, not the zeros of a fresh box)
  if (const char * poison = getenv("SWARM_AMD_POISON_MB")) {
    std::vector<void *> held;
    for (long left = atol(poison), k = 0; left > 0; left -= 256, ++k) {
      void * p = nullptr;
      if (hipMalloc(&p, (size_t)256 << 20) != hipSuccess) { break; }
      (void)hipMemset(p, (int)((const unsigned char[]){0xA5, 0xFF, 0x01, 0x5A}[k % 4]), (size_t)256 << 20);
      held.push_back(p);
    }
    (void)hipDeviceSynchronize();
    for (void * p : held) { (void)hipFree(p); }
  }
  // the status block and the views into it (swa_internal.h)
  if (swa_reserve(ctx, ctx->d_status, 4096) != SWA_OK || hipMemset(ctx->d_status.ptr, 0, 4096) != hipSuccess ||
      hipHostMalloc(&ctx->h_status, 4096, hipHostMallocDefault) != hipSuccess) {
    swa_release(ctx->d_status);
    if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
    return SWA_E_NOMEM;
  }
  auto * base = static_cast<uint8_t *>(ctx->d_status.ptr);
  ctx->d_flags.ptr = base;            ctx->d_flags.bytes = 64;
  ctx->d_stats.ptr = base + 64;       ctx->d_stats.bytes = 128;
  ctx->d_guard.ptr = base + 192;      ctx->d_guard.bytes = 192;
  ctx->d_acounters.ptr = base + 1024; ctx->d_acounters.bytes = 1024;
  *out = ctx;
  return SWA_OK;
}

// First use of a device costs: memory pools, copy queues and — per translation unit — the load of its code object
// at the first kernel launch (~0.25 s in all at 10 M amplicons).  A caller that has something else to do meanwhile
// (reading a FASTA file) runs this on a helper thread right after swa_ctx_create.
void swa_warm_d1(swa_ctx * ctx);          // d1.hip
void swa_warm_cluster(swa_ctx * ctx);     // cluster_gpu.hip
void swa_warm_dn(swa_ctx * ctx);          // dn_graph.hip
void swa_warm_align(swa_ctx * ctx);       // align.hip
void swa_warm_qgram(swa_ctx * ctx);       // qgram.hip
void swa_warm_scan(swa_ctx * ctx);        // scan.hip
__global__ void k_warm(uint32_t * p) { if (threadIdx.x == 0u && p != nullptr) { p[0] = 1u; } }

// differences: the d the caller is going to cluster with — 1: the d = 1 step and the clustering on the device; >= 2: the
// q-gram / pair-graph / alignment kernels and the clustering; anything else (swa_ctx_warmup): all of them.  A code object
// costs 3-5 ms to load: a `swarm -d 1` run does not wait for the four it never launches.
extern "C" int swa_ctx_warmup_for(swa_ctx * ctx, int differences) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  void * d = nullptr;
  std::vector<uint8_t> host(1u << 20, 1);
  SWA_HIP(ctx, hipMalloc(&d, 64u << 20));
  SWA_HIP(ctx, hipMemcpyAsync(d, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, ctx->stream, static_cast<uint32_t *>(d));
  const bool all = differences < 1;
  if (all || differences == 1) { swa_warm_d1(ctx); }
  swa_warm_cluster(ctx);
  if (all || differences >= 2) { swa_warm_dn(ctx); swa_warm_align(ctx); swa_warm_qgram(ctx); swa_warm_scan(ctx); }
  SWA_HIP(ctx, hipMemcpyAsync(host.data(), d, 4096, hipMemcpyDeviceToHost, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SWA_HIP(ctx, hipFree(d));
  return SWA_OK;
}

extern "C" int swa_ctx_warmup(swa_ctx * ctx) { return swa_ctx_warmup_for(ctx, -1); }

// The first download of a process into pinned memory costs the runtime ~7.5 ms whatever its size (the copy engine's queue for
// that direction is made then: tools/experiments/d2h_cost.hip, lease r6g — 40 MB: 7.4 ms in the call + 0.8 ms of transfer the
// first time, 0.75 ms afterwards).  A caller with a thread to spare pays it here, on a stream of its own, beside the code-object
// loads and uploads of its start-up — not in front of the first result it waits for.
extern "C" int swa_ctx_warmup_downloads(swa_ctx * ctx) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  // (called beside the context's owner at work: the status comes back, the context's error text is not touched from here)
  if (hipSetDevice(ctx->device) != hipSuccess) { return SWA_E_DEVICE; }
  constexpr size_t kBytes = 256u << 10;
  void * d = nullptr, * h = nullptr;
  hipStream_t s = nullptr;
  if (hipMalloc(&d, kBytes) != hipSuccess) { return SWA_E_NOMEM; }
  if (hipHostMalloc(&h, kBytes, hipHostMallocDefault) != hipSuccess) { (void)hipFree(d); return SWA_E_NOMEM; }
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e == hipSuccess) { e = hipMemcpyAsync(h, d, kBytes, hipMemcpyDeviceToHost, s); }
  if (e == hipSuccess) { e = hipStreamSynchronize(s); }
  if (s != nullptr) { (void)hipStreamDestroy(s); }
  (void)hipHostFree(h);
  (void)hipFree(d);
  return e == hipSuccess ? SWA_OK : SWA_E_DEVICE;
}

extern "C" void swa_ctx_destroy(swa_ctx * ctx) {
  if (ctx == nullptr) { return; }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (swa_dbuf * b : {&ctx->d_seqs, &ctx->d_seq_off, &ctx->d_seqlen, &ctx->d_abund, &ctx->d_zobrist,
                       &ctx->d_seqhash, &ctx->d_table, &ctx->d_bloom, &ctx->d_patterns, &ctx->d_status,
                       &ctx->d_edges, &ctx->d_counts, &ctx->d_cursor, &ctx->d_scan_tmp,
                       &ctx->d_offsets_tmp, &ctx->d_nb_tmp, &ctx->d_qgrams, &ctx->d_list_a, &ctx->d_list_b,
                       &ctx->d_list_c, &ctx->d_list_d, &ctx->d_light, &ctx->d_graft, &ctx->d_bloomflex,
                       &ctx->d_fpatterns, &ctx->d_queue, &ctx->d_fcounters, &ctx->d_scan_est, &ctx->d_scan_swarmed,
                       &ctx->d_scan_targets, &ctx->d_scan_diffs, &ctx->d_scan_hits, &ctx->d_scan_counters, &ctx->d_scan_cand, &ctx->d_scan_seeds, &ctx->d_scan_compares, &ctx->d_aux,
                       &ctx->d_afallback, &ctx->d_arank, &ctx->d_rank_tmp, &ctx->d_wfa, &ctx->d_long_rows, &ctx->d_seg_fill, &ctx->d_seg_base, &ctx->d_akeys[0], &ctx->d_acounts[0],
                       &ctx->d_aitems[0], &ctx->d_aitems[1],
                       &ctx->d_frole, &ctx->d_fkeys, &ctx->d_fcnt, &ctx->d_foff, &ctx->d_fslot, &ctx->d_fmembers, &ctx->d_fitems,
                       &ctx->d_fpairs, &ctx->d_dn_keys, &ctx->d_dn_vals, &ctx->d_cluster, &ctx->d_cluster_ctl, &ctx->d_words_stage}) {
    swa_release(*b);
  }
  for (auto & b : ctx->d_stream) { swa_release(b); }
  if (ctx->h_scan_pinned != nullptr) { (void)hipHostFree(ctx->h_scan_pinned); }
  if (ctx->h_status != nullptr) { (void)hipHostFree(ctx->h_status); }
  if (ctx->ev_ready) { for (auto & e : ctx->ev) { (void)hipEventDestroy(e); } }
  if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); }
  delete ctx;
}

extern "C" const char * swa_last_error(const swa_ctx * ctx) {
  return ctx != nullptr ? ctx->err.c_str() : "null context";
}

extern "C" int swa_timing_enable(swa_ctx * ctx, int on) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  if (on != 0 && !ctx->ev_ready) {
    for (auto & e : ctx->ev) { SWA_HIP(ctx, hipEventCreate(&e)); }
    ctx->ev_ready = true;
  }
  ctx->timing = on != 0;
  return SWA_OK;
}

static int timing_read(swa_ctx * ctx, float * ms, int first, int count) {
  if (ctx == nullptr || ms == nullptr) { return SWA_E_ARG; }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int s = 0; s < count; ++s) {
    ms[s] = 0.0f;
    if (ctx->ev_ready && ctx->ev_used[first + s]) { (void)hipEventElapsedTime(&ms[s], ctx->ev[2 * (first + s)], ctx->ev[2 * (first + s) + 1]); }
  }
  return SWA_OK;
}

extern "C" int swa_timing_read(swa_ctx * ctx, float * ms8) { return timing_read(ctx, ms8, 0, 8); }
extern "C" int swa_timing_read_stream(swa_ctx * ctx, float * ms8) { return timing_read(ctx, ms8, 8, 8); }

extern "C" int swa_ctx_synchronize(swa_ctx * ctx) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWA_OK;
}

static int check_view(swa_ctx * ctx, const swa_db_view * v) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (v == nullptr || v->seqs == nullptr || v->seq_off == nullptr || v->seqlen == nullptr ||
      v->abundance == nullptr) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_db: null array in db view");
  }
  if (v->n == 0 || v->longest == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_db: empty database"); }
  return SWA_OK;
}

static void invalidate(swa_ctx * ctx) {
  ctx->d1_ready = false;
  ctx->full_index = false;
  ctx->anchor_ready = false;
  ctx->rank_ready = false;
  ctx->db_unordered = false;
  ctx->props_ready = false;
  ctx->windows_ready = false;
  ctx->lines_ready = false;
  ctx->stream_index = false;
  ctx->member_index = false;
  ctx->stream_extra_bits = 0;
  ctx->anchor_a = ctx->anchor_b = 0;
  ctx->anchor_w = ctx->windows_w = 32;
  ctx->guard_index = false;
  ctx->guard_keys_done = ctx->guard_keys_pending = false;
  ctx->qgram_ready = false;
  ctx->scan_ready = false;
  ctx->dn_graph_ready = false;
  ctx->dn_shortest = 0;
  ctx->csr_ready = false;
  ctx->cluster_ready = false;
}

extern "C" int swa_db_upload(swa_ctx * ctx, const swa_db_view * h) {
  SWA_TRY(check_view(ctx, h));
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t words = h->seq_off[h->n];
  // one zero word of slack after the last sequence: kernels may read seq[nw] as padding
  SWA_TRY(swa_reserve(ctx, ctx->d_seqs, (words + 2) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_seq_off, (uint64_t(h->n) + 1) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_seqlen, uint64_t(h->n) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_abund, uint64_t(h->n) * sizeof(uint64_t)));
  SWA_HIP(ctx, hipMemsetAsync(static_cast<uint64_t *>(ctx->d_seqs.ptr) + words, 0, 2 * sizeof(uint64_t), ctx->stream));
  SWA_HIP(ctx, hipMemcpyAsync(ctx->d_seqs.ptr, h->seqs, words * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  SWA_HIP(ctx, hipMemcpyAsync(ctx->d_seq_off.ptr, h->seq_off, (uint64_t(h->n) + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  SWA_HIP(ctx, hipMemcpyAsync(ctx->d_seqlen.ptr, h->seqlen, uint64_t(h->n) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  SWA_HIP(ctx, hipMemcpyAsync(ctx->d_abund.ptr, h->abundance, uint64_t(h->n) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->db.n = h->n;
  ctx->db.longest = h->longest;
  ctx->db.seqs = static_cast<const uint64_t *>(ctx->d_seqs.ptr);
  ctx->db.seq_off = static_cast<const uint64_t *>(ctx->d_seq_off.ptr);
  ctx->db.seqlen = static_cast<const uint32_t *>(ctx->d_seqlen.ptr);
  ctx->db.abundance = static_cast<const uint64_t *>(ctx->d_abund.ptr);
  ctx->db_owned = true;
  invalidate(ctx);
  return SWA_OK;
}

// ---- the database in file order, put in db order on the device -------------------------------------------------------
// words[k] = words of amplicon k; word offsets by an exclusive scan; one lane per amplicon copies its words (5 at 150 nt)
__global__ __launch_bounds__(256) void k_db_word_counts(const uint32_t * __restrict__ seqlen, uint32_t n, uint64_t * __restrict__ counts) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += gridDim.x * blockDim.x) { counts[k] = k < n ? (uint64_t)((seqlen[k] + 31u) >> 5) : 0ull; }
}
__global__ __launch_bounds__(256) void k_db_gather_words(const uint64_t * __restrict__ file_words, const uint64_t * __restrict__ src_off,
                                                         const uint64_t * __restrict__ seq_off, uint32_t n, uint64_t * __restrict__ seqs) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const uint64_t from = src_off[k], to = seq_off[k], nw = seq_off[k + 1] - to;
    for (uint64_t j = 0; j < nw; ++j) { seqs[to + j] = file_words[from + j]; }
  }
}

extern "C" int swa_db_stage_words(swa_ctx * ctx, const uint64_t * const * piece_words, const uint64_t * piece_word_count, uint32_t pieces) {
  if (ctx == nullptr || piece_words == nullptr || piece_word_count == nullptr || pieces == 0) { return SWA_E_ARG; }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t total = 0;
  for (uint32_t p = 0; p < pieces; ++p) { total += piece_word_count[p]; }
  SWA_TRY(swa_reserve(ctx, ctx->d_words_stage, (total + 2) * sizeof(uint64_t)));
  uint64_t at = 0;
  for (uint32_t p = 0; p < pieces; ++p) {
    if (piece_word_count[p] != 0) {
      SWA_HIP(ctx, hipMemcpyAsync(static_cast<uint64_t *>(ctx->d_words_stage.ptr) + at, piece_words[p], piece_word_count[p] * sizeof(uint64_t),
                                  hipMemcpyHostToDevice, ctx->stream));
    }
    at += piece_word_count[p];
  }
  ctx->staged_first = piece_words[0];
  ctx->staged_words = total;
  return SWA_OK;
}

extern "C" int swa_db_upload_unordered(swa_ctx * ctx, const swa_db_unordered_view * h) {
  if (ctx == nullptr) { return SWA_E_ARG; }
  if (h == nullptr || h->piece_words == nullptr || h->piece_word_count == nullptr || h->src_off == nullptr || h->seqlen == nullptr ||
      h->abundance == nullptr) {
    return swa_fail_msg(ctx, SWA_E_ARG, "swa_db: null array in db view");
  }
  if (h->n == 0 || h->longest == 0 || h->pieces == 0) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_db: empty database"); }
  SWA_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t total = 0;
  for (uint32_t p = 0; p < h->pieces; ++p) { total += h->piece_word_count[p]; }
  if (ctx->staged_first != h->piece_words[0] || ctx->staged_words != total || ctx->d_words_stage.ptr == nullptr) {
    SWA_TRY(swa_db_stage_words(ctx, h->piece_words, h->piece_word_count, h->pieces));
  }
  const uint32_t n = h->n;
  SWA_TRY(swa_reserve(ctx, ctx->d_seqs, (total + 2) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_seq_off, (uint64_t(n) + 1) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_seqlen, uint64_t(n) * sizeof(uint32_t)));
  SWA_TRY(swa_reserve(ctx, ctx->d_abund, uint64_t(n) * sizeof(uint64_t)));
  // src_off travels through the buffer that will hold the offsets' scan input afterwards
  // (While the gather runs the packed words are in HBM twice — the staged pools and d_seqs: 2 x 0.4 GB at 10 M x 150, 2 x 4 GB
  // at 100 M; on SWA_E_NOMEM a caller can still take swa_hostdb_view + swa_db_upload, which gathers on the host.)  Every way
  // out of this function releases the temporaries; a failure also gives the staged words back.
  swa_dbuf d_src, d_counts, d_tmp;
  bool uploaded = false;
  struct Cleanup {
    swa_ctx * ctx; swa_dbuf & a; swa_dbuf & b; swa_dbuf & c; bool & ok;
    ~Cleanup() {
      swa_release(a); swa_release(b); swa_release(c);
      if (!ok) { swa_release(ctx->d_words_stage); ctx->staged_first = nullptr; ctx->staged_words = 0; }
    }
  } cleanup{ctx, d_src, d_counts, d_tmp, uploaded};
  SWA_TRY(swa_reserve(ctx, d_src, uint64_t(n) * sizeof(uint64_t)));
  SWA_TRY(swa_reserve(ctx, d_counts, (uint64_t(n) + 1) * sizeof(uint64_t)));
  auto release_tmp = [&]() {};
  auto fail_hip = [&](hipError_t e, const char * what) { release_tmp(); return swa_fail(ctx, SWA_E_DEVICE, what, e); };
  hipError_t e;
  if ((e = hipMemcpyAsync(d_src.ptr, h->src_off, uint64_t(n) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) { return fail_hip(e, "hipMemcpyAsync"); }
  if ((e = hipMemcpyAsync(ctx->d_seqlen.ptr, h->seqlen, uint64_t(n) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) { return fail_hip(e, "hipMemcpyAsync"); }
  if ((e = hipMemcpyAsync(ctx->d_abund.ptr, h->abundance, uint64_t(n) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) { return fail_hip(e, "hipMemcpyAsync"); }
  const int grid = std::max(1, std::min<int>(ctx->num_cus * 8, (int)((uint64_t(n) + 256) / 256)));
  hipLaunchKernelGGL(k_db_word_counts, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint32_t *>(ctx->d_seqlen.ptr), n,
                     static_cast<uint64_t *>(d_counts.ptr));
  size_t need = 0;
  (void)rocprim::exclusive_scan(nullptr, need, static_cast<uint64_t *>(d_counts.ptr), static_cast<uint64_t *>(ctx->d_seq_off.ptr), 0ull,
                                (size_t)n + 1, rocprim::plus<uint64_t>(), ctx->stream);
  if (swa_reserve(ctx, d_tmp, need + 16) != SWA_OK) { return SWA_E_NOMEM; }
  e = rocprim::exclusive_scan(d_tmp.ptr, need, static_cast<uint64_t *>(d_counts.ptr), static_cast<uint64_t *>(ctx->d_seq_off.ptr), 0ull,
                              (size_t)n + 1, rocprim::plus<uint64_t>(), ctx->stream);
  if (e != hipSuccess) { return fail_hip(e, "rocprim::exclusive_scan"); }
  hipLaunchKernelGGL(k_db_gather_words, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint64_t *>(ctx->d_words_stage.ptr),
                     static_cast<const uint64_t *>(d_src.ptr), static_cast<const uint64_t *>(ctx->d_seq_off.ptr), n,
                     static_cast<uint64_t *>(ctx->d_seqs.ptr));
  uint64_t placed = 0;                                        // = seq_off[n]: the words the amplicons take, <= the pools' words
  if ((e = hipMemcpyAsync(&placed, static_cast<uint64_t *>(ctx->d_seq_off.ptr) + n, sizeof(placed), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) { return fail_hip(e, "hipMemcpyAsync"); }
  if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) { return fail_hip(e, "hipStreamSynchronize"); }
  if (placed > total) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_db_upload_unordered: the amplicons' words exceed the pools'"); }
  SWA_HIP(ctx, hipMemsetAsync(static_cast<uint64_t *>(ctx->d_seqs.ptr) + placed, 0, 2 * sizeof(uint64_t), ctx->stream));
  SWA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uploaded = true;
  swa_release(ctx->d_words_stage);
  ctx->staged_first = nullptr;
  ctx->staged_words = 0;
  ctx->db.n = n;
  ctx->db.longest = h->longest;
  ctx->db.seqs = static_cast<const uint64_t *>(ctx->d_seqs.ptr);
  ctx->db.seq_off = static_cast<const uint64_t *>(ctx->d_seq_off.ptr);
  ctx->db.seqlen = static_cast<const uint32_t *>(ctx->d_seqlen.ptr);
  ctx->db.abundance = static_cast<const uint64_t *>(ctx->d_abund.ptr);
  ctx->db_owned = true;
  invalidate(ctx);
  return SWA_OK;
}

// Host memory of the caller's, pinned: a download into it (swa_d1_cluster_device's member order, swa_d1_network's lists) is
// one DMA at the link's speed instead of a staged copy through the runtime's bounce buffers (40 MB: ~1 ms instead of 5-8).
// Any thread may call these; the memory stays the caller's.
extern "C" int swa_host_pin(swa_ctx * ctx, void * ptr, size_t bytes) {
  if (ctx == nullptr || ptr == nullptr || bytes == 0) { return SWA_E_ARG; }
  // (a helper thread's call, beside the context's owner at work: the status comes back, the context's error text is not
  // touched from here)
  if (hipSetDevice(ctx->device) != hipSuccess) { return SWA_E_DEVICE; }
  return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? SWA_OK : SWA_E_DEVICE;
}
extern "C" void swa_host_unpin(void * ptr) { if (ptr != nullptr) { (void)hipHostUnregister(ptr); } }

extern "C" int swa_db_attach(swa_ctx * ctx, const swa_db_view * d) {
  SWA_TRY(check_view(ctx, d));
  ctx->db = *d;
  ctx->db_owned = false;
  invalidate(ctx);
  return SWA_OK;
}
