// stubs.hip — seams not implemented yet in this build return an explicit error
// (never a CPU fallback).  Each stub disappears when its kernel file lands.
#include "swa_internal.h"

#ifndef SWA_HAVE_QGRAM
extern "C" int swa_qgram_build(swa_ctx * ctx) { return swa_fail_msg(ctx, SWA_E_ARG, "swa_qgram_build: not implemented in this build"); }
extern "C" int swa_qgram_diff(swa_ctx * ctx, uint64_t, uint64_t, const uint64_t *, uint64_t *) {
  return swa_fail_msg(ctx, SWA_E_ARG, "swa_qgram_diff: not implemented in this build");
}
extern "C" int swa_qgram_debug_read(swa_ctx * ctx, uint8_t *, size_t) {
  return swa_fail_msg(ctx, SWA_E_ARG, "swa_qgram_debug_read: not implemented in this build");
}
#endif
#ifndef SWA_HAVE_SEARCH
extern "C" int swa_search_begin(swa_ctx * ctx, uint64_t, uint64_t, uint64_t, uint64_t) {
  return swa_fail_msg(ctx, SWA_E_ARG, "swa_search_begin: not implemented in this build");
}
extern "C" int swa_search_do(swa_ctx * ctx, uint64_t, uint64_t, const uint64_t *, uint64_t *, uint64_t *, uint64_t *) {
  return swa_fail_msg(ctx, SWA_E_ARG, "swa_search_do: not implemented in this build");
}
#endif
