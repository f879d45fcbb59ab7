// prefill_attn.hip — placeholder until the MFMA flash kernel lands (next commit).
#include "common.h"
extern "C" size_t spatten_prefill_workspace_bytes(int, int, int, int, int, int, int) { return 256; }
extern "C" int spatten_attn_prefill(int, const void*, int64_t, int64_t, int64_t, const void*, const void*, int64_t,
                                    int64_t, const void*, const void*, int, const int64_t*, int64_t, const void*,
                                    int64_t, int64_t, void*, int64_t, int64_t, void*, int64_t, int64_t, int64_t,
                                    float*, void*, int, int, int, int, int, int, int, int, void*) {
  return SPATTEN_ERR_UNSUPPORTED;
}
