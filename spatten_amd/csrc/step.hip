// step.hip — the device-resident step state (include/spatten.h, ABI 3).
//
// The reference's decode loop (run_spatten_llama.py:27-35) re-enters Python for every token because the cache length —
// which sizes `torch.cat`, the rotary gather `cos[position_ids]` and the stash — is a host integer.  Here the length and
// the query position live in a 64-byte header in device memory, followed by the rotary rows of the two positions a
// decode step rotates with (the query's, and the appended key's slot: modify_llama.py:92,103-104).  One tiny launch per
// TOKEN (not per layer) moves the state forward; the attention launches of all layers read it.  Nothing in a token's
// launch sequence depends on a host value any more, so the sequence is captured once and replayed.
#include "common.h"

namespace spatten {

// state: int32 words {kv_len, pos_q, steps since the last set, rows the previous step's stash holds, 0...}
//        | cos rows [2][half] | sin rows [2][half]   (row 0: pos_q, row 1: kv_len - 1)
template <typename T>
__global__ void step_update_kernel(int32_t* st, const T* cos, const T* sin, int table_rows, int half, int set, int kv_len,
                                   int pos_q, int delta) {
  int n = set ? kv_len : st[0] + delta;
  int pq = set ? pos_q : st[1] + delta;
  // words 2 / 3 (the fused cascade accumulation of a captured step, decode_attn.hip): how many steps ran since the set,
  // and how many rows the PREVIOUS step's stash holds (0 for the first step after a set: nothing to fold)
  const int cnt = set ? 0 : st[2] + 1;
  const int prev = (set || st[2] == 0) ? 0 : st[0];
  __syncthreads();                         // every thread has read the old words before thread 0 replaces them
  if (threadIdx.x == 0) { st[0] = n; st[1] = pq; st[2] = cnt; st[3] = prev; }
  T* rows = reinterpret_cast<T*>(reinterpret_cast<char*>(st) + kStepHeader);
  const int rq = min(max(pq, 0), table_rows - 1);
  const int rn = min(max(n - 1, 0), table_rows - 1);
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    rows[i] = cos[(int64_t)rq * half + i];
    rows[half + i] = cos[(int64_t)rn * half + i];
    rows[2 * half + i] = sin[(int64_t)rq * half + i];
    rows[3 * half + i] = sin[(int64_t)rn * half + i];
  }
}

static int step_update(void* state, int dtype, int head_dim, const void* cos, const void* sin, int table_rows, int set,
                       int kv_len, int pos_q, int delta, hipStream_t st) {
  if (!state || !cos || !sin || table_rows <= 0 || !ok_dtype(dtype)) return SPATTEN_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128 && head_dim != 256) return SPATTEN_ERR_UNSUPPORTED;
  if (set && (kv_len < 0 || pos_q < 0)) return SPATTEN_ERR_INVALID;
  SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((step_update_kernel<T>), dim3(1), dim3(128), 0, st, (int32_t*)state,
                                             (const T*)cos, (const T*)sin, table_rows, head_dim / 2, set, kv_len, pos_q, delta));
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

}  // namespace spatten

using namespace spatten;

extern "C" size_t spatten_step_state_bytes(int dtype, int head_dim) {
  if (!ok_dtype(dtype) || head_dim <= 0) return 0;
  const size_t elt = dtype == SPATTEN_F32 ? 4 : 2;
  return (kStepHeader + 4 * (size_t)(head_dim / 2) * elt + 255) / 256 * 256;
}

extern "C" int spatten_step_set(void* state, int dtype, int head_dim, const void* cos, const void* sin, int table_rows,
                                int kv_len, int pos_q, void* stream) {
  return step_update(state, dtype, head_dim, cos, sin, table_rows, 1, kv_len, pos_q, 0, (hipStream_t)stream);
}

extern "C" int spatten_step_advance(void* state, int dtype, int head_dim, const void* cos, const void* sin, int table_rows,
                                    int delta, void* stream) {
  return step_update(state, dtype, head_dim, cos, sin, table_rows, 0, 0, 0, delta, (hipStream_t)stream);
}

