// cascade.hip — SpAtten semantics beyond what the reference's Python implements:
// cumulative importance, its compaction at a prune, head scores, local-V-pruned P·V.
// PARITY UNPINNED: no numeric implementation exists in the reference; checked against oracle/spatten_oracle.py.
#include "common.h"

namespace spatten {

// (max, sum exp) of one masked logit row — one workgroup per (b, h, i)
template <typename T>
__global__ __launch_bounds__(256) void row_lse_kernel(const T* __restrict__ stash, int64_t sb, int64_t sh, int64_t sq,
                                                      const T* __restrict__ mask, int64_t mask_sb, int64_t mask_sq,
                                                      float* __restrict__ lse, int H, int Q, int L, int causal) {
  __shared__ float red[4];
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* row = stash + b * sb + h * sh + i * sq;
  const T* mrow = mask ? mask + b * mask_sb + i * mask_sq : nullptr;
  const int vis = causal ? min(L, L - Q + i + 1) : L;
  float m = -INFINITY;
  for (int j = tid; j < vis; j += 256) {
    float s = DT<T>::to_f32(row[j]);
    if (mrow) s = DT<T>::round(s + DT<T>::to_f32(mrow[j]));
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  const float mu = (m == -INFINITY) ? 0.f : m;
  float l = 0.f;
  for (int j = tid; j < vis; j += 256) {
    float s = DT<T>::to_f32(row[j]);
    if (mrow) s = DT<T>::round(s + DT<T>::to_f32(mrow[j]));
    l += __expf(s - mu);
  }
  l = wave_sum(l);
  if (lane == 0) red[wave] = l;
  __syncthreads();
  if (tid == 0) {
    float* o = lse + ((int64_t)(b * H + h) * Q + i) * 2;
    o[0] = m;
    o[1] = red[0] + red[1] + red[2] + red[3];
  }
}

// acc[h, j] += sum_{b, i} exp(s - max) / sum    — one thread per (h, j); rows visited in a fixed order (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void importance_accumulate_kernel(const T* __restrict__ stash, int64_t sb, int64_t sh,
                                                                    int64_t sq, const float* __restrict__ lse,
                                                                    const T* __restrict__ mask, int64_t mask_sb,
                                                                    int64_t mask_sq, float* __restrict__ acc,
                                                                    int64_t acc_sh, int B, int H, int Q, int L, int causal) {
  const int j = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (j >= L) return;
  float a = 0.f;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < Q; ++i) {
      if (causal && j > L - Q + i) continue;
      const float* ml = lse + ((int64_t)(b * H + h) * Q + i) * 2;
      float s = DT<T>::to_f32(stash[b * sb + h * sh + i * sq + j]);
      if (mask) s = DT<T>::round(s + DT<T>::to_f32(mask[b * mask_sb + i * mask_sq + j]));
      const float mu = (ml[0] == -INFINITY) ? 0.f : ml[0];
      a += __expf(s - mu) / ml[1];
    }
  acc[h * acc_sh + j] += a;
}

__global__ __launch_bounds__(256) void importance_compact_kernel(const float* __restrict__ src, int64_t src_sh,
                                                                 float* __restrict__ dst, int64_t dst_sh,
                                                                 const int32_t* __restrict__ idx, int64_t idx_sh,
                                                                 int start, int k, int tail_lo, int Lp) {
  const int r = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (r >= Lp) return;
  int j;
  if (r < start) j = r;
  else if (r < start + k) j = idx[h * idx_sh + (r - start)];
  else j = tail_lo + (r - start - k);
  dst[h * dst_sh + r] = src[h * src_sh + j];
}

// scores[h] += sum |out[b, i, h*d + e]| — one workgroup per head, fixed order + tree reduce (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void head_scores_kernel(const T* __restrict__ out, int64_t out_sb, int64_t out_sq,
                                                          float* __restrict__ scores, int B, int Q, int d) {
  __shared__ float red[4];
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long n = (long long)B * Q * d;
  float a = 0.f;
  for (long long t = tid; t < n; t += 256) {
    const int e = (int)(t % d);
    const long long bi = t / d;
    const int i = (int)(bi % Q), b = (int)(bi / Q);
    a += fabsf(DT<T>::to_f32(out[b * out_sb + i * out_sq + (int64_t)h * d + e]));
  }
  a = wave_sum(a);
  if (lane == 0) red[wave] = a;
  __syncthreads();
  if (tid == 0) scores[h] += (red[0] + red[1]) + (red[2] + red[3]);
}

// out[b, h, :] = sum_i p(idx_i) * V[idx_i, :]   with p = exp(s - max) / sum of the FULL row — one workgroup per (b, h)
template <typename T, int D>
__global__ __launch_bounds__(256) void pv_gather_kernel(const T* __restrict__ stash, int64_t sc_sb, int64_t sc_sh,
                                                        const float* __restrict__ lse, const T* __restrict__ mask,
                                                        int64_t mask_sb, const T* __restrict__ vc, int64_t kv_sb,
                                                        int64_t kv_sh, const int32_t* __restrict__ idx, int64_t idx_sr,
                                                        int k, T* __restrict__ out, int64_t out_sb, int H, int Hkv) {
  constexpr int LPR = D / 8;            // lanes per V row (8 elements = 16 bytes (16-bit) / 32 bytes (fp32) each)
  constexpr int RPI = 256 / LPR;
  __shared__ float s_o[4][D];
  using V8 = Vec8<T>;
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, c = tid % LPR, r = tid / LPR;
  const int lane = tid & 63, wave = tid >> 6;
  const int hkv = Hkv == H ? h : h / (H / Hkv);
  const T* srow = stash + b * sc_sb + h * sc_sh;
  const T* vbase = vc + b * kv_sb + hkv * kv_sh;
  const int32_t* ix = idx + (int64_t)(b * H + h) * idx_sr;
  const float m = lse[(b * H + h) * 2], l = lse[(b * H + h) * 2 + 1];
  const float mu = (m == -INFINITY) ? 0.f : m;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int i0 = 0; i0 < k; i0 += RPI) {
    const int i = i0 + r;
    if (i < k) {
      const int j = ix[i];
      float s = DT<T>::to_f32(srow[j]);
      if (mask) s = DT<T>::round(s + DT<T>::to_f32(mask[b * mask_sb + j]));
      const float pj = __expf(s - mu) / l;
      float v[8];
      V8::unpack(V8::ldg(vbase + (int64_t)j * D + 8 * c), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf(pj, v[e], o[e]);
    }
  }
  // reduce the row groups of the wave (lanes with equal c), then the 4 waves
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += __shfl_xor(o[e], off, kWave);
  if (lane < LPR)
#pragma unroll
    for (int e = 0; e < 8; ++e) s_o[wave][8 * lane + e] = o[e];
  __syncthreads();
  if (tid < D) out[b * out_sb + (int64_t)h * D + tid] = DT<T>::from_f32((s_o[0][tid] + s_o[1][tid]) + (s_o[2][tid] + s_o[3][tid]));
}

// The same sum split over the kept list: grid = (S, H, B), each workgroup takes `per` consecutive entries of idx —
// equal work whatever the distribution of the kept rows over the cache — first (index, probability) pairs into LDS
// (one dependent round trip for the whole slice), then the V rows 8 row-groups in flight per lane.  Partials go to the
// workspace; the LAST workgroup of a (b, h) to arrive (ticket) adds them in split order, so the result does not depend
// on arrival order.  r02: 193 -> see HISTORY.md (DESIGN r04 §3.6) us at 40 heads x 4915 kept rows.
constexpr int kPvMaxPer = 1024;
constexpr int kPvMaxSplits = 64;

template <typename T, int D>
__global__ __launch_bounds__(256) void pv_gather_split_kernel(const T* __restrict__ stash, int64_t sc_sb, int64_t sc_sh,
                                                              const float* __restrict__ lse, const T* __restrict__ mask,
                                                              int64_t mask_sb, const T* __restrict__ vc, int64_t kv_sb,
                                                              int64_t kv_sh, const int32_t* __restrict__ idx, int64_t idx_sr,
                                                              int k, int per, float* part, unsigned* cnt,
                                                              T* __restrict__ out, int64_t out_sb, int H, int Hkv) {
  constexpr int LPR = D / 8;
  constexpr int RPI = 256 / LPR;
  constexpr int U = 8;
  __shared__ int s_j[kPvMaxPer];
  __shared__ float s_p[kPvMaxPer];
  __shared__ float s_o[4][D];
  __shared__ unsigned s_ticket;
  using V8 = Vec8<T>;
  const int split = blockIdx.x, S = gridDim.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int c = tid % LPR, r = tid / LPR, lane = tid & 63, wave = tid >> 6;
  const int hkv = Hkv == H ? h : h / (H / Hkv);
  const int unit = b * H + h;
  const T* srow = stash + b * sc_sb + h * sc_sh;
  const T* vbase = vc + b * kv_sb + hkv * kv_sh;
  const int32_t* ix = idx + (int64_t)unit * idx_sr;
  const float m = lse[unit * 2], l = lse[unit * 2 + 1];
  const float mu = (m == -INFINITY) ? 0.f : m;
  const int i_lo = split * per;
  const int n = min(per, k - i_lo);            // >= 1 by construction of the grid
  for (int t = tid; t < n; t += 256) {
    const int j = ix[i_lo + t];
    float sc = DT<T>::to_f32(srow[j]);
    if (mask) sc = DT<T>::round(sc + DT<T>::to_f32(mask[b * mask_sb + j]));
    s_j[t] = j;
    s_p[t] = __expf(sc - mu) / l;
  }
  __syncthreads();
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int i0 = 0; i0 < n; i0 += RPI * U) {
    typename V8::raw vr[U];
    float pp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * RPI + r;
      const bool ok = i < n;
      const int j = s_j[ok ? i : 0];
      pp[u] = ok ? s_p[i] : 0.f;
      vr[u] = V8::ldg(vbase + (int64_t)j * D + 8 * c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[8];
      V8::unpack(vr[u], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf(pp[u], v[e], o[e]);
    }
  }
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += __shfl_xor(o[e], off, kWave);
  if (lane < LPR)
#pragma unroll
    for (int e = 0; e < 8; ++e) s_o[wave][8 * lane + e] = o[e];
  __syncthreads();
  float tot = 0.f;
  if (tid < D) tot = (s_o[0][tid] + s_o[1][tid]) + (s_o[2][tid] + s_o[3][tid]);
  T* orow = out + b * out_sb + (int64_t)h * D;
  if (S == 1) {
    if (tid < D) orow[tid] = DT<T>::from_f32(tot);
    return;
  }
  // partials leave as write-through (agent-scope) stores; a wave's ticket is drawn only after its own stores were
  // acknowledged (vmcnt counts stores on CDNA4) — no L2 write-back / invalidate per workgroup, which a __threadfence()
  // would cost every one of the ~1000 workgroups of a launch
  float* mine = part + ((int64_t)unit * kPvMaxSplits + split) * D;
  if (tid < D) __hip_atomic_store(mine + tid, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_ticket = __hip_atomic_fetch_add(cnt + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ticket != (unsigned)(S - 1)) return;
  if (tid < D) {
    const float* q = part + (int64_t)unit * kPvMaxSplits * D + tid;
    float acc = 0.f;
    for (int s0 = 0; s0 < S; s0 += 8) {          // 8 loads in flight, added in split order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = __hip_atomic_load(q + (int64_t)min(s0 + u, S - 1) * D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (s0 + u < S) ? v[u] : 0.f;
    }
    orow[tid] = DT<T>::from_f32(acc);
  }
  if (tid == 0) __hip_atomic_store(cnt + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
}

// Layer-to-layer cascade (README.md:11; trace columns if_topk / topk): rank[h, j] = score[h, j] if the token in slot j of
// this layer is among the tokens the PREVIOUS layer kept for head h, else -inf — the window top-k then picks survivors
// of the previous layer first.  ids / prev_ids hold token ids, ascending per head: membership = binary search.
template <typename T>
__global__ __launch_bounds__(256) void cascade_rank_kernel(const T* __restrict__ score, int64_t score_sh,
                                                           const int32_t* __restrict__ ids, int64_t ids_sh,
                                                           const int32_t* __restrict__ prev_ids, int64_t prev_sh, int n_prev,
                                                           float* __restrict__ rank, int64_t rank_sh, int L) {
  const int j = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (j >= L) return;
  const int32_t want = ids[h * ids_sh + j];
  const int32_t* pv = prev_ids + h * prev_sh;
  int lo = 0, hi = n_prev;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (pv[mid] < want) lo = mid + 1; else hi = mid;
  }
  const bool member = lo < n_prev && pv[lo] == want;
  rank[h * rank_sh + j] = member ? DT<T>::to_f32(score[h * score_sh + j]) : -INFINITY;
}

}  // namespace spatten

using namespace spatten;

extern "C" int spatten_importance_accumulate(int dtype, const void* stash, int64_t sb, int64_t sh, int64_t sq,
                                             const float* lse, const void* mask, int64_t mask_sb, int64_t mask_sq,
                                             float* acc, int64_t acc_sh, int batch, int heads, int q_len, int kv_len,
                                             int causal, void* stream) {
  if (!ok_dtype(dtype) || !stash || !acc || !lse || batch <= 0 || heads <= 0 || q_len <= 0 || kv_len <= 0)
    return SPATTEN_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)ceil_div(kv_len, 256), (unsigned)heads);
  SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((importance_accumulate_kernel<T>), grid, dim3(256), 0, st, (const T*)stash, sb, sh,
                                             sq, lse, (const T*)mask, mask_sb, mask_sq, acc, acc_sh, batch, heads, q_len,
                                             kv_len, causal));
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_row_lse(int dtype, const void* stash, int64_t sb, int64_t sh, int64_t sq, const void* mask,
                               int64_t mask_sb, int64_t mask_sq, float* lse, int batch, int heads, int q_len,
                               int kv_len, int causal, void* stream) {
  if (!ok_dtype(dtype) || !stash || !lse || batch <= 0 || heads <= 0 || q_len <= 0 || kv_len <= 0) return SPATTEN_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)q_len, (unsigned)heads, (unsigned)batch);
  SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((row_lse_kernel<T>), grid, dim3(256), 0, st, (const T*)stash, sb, sh, sq,
                                             (const T*)mask, mask_sb, mask_sq, lse, heads, q_len, kv_len, causal));
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_importance_compact(const float* src, int64_t src_sh, float* dst, int64_t dst_sh,
                                          const int32_t* idx, int64_t idx_sh, int heads, int start, int k, int tail_lo,
                                          int tail_len, void* stream) {
  if (!src || !dst || (k > 0 && !idx) || heads <= 0 || start < 0 || k < 0 || tail_len < 0) return SPATTEN_ERR_INVALID;
  const int Lp = start + k + tail_len;
  if (Lp == 0) return SPATTEN_OK;
  hipLaunchKernelGGL(importance_compact_kernel, dim3((unsigned)ceil_div(Lp, 256), (unsigned)heads), dim3(256), 0,
                     (hipStream_t)stream, src, src_sh, dst, dst_sh, idx, idx_sh, start, k, tail_lo, Lp);
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_cascade_rank(int dtype, const void* score, int64_t score_sh, const int32_t* ids, int64_t ids_sh,
                                    const int32_t* prev_ids, int64_t prev_sh, int n_prev, float* rank, int64_t rank_sh,
                                    int heads, int len, void* stream) {
  if (!ok_dtype(dtype) || !score || !ids || !prev_ids || !rank || heads <= 0 || len <= 0 || n_prev < 0) return SPATTEN_ERR_INVALID;
  const dim3 grid((unsigned)ceil_div(len, 256), (unsigned)heads);
  SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((cascade_rank_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)score,
                                             score_sh, ids, ids_sh, prev_ids, prev_sh, n_prev, rank, rank_sh, len));
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_head_scores(int dtype, const void* out, int64_t out_sb, int64_t out_sq, float* scores,
                                   int batch, int q_len, int heads, int head_dim, void* stream) {
  if (!ok_dtype(dtype) || !out || !scores || batch <= 0 || q_len <= 0 || heads <= 0 || head_dim <= 0) return SPATTEN_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((head_scores_kernel<T>), dim3((unsigned)heads), dim3(256), 0, st, (const T*)out,
                                             out_sb, out_sq, scores, batch, q_len, head_dim));
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

static inline size_t pv_cnt_bytes(int units) { return (((size_t)units * sizeof(unsigned)) + 255) / 256 * 256; }

extern "C" size_t spatten_pv_gather_workspace_bytes(int batch, int heads, int head_dim) {
  if (batch <= 0 || heads <= 0 || head_dim <= 0) return 0;
  return pv_cnt_bytes(batch * heads) + (size_t)batch * heads * kPvMaxSplits * head_dim * sizeof(float);
}

extern "C" int spatten_pv_gather(int dtype, const void* stash, int64_t sc_sb, int64_t sc_sh, const float* lse,
                                 const void* mask, int64_t mask_sb, const void* v_cache, int64_t kv_sb, int64_t kv_sh,
                                 const int32_t* idx, int64_t idx_sr, int k, void* out, int64_t out_sb, int batch,
                                 int heads, int kv_heads, int head_dim, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (!ok_dtype(dtype) || !stash || !lse || !v_cache || !idx || !out || k <= 0 || batch <= 0 || heads <= 0 ||
      kv_heads <= 0 || heads % kv_heads)
    return SPATTEN_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int units = batch * heads;
  // slices of the kept list: about four workgroups per CU over the whole launch, at least 64 rows each
  int per = ceil_div((int)(((long long)k * units + 1023) / 1024), 16) * 16;
  if (per < 64) per = 64;
  if (per > kPvMaxPer) per = kPvMaxPer;    // the slice lives in LDS: many units (B*H large) just get more workgroups than 4 per CU
  if (per < ceil_div(k, kPvMaxSplits)) per = ceil_div(k, kPvMaxSplits);
  int S = ceil_div(k, per);
  if (!workspace || S == 1 || per > kPvMaxPer) {
    // per > kPvMaxPer only when k > kPvMaxPer * kPvMaxSplits (65536 kept rows per head): the one-workgroup kernel below
    // has no such bound (it walks the kept list in LDS-sized pieces)
    S = 1;
  } else if (workspace_bytes < spatten_pv_gather_workspace_bytes(batch, heads, head_dim)) {
    return SPATTEN_ERR_INVALID;
  }
  if (S == 1 && k > kPvMaxPer) {           // no workspace: one workgroup per (b, h)
    const dim3 grid((unsigned)heads, (unsigned)batch);
#define SPATTEN_PV(DD)                                                                                                  \
  SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((pv_gather_kernel<T, DD>), grid, dim3(256), 0, st, (const T*)stash, sc_sb, sc_sh, \
                                             lse, (const T*)mask, mask_sb, (const T*)v_cache, kv_sb, kv_sh, idx, idx_sr, k,    \
                                             (T*)out, out_sb, heads, kv_heads))
    if (head_dim == 128) { SPATTEN_PV(128); } else { SPATTEN_PV(64); }
#undef SPATTEN_PV
    return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
  }
  if (S == 1) per = k;
  unsigned* cnt = (unsigned*)workspace;
  float* part = workspace ? (float*)((char*)workspace + pv_cnt_bytes(units)) : nullptr;
  const dim3 grid((unsigned)S, (unsigned)heads, (unsigned)batch);
#define SPATTEN_PVS(DD)                                                                                                 \
  SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((pv_gather_split_kernel<T, DD>), grid, dim3(256), 0, st, (const T*)stash, sc_sb, \
                                             sc_sh, lse, (const T*)mask, mask_sb, (const T*)v_cache, kv_sb, kv_sh, idx,       \
                                             idx_sr, k, per, part, cnt, (T*)out, out_sb, heads, kv_heads))
  if (head_dim == 128) { SPATTEN_PVS(128); } else { SPATTEN_PVS(64); }
#undef SPATTEN_PVS
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}
