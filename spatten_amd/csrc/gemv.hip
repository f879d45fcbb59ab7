// gemv.hip — the projections of a single-token step (modify_llama.py:72-74 q/k/v_proj, :163 o_proj) as a weight-streaming
// kernel:  y[n] = sum_k x[k] * W[n, k] (+ bias[n]),  W [N, K] row-major (nn.Linear's layout), fp32 accumulation, one
// rounding to the model dtype at the end (what torch's linear does).
//
// Why it is here.  SURVEY §8 rows A2 / A10 leave the projections to torch, and for multi-token forwards they stay
// there (MFMA GEMMs, hipBLASLt).  But at q_len = 1 a projection is a 33.5 MB stream (4096 x 4096 bf16) with 2 FLOP per
// weight — HBM-bound like the attention kernel next to it — and the plugin's decode step moves 4 of them per layer:
// 4.3 GB per token at Llama-2-7B against 1.1 GB of K/V.  torch's GEMV path measured 13 us per matrix (2.6 TB/s) inside
// the captured decode step (profiles/r03_plugin_graph_kernel_stats.csv), i.e. four fifths of the token.
//
// Mapping: 256-thread workgroups, every wave owns R = 4 weight rows; a pass covers 8 x 512 columns: lane l holds
// columns [512 c + 8 l, +8) of chunk c — 16-byte loads of fully used 128-byte lines, 32 of them in flight per lane
// before the first use (one wave per SIMD, 128 KB in flight per CU: the decode kernel's single-shot recipe), x from L2.
// Products on the packed-dot units (v_dot2c_f32_bf16 / v_dot2_f32_f16), lanes reduced with DPP + permlane swaps.
#include "common.h"

namespace spatten {

#ifndef SPATTEN_GEMV_ROWS
#define SPATTEN_GEMV_ROWS 4
#endif
#ifndef SPATTEN_GEMV_NT
#define SPATTEN_GEMV_NT 1         // weights with the non-temporal policy (A/B switch)
#endif
constexpr int kGemvRows = SPATTEN_GEMV_ROWS;      // weight rows per wave
constexpr int kGemvChunks = 8;    // 512-column chunks per pass

template <typename T>
__global__ __launch_bounds__(256) void gemv_kernel(const T* __restrict__ x, const T* __restrict__ W, int64_t w_sn,
                                                   const T* __restrict__ bias, T* __restrict__ y, int N, int K) {
  using V8 = Vec8<T>;
  using raw_t = typename V8::raw;
  using D8 = Dot8<T>;
  constexpr int R = kGemvRows, C = kGemvChunks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * R;
  if (n0 >= N) return;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  const T* wrow[R];
#pragma unroll
  for (int r = 0; r < R; ++r) wrow[r] = W + (int64_t)min(n0 + r, N - 1) * w_sn;   // tail rows re-read the last row

  for (int k0 = 0; k0 < K; k0 += C * 512) {
    raw_t xr[C], wr[R][C];
    bool live[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {          // columns past K: clamped address, the x piece is zeroed below
      const int col = k0 + c * 512 + lane * 8;
      live[c] = col < K;
      xr[c] = V8::ldg(x + min(col, K - 8));
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < C; ++c) wr[r][c] = SPATTEN_GEMV_NT ? V8::ldg_stream(wrow[r] + min(k0 + c * 512 + lane * 8, K - 8)) : V8::ldg(wrow[r] + min(k0 + c * 512 + lane * 8, K - 8));
    __builtin_amdgcn_sched_barrier(0);     // every load of the pass is issued before the first product
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (!live[c]) {
        float z[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = 0.f;
        xr[c] = V8::pack(z);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < C; ++c) acc[r] = D8::dot(xr[c], wr[r][c], acc[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
  if (lane < R && n0 + lane < N) {
    float v = acc[0];
#pragma unroll
    for (int r = 1; r < R; ++r) v = (lane == r) ? acc[r] : v;
    if (bias) v += DT<T>::to_f32(bias[n0 + lane]);
    y[n0 + lane] = DT<T>::from_f32(v);
  }
}

}  // namespace spatten

using namespace spatten;

namespace spatten {
int gemv_rows(int dtype, const void* x, int64_t x_sm, const void* W, int64_t w_sn, const void* bias, void* y, int64_t y_sm,
              int M, int N, int K, hipStream_t stream) {
  return spatten_gemv(dtype, x, x_sm, W, w_sn, bias, y, y_sm, M, N, K, (void*)stream);
}
}  // namespace spatten

extern "C" int spatten_gemv(int dtype, const void* x, int64_t x_sm, const void* W, int64_t w_sn, const void* bias, void* y,
                            int64_t y_sm, int M, int N, int K, void* stream) {
  if (!ok_dtype(dtype) || !x || !W || !y || M <= 0 || N <= 0 || K <= 0 || w_sn < K) return SPATTEN_ERR_INVALID;
  if (K % 8 != 0 || w_sn % 8 != 0 || x_sm % 8 != 0) return SPATTEN_ERR_UNSUPPORTED;   // 16-byte pieces
  const dim3 grid((unsigned)ceil_div(N, 4 * kGemvRows));
  const size_t es = dtype == SPATTEN_F32 ? 4 : 2;
  for (int m = 0; m < M; ++m) {            // a batch of single-token rows: one stream of W per row (B = 1 is the path)
    const char* xm = (const char*)x + (size_t)m * x_sm * es;
    char* ym = (char*)y + (size_t)m * y_sm * es;
    SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((gemv_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)xm,
                                               (const T*)W, w_sn, (const T*)bias, (T*)ym, N, K));
  }
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}
