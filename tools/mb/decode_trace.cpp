// Per-workgroup phase timestamps of the decode kernel (needs a library built with -DSPATTEN_TRACE).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/spatten.h"
extern "C" int spatten_debug_set_trace(unsigned long long* buf);
int main(int argc, char** argv) {
  const int B = 1, N = argc > 1 ? atoi(argv[1]) : 2048, ns = argc > 2 ? atoi(argv[2]) : 8;
  const int H = 32, d = 128, L = 8;
  const size_t row = (size_t)d * 2, per = (size_t)B * H * (N + 128) * row;
  std::vector<void*> kr(L), v(L);
  for (int l = 0; l < L; ++l) { (void)hipMalloc(&kr[l], per); (void)hipMalloc(&v[l], per); (void)hipMemset(kr[l], 0x3c, per); (void)hipMemset(v[l], 0x3c, per); }
  void *q, *cos, *sin, *out, *sc, *ws;
  (void)hipMalloc(&q, B * H * row); (void)hipMemset(q, 0x3c, B * H * row);
  (void)hipMalloc(&cos, (N + 128) * row / 2); (void)hipMemset(cos, 0x3c, (N + 128) * row / 2);
  (void)hipMalloc(&sin, (N + 128) * row / 2); (void)hipMemset(sin, 0x3c, (N + 128) * row / 2);
  (void)hipMalloc(&out, B * H * row); (void)hipMalloc(&sc, (size_t)B * H * (N + 128) * 2);
  size_t wsb = spatten_decode_workspace_bytes(B, H, d, 64);
  (void)hipMalloc(&ws, wsb); (void)hipMemset(ws, 0, wsb);
  const int WG = ns * H;
  unsigned long long* tr; (void)hipMalloc(&tr, (size_t)L * WG * 8 * 8);
  (void)hipMemset(tr, 0, (size_t)L * WG * 8 * 8);
  auto go = [&](bool trace) {
    for (int l = 0; l < L; ++l) {
      if (trace) spatten_debug_set_trace(tr + (size_t)l * WG * 8); else spatten_debug_set_trace(nullptr);
      spatten_attn_decode(SPATTEN_BF16, q, (int64_t)H * d, d, nullptr, kr[l], v[l], (int64_t)H * (N + 128) * d,
                          (int64_t)(N + 128) * d, nullptr, nullptr, 0, 0, cos, sin, N + 128, nullptr, 0, nullptr, 0, out,
                          (int64_t)H * d, sc, (int64_t)H * (N + 128), N + 128, nullptr, ws, B, H, H, d, N, N - 1, ns, nullptr);
    }
  };
  go(false); go(false); (void)hipDeviceSynchronize();
  go(true); (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)L * WG * 8);
  (void)hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
  // layer L-1 (steady state): times relative to the earliest workgroup start, in units of the cycle counter
  const unsigned long long* t = h.data() + (size_t)(L - 1) * WG * 8;
  // cycle counters are per-XCD: only per-workgroup DELTAS are meaningful
  struct { const char* name; int from, to; } ph[] = {{"start->loads issued", 0, 5}, {"issued->q rotated", 5, 6},
      {"q rotated->scores(last tile)", 6, 7}, {"scores->loop done", 7, 1}, {"start->loop done", 0, 1},
      {"loop->wg reduced", 1, 2}, {"reduced->ticket", 2, 3}, {"ticket->merge done", 3, 4}};
  for (auto& q_ : ph) {
    std::vector<double> x;
    for (int w = 0; w < WG; ++w) if (t[w * 8 + q_.to] && t[w * 8 + q_.from]) x.push_back((double)(t[w * 8 + q_.to] - t[w * 8 + q_.from]));
    if (x.empty()) continue;
    std::sort(x.begin(), x.end());
    printf("%-30s n=%4zu  min %7.0f  median %7.0f  p90 %7.0f  max %7.0f cycles\n", q_.name, x.size(), x[0], x[x.size() / 2], x[x.size() * 9 / 10], x.back());
  }
  return 0;
}
