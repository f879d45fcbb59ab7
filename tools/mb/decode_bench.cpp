// C++ harness timing spatten_attn_decode back-to-back over L layers (developer microbenchmark).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/spatten.h"
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1, N = argc > 2 ? atoi(argv[2]) : 2048, ns = argc > 3 ? atoi(argv[3]) : 0;
  const int stash = argc > 4 ? atoi(argv[4]) : 1;
  const int L = argc > 5 ? atoi(argv[5]) : 32;   // layers rotated over: 32 x 34 MB streams from HBM, 4 x 34 MB stays in the 256 MB Infinity Cache
  const int H = 32, d = 128;
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
  auto go = [&]() {
    for (int l = 0; l < L; ++l) {
      int rc = spatten_attn_decode(SPATTEN_BF16, q, (int64_t)H * d, d, nullptr, kr[l], v[l], (int64_t)H * (N + 128) * d,
                                   (int64_t)(N + 128) * d, nullptr, nullptr, 0, 0, cos, sin, N + 64, nullptr, 0, nullptr, 0,
                                   out, (int64_t)H * d, stash ? sc : nullptr, (int64_t)H * (N + 128), N + 64, nullptr, ws, B, H, H, d,
                                   N, N - 1, ns, nullptr);
      if (rc) { printf("rc=%d\n", rc); exit(1); }
    }
  };
  go(); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) go();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / (reps * L), bytes = 2.0 * B * H * N * row;
  printf("B=%d N=%d ns=%d stash=%d layers=%d: %.2f us/launch  %.2f TB/s\n", B, N, ns, stash, L, us, bytes / us / 1e6);
  return 0;
}
