// mfma_valu.hip — how the two waves of a SIMD share its issue port and pipes on gfx950 (r05).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mb/mfma_valu tools/mb/mfma_valu.hip && tools/mb/mfma_valu
// Workgroup = 8 waves (two per SIMD), one workgroup per CU, 256 workgroups.  Waves 0-3 ("A") run a chain of
// v_mfma_f32_32x32x16_bf16 over four rotating accumulators with KA vector instructions of type TA behind each MFMA; waves
// 4-7 ("B", the SIMD partners) run a stream of type TB (with GAP s_nop states between instructions), or the same loop as A
// (TB = SAME), or nothing (TB = NONE), until every A wave of the workgroup is done.  Printed: cycles per MFMA of the A
// waves (s_memtime), and how many B instructions were issued per A-MFMA.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

enum { NONE = 0, FMA, EXP, CVT, AND, DOT2, PKMUL, MAX3, DSRD, SAME, MIX, CVT0, RINGRD };

template <int T>
__device__ inline void filler(float (&x)[8], int j, unsigned lds_addr) {
  float& r = x[j & 7];
  float& q = x[(j + 3) & 7];
  if (T == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(q));
  else if (T == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
  else if (T == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(q));
  else if (T == CVT0) asm volatile("v_cvt_pk_bf16_f32 %0, 0, %0" : "+v"(r));
  else if (T == AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r));
  else if (T == DOT2) asm volatile("v_dot2_f32_bf16 %0, %1, %1, %0" : "+v"(r) : "v"(q));
  else if (T == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&x[(j & 3) * 2])) : "v"(*reinterpret_cast<double*>(&x[((j + 1) & 3) * 2])));
  else if (T == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r) : "v"(q));
  else if (T == DSRD) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 t;
    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lds_addr));
  } else if (T == MIX) {   // the flash kernel's softmax per logit (cvt, mul, cvt, fma, exp, add, half a packed cvt) as a chain on ONE
    // register, eight registers interleaved, constants chosen so that the values stay O(1) (the first version of this mix
    // drifted into inf / NaN / denormals and ran ~200 cycles per instruction: not an issue-port effect)
    float& rr = x[(j / 7) & 7];
    const float c0 = 0.99f, c1 = 0.3f, c2 = 0.2f, c3 = -0.9f;
    switch (j % 7) {
      case 0: asm volatile("v_cvt_pk_bf16_f32 %0, 0, %0" : "+v"(rr)); break;
      case 1: asm volatile("v_mul_f32 %0, %0, %1" : "+v"(rr) : "v"(c0)); break;
      case 2: asm volatile("v_cvt_pk_bf16_f32 %0, 0, %0" : "+v"(rr)); break;
      case 3: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(rr) : "v"(c1), "v"(c2)); break;
      case 4: asm volatile("v_exp_f32 %0, %0" : "+v"(rr)); break;
      case 5: asm volatile("v_add_f32 %0, %0, %1" : "+v"(rr) : "v"(c3)); break;
      default: asm volatile("v_cvt_pk_bf16_f32 %0, 0, %0" : "+v"(rr)); break;
    }
  }
}

template <int GAP> __device__ inline void gap() {
  if (GAP == 1) asm volatile("s_nop 0");
  if (GAP == 2) asm volatile("s_nop 1");
  if (GAP == 4) asm volatile("s_nop 3");
  if (GAP == 8) asm volatile("s_nop 7");
}

// the shipped matrix phase: every MFMA takes its A operand from a KA-deep register ring fed by ds_read_b128 (one read behind
// every MFMA, s_waitcnt lgkmcnt(KA - 1) in front of it) — swizzled, lane-distinct addresses over 32 KB of LDS
template <int KA>
__device__ inline void a_ring_iter(f32x16 (&acc)[4], bf8 b, unsigned lds_addr) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 ring[KA];
#pragma unroll
  for (int i = 0; i < KA; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[i]) : "v"(lds_addr), "i"((i * 1024) & 0x7FFF));
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    if (KA == 2) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
    if (KA == 4) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
    if (KA == 8) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(ring[i % KA]), "v"(b));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[i % KA]) : "v"(lds_addr), "i"((((i + KA) * 1024) ^ ((i & 3) * 4096)) & 0x7FF0));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int KA, int TA>
__device__ inline void a_iter(f32x16 (&acc)[4], bf8 a, bf8 b, float (&x)[8], unsigned lds_addr) {
  if constexpr (TA == RINGRD) { a_ring_iter<KA>(acc, b, lds_addr); return; }
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
    for (int j = 0; j < KA; ++j) filler<TA>(x, i * KA + j, lds_addr);
  }
  if (TA == DSRD) asm volatile("s_waitcnt lgkmcnt(0)");
}

template <int KA, int TA, int TB, int GAP, int PRIO, int SWAP = 0>
__global__ __launch_bounds__(512, 1) void k(int iters, unsigned long long* cyc, unsigned long long* bcount, float* sink) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  __shared__ int done;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (threadIdx.x == 0) done = 0;
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = 1.0f + 0.001f * (lane + j);
  const unsigned lds_addr = (unsigned)((((lane & 31) * 256) + (((lane >> 5) ^ (lane & 15)) << 4)) & 0x7FF0);
  f32x16 acc[4];
  bf8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const bool role_a = SWAP ? wave >= 4 : wave < 4;
  if (PRIO == 1 && role_a) __builtin_amdgcn_s_setprio(1);
  if (PRIO == 2 && !role_a) __builtin_amdgcn_s_setprio(1);
  if (role_a || TB == SAME) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) a_iter<KA, TA>(acc, a, b, x, lds_addr);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + (SWAP ? (wave ^ 4) : wave)] = t1 - t0;
    if (role_a && lane == 0) atomicAdd(&done, 1);
  } else if (TB != NONE) {
    unsigned long long n = 0;
    while (true) {
#pragma unroll
      for (int j = 0; j < 256; ++j) { filler<TB>(x, j, lds_addr); gap<GAP>(); }
      if (TB == DSRD) asm volatile("s_waitcnt lgkmcnt(0)");
      n += 256;
      if (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 4) break;
    }
    if (lane == 0) bcount[blockIdx.x * 8 + (SWAP ? (wave ^ 4) : wave)] = n;
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int KA, int TA, int TB, int GAP, int PRIO, int SWAP = 0>
void run(const char* name) {
  const int iters = 200, G = 256;
  unsigned long long *cyc, *bc;
  float* sink;
  hipMalloc(&cyc, G * 8 * 8);
  hipMalloc(&bc, G * 8 * 8);
  hipMalloc(&sink, 4096);
  hipMemset(cyc, 0, G * 8 * 8);
  hipMemset(bc, 0, G * 8 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KA, TA, TB, GAP, PRIO, SWAP><<<G, 512>>>(20, cyc, bc, sink);
  hipMemset(bc, 0, G * 8 * 8);
  hipEventRecord(e0);
  k<KA, TA, TB, GAP, PRIO, SWAP><<<G, 512>>>(iters, cyc, bc, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(G * 8), hb(G * 8);
  hipMemcpy(h.data(), cyc, G * 8 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), bc, G * 8 * 8, hipMemcpyDeviceToHost);
  double sa = 0, sb = 0, nb = 0;
  for (int g = 0; g < G; ++g)
    for (int w = 0; w < 8; ++w) {
      if (w < 4) sa += h[g * 8 + w];
      else { sb += h[g * 8 + w]; nb += hb[g * 8 + w]; }
    }
  const double mf = (double)iters * 64;
  const double tf = 2.0 * 32 * 32 * 16 * mf * (TB == SAME ? 8 : 4) * G / (ms * 1e-3) / 1e12;
  printf("%-58s A %6.1f cyc/MFMA", name, sa / (G * 4) / mf);
  if (TB == SAME) printf("  B %6.1f cyc/MFMA", sb / (G * 4) / mf);
  else if (TB != NONE) printf("  B issued %5.2f per A-MFMA", nb / (G * 4) / mf);
  printf("   %7.3f ms  %6.0f TFLOP/s  clk %.2f GHz\n", ms, tf, sa / (G * 4) / (ms * 1e-3) / 1e9);
  hipFree(cyc); hipFree(bc); hipFree(sink);
}

int main() {
  printf("== A alone (B exits)\n");
  run<0, NONE, NONE, 0, 0>("A bare MFMA chain, B none");
  run<2, FMA, NONE, 0, 0>("A + 2 fma per MFMA, B none");
  run<4, FMA, NONE, 0, 0>("A + 4 fma, B none");
  run<5, FMA, NONE, 0, 0>("A + 5 fma, B none");
  run<6, FMA, NONE, 0, 0>("A + 6 fma, B none");
  run<7, FMA, NONE, 0, 0>("A + 7 fma, B none");
  run<8, FMA, NONE, 0, 0>("A + 8 fma, B none");
  run<10, FMA, NONE, 0, 0>("A + 10 fma, B none");
  run<7, MIX, NONE, 0, 0>("A + 7 softmax-mix, B none");
  run<14, MIX, NONE, 0, 0>("A + 14 softmax-mix (2 logits per MFMA), B none");
  run<5, DOT2, NONE, 0, 0>("A + 5 dot2, B none");
  run<5, EXP, NONE, 0, 0>("A + 5 exp, B none");
  run<5, CVT0, NONE, 0, 0>("A + 5 cvt_pk(0,x), B none");
  run<3, PKMUL, NONE, 0, 0>("A + 3 pk_mul, B none");
  run<2, DSRD, NONE, 0, 0>("A + 2 ds_read_b128, B none");
  printf("== A bare, B a vector stream (the shipped ping-pong: matrix phase beside vector phase)\n");
  run<0, NONE, FMA, 0, 0>("A bare, B fma dense");
  run<0, NONE, FMA, 1, 0>("A bare, B fma + s_nop 0");
  run<0, NONE, FMA, 4, 0>("A bare, B fma + s_nop 3");
  run<0, NONE, FMA, 8, 0>("A bare, B fma + s_nop 7");
  run<0, NONE, EXP, 0, 0>("A bare, B exp dense");
  run<0, NONE, CVT, 0, 0>("A bare, B cvt_pk dense");
  run<0, NONE, AND, 0, 0>("A bare, B and dense");
  run<0, NONE, DOT2, 0, 0>("A bare, B dot2 dense");
  run<0, NONE, PKMUL, 0, 0>("A bare, B pk_mul dense");
  run<0, NONE, MAX3, 0, 0>("A bare, B max3 dense");
  run<0, NONE, MIX, 0, 0>("A bare, B softmax-mix dense");
  run<0, NONE, MIX, 0, 1>("A bare prio1, B softmax-mix dense");
  run<0, NONE, MIX, 0, 2>("A bare, B softmax-mix dense prio1");
  run<0, NONE, DSRD, 0, 0>("A bare, B ds_read_b128 dense");
  run<2, DSRD, MIX, 0, 0>("A + 2 ds_read (the shipped matrix phase), B softmax-mix");
  run<2, DSRD, MIX, 0, 1>("A + 2 ds_read prio1, B softmax-mix");
  printf("== both waves of a SIMD run MFMA + fillers (lock-step, software-pipelined softmax)\n");
  run<0, NONE, SAME, 0, 0>("both bare MFMA");
  run<5, FMA, SAME, 0, 0>("both MFMA + 5 fma");
  run<7, MIX, SAME, 0, 0>("both MFMA + 7 softmax-mix");
  run<8, MIX, SAME, 0, 0>("both MFMA + 8 softmax-mix");
  run<10, MIX, SAME, 0, 0>("both MFMA + 10 softmax-mix");
  run<14, MIX, SAME, 0, 0>("both MFMA + 14 softmax-mix");
  printf("== A = MFMA + own fillers, B = LDS reads only (the guide's compute / load role split)\n");
  run<5, MIX, DSRD, 0, 0>("A + 5 mix, B ds_read dense");
  run<7, MIX, DSRD, 0, 0>("A + 7 mix, B ds_read dense");
  run<7, MIX, DSRD, 4, 0>("A + 7 mix, B ds_read + s_nop 3");
  run<10, MIX, DSRD, 4, 0>("A + 10 mix, B ds_read + s_nop 3");
  printf("== r05b: the shipped matrix phase (operand ring from LDS) beside the partner's softmax; which half is older\n");
  run<4, RINGRD, NONE, 0, 0>("A ring-4 from LDS, B none");
  run<8, RINGRD, NONE, 0, 0>("A ring-8 from LDS, B none");
  run<4, RINGRD, FMA, 0, 0>("A ring-4 (older half), B fma dense");
  run<4, RINGRD, MIX, 0, 0>("A ring-4 (older half), B softmax mix");
  run<4, RINGRD, MIX, 0, 0, 1>("A ring-4 (YOUNGER half), B softmax mix (older)");
  run<4, RINGRD, MIX, 0, 1, 1>("A ring-4 (younger, prio 1), B softmax mix (older)");
  run<8, RINGRD, MIX, 0, 0, 1>("A ring-8 (younger half), B softmax mix (older)");
  run<2, RINGRD, MIX, 0, 0, 1>("A ring-2 (younger half), B softmax mix (older)");
  run<0, NONE, MIX, 0, 0, 1>("A bare (younger half), B softmax mix (older)");
  run<0, NONE, MIX, 0, 0>("A bare (older half), B softmax mix");
  run<4, RINGRD, SAME, 0, 0>("both ring-4 MFMA (lock-step matrix phases)");
  run<7, MIX, SAME, 0, 0>("both MFMA + 7 stable softmax mix");
  run<8, MIX, SAME, 0, 0>("both MFMA + 8 stable softmax mix");
  run<7, MIX, NONE, 0, 0>("A + 7 stable softmax mix alone");
  return 0;
}
