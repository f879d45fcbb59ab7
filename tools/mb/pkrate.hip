// VALU issue rate of packed-f32 instructions on gfx950: cycles per instruction for one wave per SIMD and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mb/pkrate tools/mb/pkrate.hip && tools/mb/pkrate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float c = 1.0001f; const f2 pc = {1.0001f, 0.9999f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {        // 8 independent v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 1) { // 4 independent v_pk_mul_f32 (the same 8 multiplies)
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
    } else if (MODE == 2) { // 8 v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 3) { // 4 v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
    } else if (MODE == 4) { // 8 v_exp_f32
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 5) { // 8 v_cvt_pk_bf16_f32
      asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                   "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 6) { // 4 v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int threads, int n_instr) {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 4096;
  k<MODE><<<256, threads>>>(out, cyc, iters); hipDeviceSynchronize();
  k<MODE><<<256, threads>>>(out, cyc, iters); hipDeviceSynchronize();   // (64 threads = one wave on one SIMD; 1024 = four per SIMD)
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-22s %4d threads/WG: %.2f s_memtime ticks per instruction per wave (%d instr per iteration)\n", name, threads, (double)c / iters / n_instr, n_instr);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int th : {64, 256, 512, 1024}) {
    run<0>("v_mul_f32 x8", th, 8); run<1>("v_pk_mul_f32 x4", th, 4); run<2>("v_fma_f32 x8", th, 8); run<3>("v_pk_fma_f32 x4", th, 4);
    run<6>("v_pk_add_f32 x4", th, 4); run<4>("v_exp_f32 x8", th, 8); run<5>("v_cvt_pk_bf16_f32 x8", th, 8);
  }
  return 0;
}
