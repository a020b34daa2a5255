// Issue cost of the non-FMA VALU ops k_residual uses (gfx950).  Each mode runs 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed, int iseed) {
  float a[8]; int n[8]; double d[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x * 0.37f; n[i] = iseed + i + threadIdx.x; d[i] = a[i]; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) a[i] = a[i] + 1.000001f;                                  // v_add_f32
      if (MODE == 1) n[i] = n[i] + (n[i] >> 3);                                // v_lshr + v_add_u32 (2 int ops)
      if (MODE == 2) a[i] = (a[i] > 3.0f) ? a[i] - 2.5f : a[i] + 0.5f;         // cmp + 2 add + cndmask (4)
      if (MODE == 3) a[i] = truncf(a[i]) + 0.3f;                               // trunc + add (2)
      if (MODE == 4) a[i] = (float)((int)a[i]) + 0.3f;                         // cvt_i32_f32 + cvt_f32_i32 + add (3)
      if (MODE == 5) n[i] = __builtin_amdgcn_sbfe(n[i], 3, 9) + it;            // v_bfe_i32 + add (2)
      if (MODE == 6) d[i] = (double)(float)d[i] * 1.0000001;                   // cvt_f32_f64 + cvt_f64_f32 + mul_f64 (3)
      if (MODE == 7) n[i] = min(n[i] * 3 + 1, 1 << 30);                        // mad_u32 (or mul_lo) + min (2)
      if (MODE == 8) n[i] = __umul24(n[i], 5) + 7;              // v_mad_u32_u24 (1)
      if (MODE == 9) a[i] = fminf(a[i] * 1.0001f, 5.0f);                       // mul + min (2)
      if (MODE == 10) d[i] = d[i] + (double)n[i];                              // cvt_f64_i32 + add_f64 (2)
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + (float)n[i] + (float)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int instr) {
  float* out; (void)hipMalloc(&out, 256 * 2048 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 2048;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 3);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * 4 * ITERS * 8);
  printf("%-44s %.3f ms -> %.2f cycles(@2.4GHz) per chain step (%d instr expected => %.2f each)\n", name, ms, cyc, instr, cyc / instr);
  (void)hipFree(out);
}
int main() {
  run<0>("v_add_f32", 1); run<1>("lshr+add_u32", 2); run<2>("cmp+2add+cndmask", 4); run<3>("trunc+add", 2);
  run<4>("cvt_i32_f32+cvt_f32_i32+add", 3); run<5>("bfe_i32+add", 2); run<6>("cvt_f32_f64+cvt_f64_f32+mul_f64", 3);
  run<7>("mul_lo/mad+min_i32", 2); run<8>("mad_u32_u24", 1); run<9>("mul_f32+min_f32", 2); run<10>("cvt_f64_i32+add_f64", 2);
  return 0;
}
