// VALU issue-rate microbenchmark for gfx950: plain f32 FMA, packed f32 FMA, f64 FMA, f32 mul, cvt f32->f64, IEEE div.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[8]; double d[8]; float2_ p[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; d[i] = a[i]; p[i] = float2_{a[i], a[i] + 1.f}; }
  const float m = 1.0000001f; const double md = 1.0000001; const float2_ mp = {m, m};
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, 0.5f);
      if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], mp, mp);
      if (MODE == 2) d[i] = __builtin_fma(d[i], md, 0.5);
      if (MODE == 3) a[i] = a[i] * m;
      if (MODE == 4) d[i] = d[i] + (double)a[i];   // cvt + dadd
      if (MODE == 5) a[i] = seed / a[i];            // IEEE division
      if (MODE == 6) a[i] = __builtin_amdgcn_rcpf(a[i]);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + (float)d[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int ops_per_iter_instr) {
  float* out; hipMalloc(&out, 256 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {1024, 2048}) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * ITERS * 8 * ops_per_iter_instr;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal: 1024 SIMDs
    double cyc = ms * 1e-3 * 2.4e9 * 1024 / wave_instr;
    printf("%-22s blocks %4d: %.3f ms  -> %.2f cycles(@2.4GHz)/wave-instr/SIMD\n", name, blocks, ms, cyc);
  }
  hipFree(out);
}
int main() {
  run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 1); run<2>("v_fma_f64", 1); run<3>("v_mul_f32", 1);
  run<4>("cvt_f64_f32+add_f64", 2); run<5>("IEEE div f32 (~11 instr)", 1); run<6>("v_rcp_f32", 1);
  return 0;
}
