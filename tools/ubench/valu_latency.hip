// Dependent-issue behaviour of the VALU at the occupancy k_residual runs at (4 waves/SIMD) and below:
// cycles per instruction per SIMD for f32 / f64 FMA chains of ILP 1..8, and for an f32/f64 mix.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 2048;
template <int ILP, int KIND>  // KIND 0: f32 fma, 1: f64 fma, 2: alternate 1 f64 : 3 f32
__global__ __launch_bounds__(256) void k(float* out, float seed, int lds_words) {
  extern __shared__ float lds[];
  float a[8]; double d[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x * 0.01f; d[i] = a[i]; }
  const float m = 1.0000001f, c = 1e-7f;
  const double md = 1.0000001, cd = 1e-7;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < 8 / ILP; r++)
#pragma unroll
      for (int i = 0; i < ILP; i++) {
        if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, c);
        if (KIND == 1) d[i] = __builtin_fma(d[i], md, cd);
        if (KIND == 2) { if ((r * ILP + i) % 4 == 0) d[i] = __builtin_fma(d[i], md, cd); else a[i] = __builtin_fmaf(a[i], m, c); }
      }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + (float)d[i];
  if (lds_words < 0) lds[threadIdx.x] = s;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ILP, int KIND> void run(const char* name, int lds_bytes, int waves_per_simd) {
  float* out; (void)hipMalloc(&out, 256 * 4096 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * waves_per_simd * 4;  // whole rounds of resident blocks: 256 CUs x (waves_per_simd) blocks of 4 waves
  (void)hipFuncSetAttribute((const void*)k<ILP, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(256), lds_bytes, 0, out, 1.5f, 0);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(256), lds_bytes, 0, out, 1.5f, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: rounds = blocks / (256 * waves_per_simd); each round runs waves_per_simd waves x ITERS*8 instrs
  const double instr_per_simd = (double)blocks / 256 * ITERS * 8;  // one wave of each block lands on each SIMD
  printf("%-10s ILP %d  %d waves/SIMD: %.3f ms  %.2f cycles(@2.4GHz)/instr/SIMD\n", name, ILP, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
  (void)hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    const int lds = 160 * 1024 / w - 1024;  // LDS caps the CU at w blocks = w waves per SIMD
    run<1, 0>("f32", lds, w); run<2, 0>("f32", lds, w); run<4, 0>("f32", lds, w); run<8, 0>("f32", lds, w);
    run<1, 1>("f64", lds, w); run<2, 1>("f64", lds, w); run<4, 1>("f64", lds, w); run<8, 1>("f64", lds, w);
    run<4, 2>("mix1:3", lds, w); run<8, 2>("mix1:3", lds, w);
  }
  return 0;
}
