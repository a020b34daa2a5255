// Can the f64 MFMA pipe take part of k_residual's lane-wise f64 accumulation off the VALU?
// v_mfma_f64_4x4x4_4b_f64 with A = a(lane), B = b(lane): the diagonal of every 4x4 block is sum over the four lanes
// {i, i+4, i+8, i+12} of a*b — a lane-wise product reduced over 4 lanes, 64 useful MACs per instruction.
// Measures: cycles per MFMA per SIMD alone, and together with independent f32 / f64 VALU FMAs (do the pipes overlap?).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 2048;
template <int NM, int NF32, int NF64>  // per loop step: NM MFMAs, NF32 f32 FMAs, NF64 f64 VALU FMAs (all independent chains)
__global__ __launch_bounds__(256) void k(double* out, float seed) {
  double acc[8], d[8];
  float a[16];
  for (int i = 0; i < 8; i++) { acc[i] = 0.0; d[i] = seed + i; }
  for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x * 0.01f;
  double x = seed + threadIdx.x * 1e-3, y = seed * 0.5 + threadIdx.x * 1e-4;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < NM; i++) acc[i % 8] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[i % 8], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NF32; i++) a[i % 16] = __builtin_fmaf(a[i % 16], 1.0000001f, 1e-7f);
#pragma unroll
    for (int i = 0; i < NF64; i++) d[i % 8] = __builtin_fma(d[i % 8], 1.0000001, 1e-7);
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += acc[i] + d[i];
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM, int NF32, int NF64> void run(const char* name) {
  double* out; (void)hipMalloc(&out, 256 * 4096 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4 * 4;  // 4 rounds of 4 blocks per CU (LDS-free; VGPR use is small, so up to 8 waves/SIMD resident)
  hipLaunchKernelGGL((k<NM, NF32, NF64>), dim3(blocks), dim3(256), 0, 0, out, 1.5f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NF32, NF64>), dim3(blocks), dim3(256), 0, 0, out, 1.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double steps_per_simd = (double)blocks / 256 * ITERS;  // one wave of each block per SIMD
  printf("%-34s %.3f ms  %.1f cycles(@2.4GHz) per loop step per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / steps_per_simd);
  (void)hipFree(out);
}
int main() {
  run<8, 0, 0>("8 mfma");
  run<0, 32, 0>("32 f32");
  run<0, 0, 8>("8 f64 valu");
  run<8, 32, 0>("8 mfma + 32 f32");
  run<8, 0, 8>("8 mfma + 8 f64 valu");
  run<4, 32, 8>("4 mfma + 32 f32 + 8 f64 valu");
  run<0, 32, 12>("32 f32 + 12 f64 valu");
  run<2, 32, 10>("2 mfma + 32 f32 + 10 f64 valu");
  return 0;
}
