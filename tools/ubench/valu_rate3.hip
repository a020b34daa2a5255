// packed f32 mul / add issue cost vs scalar (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float2_ p[8]; float a[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x * 0.01f; p[i] = float2_{a[i], a[i] * 0.5f}; }
  const float2_ m = {1.0000001f, 0.9999999f};
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) p[i] = p[i] * m;          // v_pk_mul_f32
      if (MODE == 1) p[i] = p[i] + m;          // v_pk_add_f32
      if (MODE == 2) { a[i] = a[i] * m.x; }    // v_mul_f32
      if (MODE == 3) { p[i].x = p[i].x * m.x; p[i].y = p[i].y * m.y; }  // two scalar muls (compiler may pack)
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
  float* out; (void)hipMalloc(&out, 256 * 2048 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 2048;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms -> %.2f cycles(@2.4GHz) per chain step\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * 4 * ITERS * 8));
  (void)hipFree(out);
}
int main() { run<0>("v_pk_mul_f32"); run<1>("v_pk_add_f32"); run<2>("v_mul_f32"); run<3>("2 x v_mul_f32 (-O3)"); return 0; }
