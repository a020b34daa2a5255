// What an EXEC-masked region costs around a run of f64 FMAs (masked_sums_lo / _hi of k_residual: s_and_saveexec_b64,
// 14 v_fmac_f64, s_mov_b64 exec), at the kernel's occupancy (4 waves / SIMD: 160 KB LDS / 4 blocks) and at 8.
//   hipcc -O3 --offload-arch=gfx950 exec_toggle.hip -o exec_toggle
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 4096;
#define FMA14 \
  "v_fmac_f64 %0, %14, %15\n v_fmac_f64 %1, %14, %16\n v_fmac_f64 %2, %14, %17\n v_fmac_f64 %3, %15, %15\n v_fmac_f64 %4, %15, %16\n" \
  "v_fmac_f64 %5, %15, %17\n v_fmac_f64 %6, %16, %16\n v_fmac_f64 %7, %16, %17\n v_fmac_f64 %8, %17, %17\n v_fmac_f64 %9, %14, %14\n" \
  "v_fmac_f64 %10, %14, %17\n v_fmac_f64 %11, %15, %14\n v_fmac_f64 %12, %16, %14\n v_fmac_f64 %13, %17, %14\n"
#define OPS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
            "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]) : "v"(j0), "v"(j1), "v"(j2), "v"(j3)
template <int MODE>   // 0: plain  1: saveexec / restore around each run of 14  2: around each run of 28
__global__ __launch_bounds__(256) void k(double* out, double seed, unsigned long long mask, unsigned long long* clk) {
  extern __shared__ float lds[];
  double a[14];
  for (int i = 0; i < 14; i++) a[i] = seed + i;
  double j0 = seed * 1e-9 + threadIdx.x * 1e-12, j1 = j0 * 1.1, j2 = j0 * 1.2, j3 = j0 * 1.3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
    if (MODE == 0) {
      asm volatile(FMA14 : OPS);
      asm volatile(FMA14 : OPS);
    } else if (MODE == 1) {
      asm volatile("s_and_saveexec_b64 s[20:21], %18\n" FMA14 "s_mov_b64 exec, s[20:21]" : OPS, "s"(mask) : "scc", "s20", "s21");
      asm volatile("s_and_saveexec_b64 s[20:21], %18\n" FMA14 "s_mov_b64 exec, s[20:21]" : OPS, "s"(mask) : "scc", "s20", "s21");
    } else {
      asm volatile("s_and_saveexec_b64 s[20:21], %18\n" FMA14 FMA14 "s_mov_b64 exec, s[20:21]" : OPS, "s"(mask) : "scc", "s20", "s21");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 14; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  if (seed == 12345.0) lds[threadIdx.x] = (float)s;
}
template <int MODE> void run(const char* name, int lds_bytes, int waves) {
  double* out; unsigned long long* clk;
  (void)hipMalloc(&out, 256 * 4096 * 8); (void)hipMalloc(&clk, 2 * 4096 * 8);
  const int blocks = 256 * waves;   // `waves` blocks of 4 waves per CU = waves per SIMD
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  for (int w = 0; w < 40; w++) k<MODE><<<blocks, 256, lds_bytes>>>(out, 1.0, 0xffffffff0fffffffull, clk);   // (the clock settles)
  (void)hipDeviceSynchronize();
  static unsigned long long h[2 * 4096];
  (void)hipMemcpy(h, clk, blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < blocks; i++) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  cyc /= blocks; rt /= blocks;
  // a wave's 100 MHz ticks per iteration, over the waves sharing its SIMD = SIMD time per 28 FMAs issued; clock = cycles / time
  printf("%-34s %d waves/SIMD: %.2f ns of SIMD time per run of 28 v_fmac_f64 (%.2f per FMA), s_memtime / s_memrealtime = %.2f GHz\n", name, waves,
         rt * 10.0 / ITERS / waves, rt * 10.0 / ITERS / waves / 28.0, cyc / (rt * 10.0));
  (void)hipFree(out); (void)hipFree(clk);
}
int main() {
  for (int waves : {2, 4, 5, 6, 8}) {
    const int lds = 160 * 1024 / waves - 1024;
    run<0>("plain", lds, waves);
    run<1>("masked, runs of 14 (the kernel's)", lds, waves);
    run<2>("masked, runs of 28", lds, waves);
  }
  return 0;
}
