// Issue cost (cycles per wave64 instruction per SIMD, many waves, independent operands) of the instruction forms
// k_residual's hot loop is made of — the ones valu_rate*.hip did not cover.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 1024;
#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                             \
  __global__ __launch_bounds__(256) void name(float* out, float seed, int si) {        \
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, c = seed * 3, d = seed * 5; \
    float e = a + 1, f = b + 1, g = c + 1, h = d + 1;                                   \
    double da = a, db = b, dc = c, dd = d;                                              \
    int ia = (int)a, ib = (int)b, ic = threadIdx.x * 3, id = threadIdx.x * 7;           \
    unsigned long long m0 = 0x5555555555555555ull ^ (unsigned long long)si, m1 = ~m0;   \
    for (int it = 0; it < ITERS; it++) { REP8(body) }                                   \
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (float)(da + db + dc + dd) + ia + ib + ic + id + (float)(m0 & 1) + (float)(m1 & 1); \
  }
KERNEL(k_fma32, asm volatile("v_fmac_f32 %0, %4, %0\n v_fmac_f32 %1, %4, %1\n v_fmac_f32 %2, %4, %2\n v_fmac_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mul_vgpr, asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_fmac_sgpr, asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed), "v"(e));)
KERNEL(k_cmp_vcc, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %3, %0" : : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");)
KERNEL(k_med3, asm volatile("v_med3_f32 %0, %0, 0, %4\n v_med3_f32 %1, %1, 0, %4\n v_med3_f32 %2, %2, 0, %4\n v_med3_f32 %3, %3, 0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_minf, asm volatile("v_min_f32 %0, %4, %0\n v_min_f32 %1, %4, %1\n v_min_f32 %2, %4, %2\n v_min_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_maxf0, asm volatile("v_max_f32 %0, 0, %0\n v_max_f32 %1, 0, %1\n v_max_f32 %2, 0, %2\n v_max_f32 %3, 0, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_and, asm volatile("v_and_b32 %0, %4, %0\n v_and_b32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_and_b32 %3, %4, %3" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia));)
KERNEL(k_addu, asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7" : "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(a), "v"(b), "v"(c), "v"(d));)
KERNEL(k_cnd_vcc, asm volatile("v_cndmask_b32 %0, 0, %0, vcc\n v_cndmask_b32 %1, 0, %1, vcc\n v_cndmask_b32 %2, 0, %2, vcc\n v_cndmask_b32 %3, 0, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");)
KERNEL(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(da), "+v"(db) : "v"(dc));)
KERNEL(k_mul_sgpr, asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));)
KERNEL(k_mul_e64neg, asm volatile("v_mul_f32_e64 %0, %4, -%0\n v_mul_f32_e64 %1, %4, -%1\n v_mul_f32_e64 %2, %4, -%2\n v_mul_f32_e64 %3, %4, -%3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_fma_e64, asm volatile("v_fma_f32 %0, -%4, %0, %5\n v_fma_f32 %1, -%4, %1, %5\n v_fma_f32 %2, -%4, %2, %5\n v_fma_f32 %3, -%4, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
KERNEL(k_cmp_sgpr, asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\n v_cmp_lt_f32_e64 %1, %4, %5\n v_cmp_lt_f32_e64 %0, %3, %4\n v_cmp_lt_f32_e64 %1, %5, %2" : "+s"(m0), "+s"(m1) : "v"(a), "v"(b), "v"(c), "v"(d));)
KERNEL(k_cmp_and, asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\n s_and_b64 %1, %1, %0\n v_cmp_lt_f32_e64 %0, %4, %5\n s_and_b64 %1, %1, %0\n v_cmp_lt_f32_e64 %0, %3, %4\n s_and_b64 %1, %1, %0\n v_cmp_lt_f32_e64 %0, %5, %2\n s_and_b64 %1, %1, %0" : "+s"(m0), "+s"(m1) : "v"(a), "v"(b), "v"(c), "v"(d));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32_e64 %0, 0, %0, %4\n v_cndmask_b32_e64 %1, 0, %1, %5\n v_cndmask_b32_e64 %2, 0, %2, %4\n v_cndmask_b32_e64 %3, 0, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m0), "s"(m1));)
KERNEL(k_cvt_sdwa, asm volatile("v_cvt_f32_i32_sdwa %0, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %1, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %2, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %3, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(ia), "v"(ib));)
KERNEL(k_cvt_f32_i32, asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(ia), "v"(ib), "v"(ic), "v"(id));)
KERNEL(k_sub_sdwa, asm volatile("v_sub_u32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_sub_u32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_u32_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_sub_u32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(ia), "=v"(ib), "=v"(ic), "=v"(id) : "v"(ia), "v"(ib));)
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "s"(si));)
KERNEL(k_mul_i24, asm volatile("v_mul_i32_i24 %0, %0, %1\n v_mul_i32_i24 %1, %1, %2\n v_mul_i32_i24 %2, %2, %3\n v_mul_i32_i24 %3, %3, %0" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));)
KERNEL(k_min_i32, asm volatile("v_min_i32 %0, %4, %0\n v_min_i32 %1, %4, %1\n v_min_i32 %2, %4, %2\n v_min_i32 %3, %4, %3" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "s"(si));)
KERNEL(k_cvt_rpi, asm volatile("v_cvt_rpi_i32_f32 %0, %4\n v_cvt_rpi_i32_f32 %1, %5\n v_cvt_rpi_i32_f32 %2, %6\n v_cvt_rpi_i32_f32 %3, %7" : "=v"(ia), "=v"(ib), "=v"(ic), "=v"(id) : "v"(a), "v"(b), "v"(c), "v"(d));)
KERNEL(k_cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "=v"(da), "=v"(db), "=v"(dc), "=v"(dd) : "v"(a), "v"(b), "v"(c), "v"(d));)
KERNEL(k_cvt_f64_i32, asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "=v"(da), "=v"(db), "=v"(dc), "=v"(dd) : "v"(ia), "v"(ib), "v"(ic), "v"(id));)
KERNEL(k_fma64, asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"((double)e), "v"((double)f));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %5\n v_rcp_f32 %2, %6\n v_rcp_f32 %3, %7" : "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(a), "v"(b), "v"(c), "v"(d));)
KERNEL(k_add_const, asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
template <typename K> void run(const char* name, K kern, int per_body) {
  printf("%-22s ", name); fflush(stdout);
  float* out; (void)hipMalloc(&out, 256 * 4096 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 blocks per CU = 8 waves per SIMD resident
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 3);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)blocks / 256 * ITERS * 8 * per_body;
  printf("%.3f ms  %.2f cycles(@2.4GHz)/instr/SIMD\n", ms, ms * 1e-3 * 2.4e9 / instr_per_simd); fflush(stdout);
  (void)hipFree(out);
}
int main() {
  run("v_fmac_f32", k_fma32, 4); run("v_mul_f32 (vgpr srcs)", k_mul_vgpr, 4); run("v_mul_f32 (sgpr src)", k_mul_sgpr, 4); run("v_med3_f32", k_med3, 4); run("v_min_f32", k_minf, 4); run("v_max_f32 const", k_maxf0, 4); run("v_and_b32", k_and, 4); run("v_add_u32", k_addu, 4); run("v_mov_b32", k_mov, 4); run("v_cndmask vcc", k_cnd_vcc, 4); run("v_pk_mul_f32", k_pk_mul, 4); run("v_fmac_f32 (sgpr src)", k_fmac_sgpr, 4); run("v_cmp_lt_f32 -> vcc", k_cmp_vcc, 4); run("v_mul_f32_e64 neg", k_mul_e64neg, 4);
  run("v_fma_f32 (vop3 neg)", k_fma_e64, 4); run("v_cmp_lt_f32_e64->sgpr", k_cmp_sgpr, 4);
  run("v_cndmask_e64 (sgpr)", k_cndmask, 4); run("v_cvt_f32_i32_sdwa", k_cvt_sdwa, 4); run("v_cvt_f32_i32", k_cvt_f32_i32, 4);
  run("v_sub_u32_sdwa", k_sub_sdwa, 4); run("v_mad_u32_u24", k_mad24, 4); run("v_mul_i32_i24", k_mul_i24, 4); run("v_min_i32", k_min_i32, 4);
  run("v_cvt_rpi_i32_f32", k_cvt_rpi, 4); run("v_cvt_f64_f32", k_cvt_f64_f32, 4); run("v_cvt_f64_i32", k_cvt_f64_i32, 4);
  run("v_fmac_f64", k_fma64, 4); run("v_rcp_f32", k_rcp, 4); run("v_add_f32 const", k_add_const, 4);
  return 0;
}
