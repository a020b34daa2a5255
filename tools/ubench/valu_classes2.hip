// Round-2 companion of valu_classes.hip: issue cost (cycles per wave64 instruction per SIMD, 8 waves/SIMD, independent
// operands) of the candidate replacements for k_residual's half-rate instructions — packed f32 with a scalar operand,
// three-operand min/med, f32 ops with literal / inline constants, 24-bit mad, and VALU beside LDS reads.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 1024;
#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                             \
  __global__ __launch_bounds__(256) void name(float* out, float seed, int si, unsigned long long* clk) { \
    __shared__ float4 lds4[256];                                                        \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                         \
    lds4[threadIdx.x] = make_float4(seed, seed + 1, seed + 2, seed + 3);                \
    __syncthreads();                                                                    \
    const unsigned lds_addr = (threadIdx.x & 63) * 16;                                  \
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, c = seed * 3, d = seed * 5; \
    float e = a + 1, f = b + 1, g = c + 1, h = d + 1;                                   \
    double da = a, db = b, dc = c, dd = d;                                              \
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0;                                       \
    int ia = (int)a, ib = (int)b, ic = threadIdx.x * 3, id = threadIdx.x * 7;           \
    double sseed = __builtin_bit_cast(double, ((unsigned long long)__float_as_uint(seed) << 32) | __float_as_uint(seed)); \
    for (int it = 0; it < ITERS; it++) { REP8(body) }                                   \
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime() - t0; \
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (float)(da + db + dc + dd) + ia + ib + ic + id + q0.x + q1.y + (float)sseed + lds_addr; \
  }
// reference points
KERNEL(k_mul_vgpr, asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mul_sgpr, asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));)
// packed f32: two pixels per instruction
KERNEL(k_pk_mul_vgpr, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(sseed));)
KERNEL(k_pk_mul_sgpr, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "s"(sseed));)
KERNEL(k_pk_fma_vgpr, asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(sseed));)
KERNEL(k_pk_fma_sgpr, asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "s"(sseed));)
KERNEL(k_pk_add_sgpr, asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "s"(sseed));)
// min / max / med forms
KERNEL(k_min_vgpr, asm volatile("v_min_f32 %0, %4, %0\n v_min_f32 %1, %4, %1\n v_min_f32 %2, %4, %2\n v_min_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_min_sgpr, asm volatile("v_min_f32 %0, %4, %0\n v_min_f32 %1, %4, %1\n v_min_f32 %2, %4, %2\n v_min_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));)
KERNEL(k_max_vgpr, asm volatile("v_max_f32 %0, %4, %0\n v_max_f32 %1, %4, %1\n v_max_f32 %2, %4, %2\n v_max_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_med3_vgpr, asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
KERNEL(k_min3_vgpr, asm volatile("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
// f32 with inline constant / literal / abs-neg modifiers
KERNEL(k_add_inline, asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_add_literal, asm volatile("v_add_f32 %0, 0x40400000, %0\n v_add_f32 %1, 0x40400000, %1\n v_add_f32 %2, 0x40400000, %2\n v_add_f32 %3, 0x40400000, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_fma_inline, asm volatile("v_fma_f32 %0, %4, %0, 1.0\n v_fma_f32 %1, %4, %1, 1.0\n v_fma_f32 %2, %4, %2, 1.0\n v_fma_f32 %3, %4, %3, 1.0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_fma_vop3_vgpr, asm volatile("v_fma_f32 %0, %4, %0, %5\n v_fma_f32 %1, %4, %1, %5\n v_fma_f32 %2, %4, %2, %5\n v_fma_f32 %3, %4, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
KERNEL(k_fma_vop3_neg, asm volatile("v_fma_f32 %0, -%4, %0, %5\n v_fma_f32 %1, -%4, %1, %5\n v_fma_f32 %2, -%4, %2, %5\n v_fma_f32 %3, -%4, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
KERNEL(k_mul_e64_neg, asm volatile("v_mul_f32_e64 %0, %4, -%0\n v_mul_f32_e64 %1, %4, -%1\n v_mul_f32_e64 %2, %4, -%2\n v_mul_f32_e64 %3, %4, -%3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_sub_vgpr, asm volatile("v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %4, %1\n v_sub_f32 %2, %4, %2\n v_sub_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
// integer forms
KERNEL(k_mad_i24, asm volatile("v_mad_i32_i24 %0, %1, %1, %0\n v_mad_i32_i24 %1, %2, %2, %1\n v_mad_i32_i24 %2, %3, %3, %2\n v_mad_i32_i24 %3, %0, %0, %3" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));)
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %1, %2, %0\n v_add3_u32 %1, %2, %3, %1\n v_add3_u32 %2, %3, %0, %2\n v_add3_u32 %3, %0, %1, %3" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));)
KERNEL(k_cvt_ubyte, asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(ia));)
KERNEL(k_cmp_u32_sgpr, asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %2, %3\n v_cmp_lt_u32_e64 s[20:21], %1, %2\n v_cmp_lt_u32_e64 s[22:23], %3, %0" : : "v"(ia), "v"(ib), "v"(ic), "v"(id) : "s20", "s21", "s22", "s23");)
KERNEL(k_cmp_class, asm volatile("v_cmp_class_f32_e64 s[20:21], %0, %4\n v_cmp_class_f32_e64 s[22:23], %1, %4\n v_cmp_class_f32_e64 s[20:21], %2, %4\n v_cmp_class_f32_e64 s[22:23], %3, %4" : : "v"(a), "v"(b), "v"(c), "v"(d), "v"(ia) : "s20", "s21", "s22", "s23");)
// f64 forms
KERNEL(k_fma64, asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"((double)e), "v"((double)f));)
KERNEL(k_mul64, asm volatile("v_mul_f64 %0, %4, %5\n v_mul_f64 %1, %4, %5\n v_mul_f64 %2, %4, %5\n v_mul_f64 %3, %4, %5" : "=v"(da), "=v"(db), "=v"(dc), "=v"(dd) : "v"((double)e), "v"((double)f));)
KERNEL(k_add64, asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"((double)e));)
KERNEL(k_cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "=v"(da), "=v"(db), "=v"(dc), "=v"(dd) : "v"(a), "v"(b), "v"(c), "v"(d));)
// VALU beside LDS reads: 4 fmac + 1 ds_read_b128 (does the read take a vector issue slot?) and the read alone
KERNEL(k_fma_plus_ds, asm volatile("ds_read_b128 %4, %6\n v_fmac_f32 %0, %5, %0\n v_fmac_f32 %1, %5, %1\n v_fmac_f32 %2, %5, %2\n v_fmac_f32 %3, %5, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(q0) : "v"(e), "v"(lds_addr));)
KERNEL(k_fma_only4, asm volatile("v_fmac_f32 %0, %4, %0\n v_fmac_f32 %1, %4, %1\n v_fmac_f32 %2, %4, %2\n v_fmac_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_ds_only, asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2\n ds_read_b128 %0, %2\n ds_read_b128 %1, %2\n s_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1) : "v"(lds_addr));)
// mixes as in the kernel: 1 f64 fma + 2 f32 (all-vgpr) ; 1 cndmask + 2 f32
KERNEL(k_mix_f64_2f32, asm volatile("v_fmac_f64 %4, %6, %7\n v_fmac_f32 %0, %8, %0\n v_fmac_f32 %1, %8, %1\n v_fmac_f64 %5, %6, %7\n v_fmac_f32 %2, %8, %2\n v_fmac_f32 %3, %8, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(da), "+v"(db) : "v"(dc), "v"(dd), "v"(e));)
// the remaining instruction forms of k_residual's loop (as in valu_classes.hip), so that one table covers the kernel
KERNEL(k_cmp_sgpr, asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %2, %3\n v_cmp_lt_f32_e64 s[20:21], %1, %2\n v_cmp_lt_f32_e64 s[22:23], %3, %0" : : "v"(a), "v"(b), "v"(c), "v"(d) : "s20", "s21", "s22", "s23");)
KERNEL(k_cndmask, asm volatile("s_mov_b64 s[20:21], 0x5555\n v_cndmask_b32_e64 %0, 0, %0, s[20:21]\n v_cndmask_b32_e64 %1, 0, %1, s[20:21]\n v_cndmask_b32_e64 %2, 0, %2, s[20:21]\n v_cndmask_b32_e64 %3, 0, %3, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");)
KERNEL(k_cvt_sdwa, asm volatile("v_cvt_f32_i32_sdwa %0, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %1, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %2, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %3, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(ia), "v"(ib));)
KERNEL(k_sub_sdwa, asm volatile("v_sub_u32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_sub_u32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_u32_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_sub_u32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(ia), "=v"(ib), "=v"(ic), "=v"(id) : "v"(ia), "v"(ib));)
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "s"(si));)
KERNEL(k_mul_i24, asm volatile("v_mul_i32_i24 %0, %0, %1\n v_mul_i32_i24 %1, %1, %2\n v_mul_i32_i24 %2, %2, %3\n v_mul_i32_i24 %3, %3, %0" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));)
KERNEL(k_min_i32, asm volatile("v_min_i32 %0, %4, %0\n v_min_i32 %1, %4, %1\n v_min_i32 %2, %4, %2\n v_min_i32 %3, %4, %3" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "s"(si));)
KERNEL(k_cvt_rpi, asm volatile("v_cvt_rpi_i32_f32 %0, %4\n v_cvt_rpi_i32_f32 %1, %5\n v_cvt_rpi_i32_f32 %2, %6\n v_cvt_rpi_i32_f32 %3, %7" : "=v"(ia), "=v"(ib), "=v"(ic), "=v"(id) : "v"(a), "v"(b), "v"(c), "v"(d));)
KERNEL(k_cvt_f64_i32, asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "=v"(da), "=v"(db), "=v"(dc), "=v"(dd) : "v"(ia), "v"(ib), "v"(ic), "v"(id));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %5\n v_rcp_f32 %2, %6\n v_rcp_f32 %3, %7" : "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(a), "v"(b), "v"(c), "v"(d));)
KERNEL(k_mov64, asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4" : "=v"(da), "=v"(db), "=v"(dc), "=v"(dd) : "v"(sseed));)
template <typename K> void run(const char* name, K kern, int per_body) {
  printf("%-26s ", name); fflush(stdout);
  float* out; (void)hipMalloc(&out, 256 * 4096 * 4);
  unsigned long long* clk; (void)hipMalloc(&clk, 256 * 8 * 4 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 blocks per CU = 8 waves per SIMD resident
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 3, clk);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 3, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[256 * 8 * 4];
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += (double)v; mean /= (256 * 8 * 4);
  const double instr_per_simd = (double)blocks / 256 * ITERS * 8 * per_body;
  // s_memtime ticks at the shader clock: a wave's own start-to-end delta while 8 waves share its SIMD, over the
  // instructions the 8 of them issue = true issue cycles per instruction, whatever clock the chip holds
  printf("%.3f ms  %.2f cycles(@2.4GHz nominal)  %.2f shader cycles/instr/SIMD (s_memtime)  [clock ~%.2f GHz]\n", ms,
         ms * 1e-3 * 2.4e9 / instr_per_simd, mean / (8.0 * ITERS * 8 * per_body), mean / (ms * 1e-3) / 1e9); fflush(stdout);
  (void)hipFree(out); (void)hipFree(clk);
}
int main() {
  run("v_mul_f32 vgpr", k_mul_vgpr, 4); run("v_mul_f32 sgpr", k_mul_sgpr, 4);
  run("v_pk_mul_f32 vgpr", k_pk_mul_vgpr, 4); run("v_pk_mul_f32 sgpr", k_pk_mul_sgpr, 4);
  run("v_pk_fma_f32 vgpr", k_pk_fma_vgpr, 4); run("v_pk_fma_f32 sgpr", k_pk_fma_sgpr, 4); run("v_pk_add_f32 sgpr", k_pk_add_sgpr, 4);
  run("v_min_f32 vgpr", k_min_vgpr, 4); run("v_min_f32 sgpr", k_min_sgpr, 4); run("v_max_f32 vgpr", k_max_vgpr, 4);
  run("v_med3_f32 vgpr", k_med3_vgpr, 4); run("v_min3_f32 vgpr", k_min3_vgpr, 4);
  run("v_add_f32 inline 1.0", k_add_inline, 4); run("v_add_f32 literal", k_add_literal, 4); run("v_fma_f32 inline 1.0", k_fma_inline, 4);
  run("v_fma_f32 vop3 vgpr", k_fma_vop3_vgpr, 4); run("v_fma_f32 vop3 neg", k_fma_vop3_neg, 4); run("v_mul_f32_e64 neg", k_mul_e64_neg, 4);
  run("v_sub_f32 vgpr", k_sub_vgpr, 4);
  run("v_mad_i32_i24", k_mad_i24, 4); run("v_add3_u32", k_add3, 4); run("v_cvt_f32_ubyteN", k_cvt_ubyte, 4);
  run("v_cmp_lt_u32 -> sgpr", k_cmp_u32_sgpr, 4); run("v_cmp_class_f32 -> sgpr", k_cmp_class, 4);
  run("v_fmac_f64", k_fma64, 4); run("v_mul_f64", k_mul64, 4); run("v_add_f64", k_add64, 4); run("v_cvt_f64_f32", k_cvt_f64_f32, 4);
  run("4 fmac + ds_read_b128 (/5)", k_fma_plus_ds, 5); run("4 fmac alone", k_fma_only4, 4); run("ds_read_b128 x4", k_ds_only, 4);
  run("2 f64 + 4 f32 mix (/6)", k_mix_f64_2f32, 6);
  run("v_cmp_lt_f32_e64 -> sgpr", k_cmp_sgpr, 4); run("v_cndmask_b32_e64 (sgpr)", k_cndmask, 4); run("v_cvt_f32_i32_sdwa", k_cvt_sdwa, 4);
  run("v_sub_u32_sdwa", k_sub_sdwa, 4); run("v_mad_u32_u24", k_mad24, 4); run("v_mul_i32_i24", k_mul_i24, 4); run("v_min_i32", k_min_i32, 4);
  run("v_cvt_rpi_i32_f32", k_cvt_rpi, 4); run("v_cvt_f64_i32", k_cvt_f64_i32, 4); run("v_rcp_f32", k_rcp, 4); run("v_mov_b64", k_mov64, 4);
  return 0;
}
