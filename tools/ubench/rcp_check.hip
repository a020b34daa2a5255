// Is one Newton correction on top of refined_rcp(d) already the correctly rounded 1/d?  Exhaustive over all 2^23
// mantissas at several exponents, against the IEEE divide (-fno-fast-math).  Prints mismatch counts for
//   r1 = refined_rcp(d)                         (hardware rcp + one Newton step)
//   r2 = fma(fma(-d, r1, 1), r1, r1)            (one more)
//   r3 = div_by(1, d, r1)                       (two more: what k_residual uses for inv_z2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float refined_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__global__ void k(int exponent, unsigned long long* bad) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= (1u << 23)) return;
  const float d = __uint_as_float(((uint32_t)(exponent + 127) << 23) | m);
  const float ref = 1.0f / d;
  const float r1 = refined_rcp(d);
  const float r2 = __builtin_fmaf(__builtin_fmaf(-d, r1, 1.0f), r1, r1);
  const float r3 = __builtin_fmaf(__builtin_fmaf(-d, r2, 1.0f), r1, r2);
  if (r1 != ref) atomicAdd(&bad[0], 1ull);
  if (r2 != ref) atomicAdd(&bad[1], 1ull);
  if (r3 != ref) atomicAdd(&bad[2], 1ull);
  const float dn = -d, refn = 1.0f / dn;
  const float n1 = refined_rcp(dn);
  const float n2 = __builtin_fmaf(__builtin_fmaf(-dn, n1, 1.0f), n1, n1);
  if (n2 != refn) atomicAdd(&bad[3], 1ull);
}
int main() {
  unsigned long long* bad; (void)hipMalloc(&bad, 32);
  for (int e : {-30, -7, -1, 0, 1, 2, 5, 13, 40}) {
    (void)hipMemset(bad, 0, 32);
    hipLaunchKernelGGL(k, dim3((1u << 23) / 256), dim3(256), 0, 0, e, bad);
    unsigned long long h[4]; (void)hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    printf("exponent %3d: mismatches vs IEEE 1/d over 2^23 mantissas: r1 %llu  r2 %llu  r3 %llu  r2(negative d) %llu\n", e, h[0], h[1], h[2], h[3]);
  }
  return 0;
}
