// How many 256-thread blocks with k_residual's resources a CU really holds: the runtime's answer (occupancy API) and a
// measurement (blocks that spin until a flag drops, counting how many are alive at once per CU via HW_REG_HW_ID / XCC_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_lds(unsigned* alive_max, unsigned* alive, int spin, float* out) {
  extern __shared__ unsigned char lds[];
  unsigned cu, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu_id = ((xcc & 0xf) << 8) | ((cu >> 8) & 0xf) | (((cu >> 13) & 0x7) << 4);   // xcc | se | cu
  const unsigned slot = cu_id & 2047u;
  if (threadIdx.x == 0) {
    const unsigned n = atomicAdd(&alive[slot], 1u) + 1u;
    atomicMax(&alive_max[slot], n);
  }
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  float acc = lds[(threadIdx.x * 7) & 255];
  for (int i = 0; i < spin; i++) acc = __builtin_fmaf(acc, 1.0000001f, 1e-7f);
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(&alive[slot], 1u);
}
int main() {
  for (int lds : {16384, 32768, 34928, 40960, 65536}) {
    int n = 0;
    (void)hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_lds, 256, lds);
    unsigned *amax, *alive; float* out;
    (void)hipMalloc(&amax, 2048 * 4); (void)hipMalloc(&alive, 2048 * 4); (void)hipMalloc(&out, 8192 * 256 * 4);
    (void)hipMemset(amax, 0, 2048 * 4); (void)hipMemset(alive, 0, 2048 * 4);
    k_lds<<<8192, 256, lds>>>(amax, alive, 200000, out);
    (void)hipDeviceSynchronize();
    static unsigned h[2048];
    (void)hipMemcpy(h, amax, 2048 * 4, hipMemcpyDeviceToHost);
    unsigned mx = 0, used = 0; double sum = 0;
    for (int i = 0; i < 2048; i++) if (h[i]) { used++; sum += h[i]; if (h[i] > mx) mx = h[i]; }
    printf("dynamic LDS %6d B: occupancy API says %d blocks per CU; measured: %u CU ids seen, max %u, mean %.2f blocks alive at once\n", lds, n, used, mx, sum / used);
  }
  return 0;
}
