// Store-side ceilings for the gradient stage (1 B read + 4 B written per pixel): what a kernel with Scharr's traffic and no
// arithmetic reaches, by store width and cache policy.   hipcc -O3 --offload-arch=gfx950 store_rate.hip -o store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: 8-byte stores, 4 px per thread (the kernel's shape)   1: 16-byte stores, 8 px per thread
// NT: nontemporal stores.   READ: also read the 1 B/px plane.
template <int MODE, bool NT, bool READ>
__global__ __launch_bounds__(256) void k_fill(const uint8_t* __restrict__ src, int16_t* __restrict__ gx, int16_t* __restrict__ gy, size_t n_px) {
  constexpr int PX = MODE ? 8 : 4;
  const size_t stride = (size_t)gridDim.x * 256 * PX;
  for (size_t p = ((size_t)blockIdx.x * 256 + threadIdx.x) * PX; p < n_px; p += stride) {
    uint32_t a = 0x00010002u, b = 0x00030004u;
    if constexpr (READ) {
      if constexpr (MODE) { const uint2 v = *reinterpret_cast<const uint2*>(src + p); a += v.x; b += v.y; }
      else { const uint32_t v = *reinterpret_cast<const uint32_t*>(src + p); a += v; b += v >> 3; }
    }
    if constexpr (MODE) {
      uint4 vx = make_uint4(a, b, a ^ b, a + b), vy = make_uint4(b, a, a + b, a ^ b);
      if constexpr (NT) {
        __builtin_nontemporal_store((u4v){vx.x, vx.y, vx.z, vx.w}, reinterpret_cast<u4v*>(gx + p));
        __builtin_nontemporal_store((u4v){vy.x, vy.y, vy.z, vy.w}, reinterpret_cast<u4v*>(gy + p));
      } else {
        *reinterpret_cast<uint4*>(gx + p) = vx;
        *reinterpret_cast<uint4*>(gy + p) = vy;
      }
    } else {
      uint2 vx = make_uint2(a, b), vy = make_uint2(b, a);
      if constexpr (NT) {
        __builtin_nontemporal_store((u2v){vx.x, vx.y}, reinterpret_cast<u2v*>(gx + p));
        __builtin_nontemporal_store((u2v){vy.x, vy.y}, reinterpret_cast<u2v*>(gy + p));
      } else {
        *reinterpret_cast<uint2*>(gx + p) = vx;
        *reinterpret_cast<uint2*>(gy + p) = vy;
      }
    }
  }
}

template <int MODE, bool NT, bool READ>
int run(const char* name, const uint8_t* src, int16_t* gx, int16_t* gy, size_t n_px, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) k_fill<MODE, NT, READ><<<blocks, 256>>>(src, gx, gy, n_px);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; i++) k_fill<MODE, NT, READ><<<blocks, 256>>>(src, gx, gy, n_px);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)n_px * (READ ? 5.0 : 4.0);
  printf("%-44s blocks %6d  %.3f ms  %.0f GB/s\n", name, blocks, ms / reps, bytes / (ms / reps) / 1e6);
  return 0;
}

int main() {
  const size_t n_px = (size_t)1024 * 640 * 480;   // the bench's 1024 reference frames at level 0
  uint8_t* src; int16_t *gx, *gy;
  CK(hipMalloc(&src, n_px)); CK(hipMalloc(&gx, n_px * 2)); CK(hipMalloc(&gy, n_px * 2));
  CK(hipMemset(src, 1, n_px));
  for (int blocks : {2048, 8192, 76800}) {
    run<0, false, false>("write only, 8 B stores", src, gx, gy, n_px, blocks);
    run<1, false, false>("write only, 16 B stores", src, gx, gy, n_px, blocks);
    run<0, true, false>("write only, 8 B nontemporal stores", src, gx, gy, n_px, blocks);
    run<1, true, false>("write only, 16 B nontemporal stores", src, gx, gy, n_px, blocks);
    run<0, false, true>("read 1 B + write 4 B, 8 B stores", src, gx, gy, n_px, blocks);
    run<1, false, true>("read 1 B + write 4 B, 16 B stores", src, gx, gy, n_px, blocks);
    run<0, true, true>("read 1 B + write 4 B, 8 B nt stores", src, gx, gy, n_px, blocks);
    run<1, true, true>("read 1 B + write 4 B, 16 B nt stores", src, gx, gy, n_px, blocks);
  }
  return 0;
}
