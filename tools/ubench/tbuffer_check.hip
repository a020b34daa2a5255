// tbuffer_check.hip — do typed buffer loads (tbuffer_load_format_*: the texture path converts i16 / u8 to f32) work on gfx950, and
// what do they return?  hipcc --offload-arch=gfx950 tools/ubench/tbuffer_check.hip -o tools/ubench/tbuffer_check && ./tbuffer_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long b = (unsigned long long)p;
  i4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));   // stride 0, no swizzle
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = __builtin_amdgcn_readfirstlane(0x00027FAC);                 // dst_sel xyzw, (format from the instruction)
  return r;
}
__global__ void k(const short* p, const unsigned char* q, float* out, int n) {
  const i4 rs = make_rsrc(p, n * 2), rq = make_rsrc(q, n);
  f4 v, w;
  float g;
  asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_SSCALED] offen" : "=v"(v) : "v"(threadIdx.x * 8u), "s"(rs));
  asm volatile("tbuffer_load_format_x %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8,BUF_NUM_FORMAT_USCALED] offen" : "=v"(g) : "v"(threadIdx.x * 3u + 1u), "s"(rq));
  asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen nt" : "=v"(w) : "v"(threadIdx.x * 4u), "s"(rq));
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v), "+v"(g), "+v"(w));
  float* o = out + threadIdx.x * 9;
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; o[4] = g; o[5] = w.x; o[6] = w.y; o[7] = w.z; o[8] = w.w;
}
int main() {
  const int n = 1024;
  std::vector<short> hs(n);
  std::vector<unsigned char> hb(n);
  for (int i = 0; i < n; i++) { hs[i] = (short)((i * 7919) % 24481 - 12240); hb[i] = (unsigned char)(i * 37 + 11); }
  hs[0] = -32768; hs[1] = 32767; hs[2] = -1; hs[3] = 0;
  short* ds; unsigned char* db; float* dout;
  hipMalloc(&ds, n * 2); hipMalloc(&db, n); hipMalloc(&dout, 64 * 9 * 4);
  hipMemcpy(ds, hs.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ds, db, dout, n);
  std::vector<float> ho(64 * 9);
  if (hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("FAILED to run\n"); return 1; }
  int bad = 0;
  for (int t = 0; t < 64; t++) {
    const float* o = ho.data() + t * 9;
    for (int c = 0; c < 4; c++) bad += o[c] != (float)hs[t * 4 + c];
    bad += o[4] != (float)hb[t * 3 + 1];
    for (int c = 0; c < 4; c++) bad += o[5 + c] != (float)hb[t * 4 + c];
  }
  std::printf("typed loads: %d mismatches of %d; lane 0: %g %g %g %g | %g | %g %g %g %g\n", bad, 64 * 9, ho[0], ho[1], ho[2], ho[3], ho[4], ho[5], ho[6], ho[7], ho[8]);
  return bad != 0;
}
