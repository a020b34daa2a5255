// Does v_cvt_rpi_i32_f32 equal C round() for positive finite inputs (incl. 0.49999997 and x.5 ties)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float* in, int* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r;
  asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(in[i]));
  out[i] = r;
}
int main() {
  std::vector<float> v;
  for (int i = 0; i < 4200; i++) {                       // every tie and its neighbours up to 4200
    float t = i + 0.5f;
    v.push_back(t); v.push_back(nextafterf(t, 0.f)); v.push_back(nextafterf(t, 1e9f));
    v.push_back((float)i); v.push_back(nextafterf((float)i, 1e9f));
    if (i) v.push_back(nextafterf((float)i, 0.f));
  }
  unsigned s = 12345;
  for (int i = 0; i < 2000000; i++) { s = s * 1664525u + 1013904223u; v.push_back((s >> 8) * (4096.0f / 16777216.0f) + 1e-30f); }
  for (float e : {1e-30f, 1e-10f, 0.25f, 0.49999997f, 0.5f, 0.50000006f, 0.99999994f, 8388607.5f, 16777216.0f}) v.push_back(e);
  float* din; int* dout; int n = v.size();
  (void)hipMalloc(&din, n * 4); (void)hipMalloc(&dout, n * 4);
  (void)hipMemcpy(din, v.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
  std::vector<int> o(n);
  (void)hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++) if (o[i] != (int)roundf(v[i])) { if (bad < 10) printf("x=%.9g rpi=%d round=%d\n", v[i], o[i], (int)roundf(v[i])); bad++; }
  printf("checked %d values, mismatches %d\n", n, bad);
  return bad != 0;
}
