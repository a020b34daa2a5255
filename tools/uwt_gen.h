// uwt_gen.h — the deterministic synthetic input generator of the benchmarks: band-limited noise textures
// (standard-normal field, Gaussian blur sigma = 3 px, min-max to u8) re-rendered under a small random SE(3) for a
// fronto-parallel plane at depth z; TUM-style u16 depth at 0.0002 m per unit with 1 % invalid zeros.  Plain C++, no
// dependency: tools/uwt_bench.cpp uses it directly, bench.py through tools/libuwt_gen.so (uwt_gen_capi.cpp) — both
// benchmarks time the same inputs for the same pair ids.  Pair `gid`: texture seed gid, motion / hole seed 1000 + gid,
// plane depth z = 0.8 + 0.4 * ((7 gid) mod 11) / 10 (eleven planes over [0.8, 1.2] m).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace uwt_gen {

struct Rng {  // splitmix64
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double normal() {
    const double u1 = uniform() + 1e-300, u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

inline void blur_axis(std::vector<float>& img, int w, int h, bool horizontal, const std::vector<float>& k) {
  const int r = (int)k.size() / 2;
  std::vector<float> out(img.size());
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float s = 0.f;
      for (int i = -r; i <= r; i++) {
        int xx = horizontal ? x + i : x, yy = horizontal ? y : y + i;
        // symmetric reflection, folded until inside: one fold is all a side of 12 pixels or more needs (the kernel reaches 12), a
        // smaller image is folded again (found by tools/sanitize_cpu.sh: a 9 x 7 frame read outside its rows)
        while (xx < 0 || xx >= w) xx = xx < 0 ? -xx - 1 : 2 * w - 1 - xx;
        while (yy < 0 || yy >= h) yy = yy < 0 ? -yy - 1 : 2 * h - 1 - yy;
        s += k[i + r] * img[(size_t)yy * w + xx];
      }
      out[(size_t)y * w + x] = s;
    }
  img.swap(out);
}

inline std::vector<uint8_t> texture(int w, int h, uint64_t seed) {
  Rng rng(seed);
  std::vector<float> f((size_t)w * h);
  for (auto& v : f) v = (float)rng.normal();
  std::vector<float> k(25);
  float ks = 0.f;
  for (int i = -12; i <= 12; i++) ks += (k[i + 12] = std::exp(-0.5f * i * i / 9.0f));
  for (auto& v : k) v /= ks;
  blur_axis(f, w, h, true, k);
  blur_axis(f, w, h, false, k);
  float lo = f[0], hi = f[0];
  for (float v : f) { lo = std::fmin(lo, v); hi = std::fmax(hi, v); }
  std::vector<uint8_t> out(f.size());
  for (size_t i = 0; i < f.size(); i++) out[i] = (uint8_t)std::lrint((f[i] - lo) * (255.0f / (hi - lo)));
  return out;
}

// tgt(u') = ref(H^-1 u'), H = K (R + t n^T / z) K^-1, bilinear, reflected border
inline std::vector<uint8_t> render_target(const std::vector<uint8_t>& ref, int w, int h, double fx, double fy, double cx, double cy,
                                   double z, Rng& rng) {
  double ax[3] = {rng.normal(), rng.normal(), rng.normal()};
  double n = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-12;
  const double ang = rng.uniform() * 0.5 * 3.141592653589793 / 180.0;
  for (double& a : ax) a = a / n * ang;
  double td[3] = {rng.normal(), rng.normal(), rng.normal()};
  n = std::sqrt(td[0] * td[0] + td[1] * td[1] + td[2] * td[2]) + 1e-12;
  const double tl = rng.uniform() * 0.01;
  for (double& a : td) a = a / n * tl;
  const double th = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (th > 1e-12) {
    const double kx = ax[0] / th, ky = ax[1] / th, kz = ax[2] / th, s = std::sin(th), c = 1 - std::cos(th);
    const double Kx[9] = {0, -kz, ky, kz, 0, -kx, -ky, kx, 0};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double kk = 0;
        for (int m = 0; m < 3; m++) kk += Kx[3 * i + m] * Kx[3 * m + j];
        R[3 * i + j] += s * Kx[3 * i + j] + c * kk;
      }
  }
  double M[9];  // R + t n^T / z, n = (0, 0, 1)
  std::memcpy(M, R, sizeof(M));
  for (int i = 0; i < 3; i++) M[3 * i + 2] += td[i] / z;
  // H = K M K^-1
  const double K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1}, Ki[9] = {1 / fx, 0, -cx / fx, 0, 1 / fy, -cy / fy, 0, 0, 1};
  double T[9], H[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      T[3 * i + j] = 0;
      for (int m = 0; m < 3; m++) T[3 * i + j] += K[3 * i + m] * M[3 * m + j];
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      H[3 * i + j] = 0;
      for (int m = 0; m < 3; m++) H[3 * i + j] += T[3 * i + m] * Ki[3 * m + j];
    }
  const double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
  const double Hi[9] = {(H[4] * H[8] - H[5] * H[7]) / det, (H[2] * H[7] - H[1] * H[8]) / det, (H[1] * H[5] - H[2] * H[4]) / det,
                        (H[5] * H[6] - H[3] * H[8]) / det, (H[0] * H[8] - H[2] * H[6]) / det, (H[2] * H[3] - H[0] * H[5]) / det,
                        (H[3] * H[7] - H[4] * H[6]) / det, (H[1] * H[6] - H[0] * H[7]) / det, (H[0] * H[4] - H[1] * H[3]) / det};
  auto at = [&](int x, int y) {
    if (x < 0) x = -x - 1;
    if (x >= w) x = 2 * w - 1 - x;
    if (y < 0) y = -y - 1;
    if (y >= h) y = 2 * h - 1 - y;
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    return (double)ref[(size_t)y * w + x];
  };
  std::vector<uint8_t> out((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const double den = Hi[6] * x + Hi[7] * y + Hi[8];
      const double u = (Hi[0] * x + Hi[1] * y + Hi[2]) / den, v = (Hi[3] * x + Hi[4] * y + Hi[5]) / den;
      const int x0 = (int)std::floor(u), y0 = (int)std::floor(v);
      const double a = u - x0, b = v - y0;
      const double val = (1 - b) * ((1 - a) * at(x0, y0) + a * at(x0 + 1, y0)) + b * ((1 - a) * at(x0, y0 + 1) + a * at(x0 + 1, y0 + 1));
      const long q = std::lrint(val);
      out[(size_t)y * w + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
  return out;
}


inline double plane_depth(int gid) { return 0.8 + 0.4 * ((gid * 7) % 11) / 10.0; }

inline void gen_pair(int w, int h, double fx, double fy, double cx, double cy, int gid, bool with_depth, std::vector<uint8_t>& ref,
                     std::vector<uint8_t>& tgt, std::vector<uint16_t>& dep) {
  Rng rng(1000 + (uint64_t)gid);
  const double z = plane_depth(gid);
  ref = texture(w, h, (uint64_t)gid);
  tgt = render_target(ref, w, h, fx, fy, cx, cy, z, rng);
  dep.clear();
  if (with_depth) {
    dep.assign((size_t)w * h, (uint16_t)std::lrint(z / 0.0002));
    for (auto& d : dep)
      if (rng.uniform() < 0.01) d = 0;
  }
}

}  // namespace uwt_gen
