// Native benchmark of the direct-tracking path over the C ABI alone (no Python, no torch in the process):
// deterministic synthetic pairs made here, resident batch, uwt_track_batch_async + uwt_sync per step, one JSON line.
//
//   make -C tools            (g++ against include/uwt.h and uw-slam_amd/libuwt_hip.so; see tools/Makefile)
//   tools/uwt_bench [--pairs 1024] [--unique 32] [--steps 10] [--warmup 3] [--width 640] [--height 480] [--levels 4]
//                   [--iters 10] [--no-depth] [--reference-schedule]
//
// Synthetic inputs follow SURVEY.md §8(d): a band-limited texture (seeded noise, Gaussian blur sigma = 3 px, min-max to
// u8), the target re-rendered under a small random SE(3) for a fronto-parallel plane at depth z, TUM-like intrinsics.
// bench.py is the contract benchmark (its generator is numpy; the two are not bit-identical inputs); this program shows
// the library needs nothing but its header.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "uwt.h"

extern "C" {
// the three HIP runtime calls this file needs for the device pose buffer
int hipMalloc(void** p, size_t n);
int hipFree(void* p);
int hipMemcpy(void* dst, const void* src, size_t n, int kind);
}

namespace {

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

void blur_axis(std::vector<float>& img, int w, int h, bool horizontal, const std::vector<float>& k) {
  const int r = (int)k.size() / 2;
  std::vector<float> out(img.size());
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float s = 0.f;
      for (int i = -r; i <= r; i++) {
        int xx = horizontal ? x + i : x, yy = horizontal ? y : y + i;
        if (xx < 0) xx = -xx - 1;
        if (xx >= w) xx = 2 * w - 1 - xx;
        if (yy < 0) yy = -yy - 1;
        if (yy >= h) yy = 2 * h - 1 - yy;
        s += k[i + r] * img[(size_t)yy * w + xx];
      }
      out[(size_t)y * w + x] = s;
    }
  img.swap(out);
}

std::vector<uint8_t> texture(int w, int h, uint64_t seed) {
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
std::vector<uint8_t> render_target(const std::vector<uint8_t>& ref, int w, int h, double fx, double fy, double cx, double cy,
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

int arg_int(int argc, char** argv, const char* name, int def) {
  for (int i = 1; i + 1 < argc; i++)
    if (!std::strcmp(argv[i], name)) return std::atoi(argv[i + 1]);
  return def;
}
bool arg_flag(int argc, char** argv, const char* name) {
  for (int i = 1; i < argc; i++)
    if (!std::strcmp(argv[i], name)) return true;
  return false;
}

#define CHK(expr)                                                                      \
  do {                                                                                 \
    const int st_ = (expr);                                                            \
    if (st_ != UWT_OK) {                                                               \
      std::fprintf(stderr, "%s -> %d (%s)\n", #expr, st_, ctx ? uwt_last_error(ctx) : ""); \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

}  // namespace

int main(int argc, char** argv) {
  const int P = arg_int(argc, argv, "--pairs", 1024), U0 = arg_int(argc, argv, "--unique", 32);
  const int steps = arg_int(argc, argv, "--steps", 10), warmup = arg_int(argc, argv, "--warmup", 3);
  const int w = arg_int(argc, argv, "--width", 640), h = arg_int(argc, argv, "--height", 480);
  int levels = arg_int(argc, argv, "--levels", 4);
  const int iters = arg_int(argc, argv, "--iters", 10);
  const bool depth = !arg_flag(argc, argv, "--no-depth"), ref_sched = arg_flag(argc, argv, "--reference-schedule");
  const int U = U0 < P ? U0 : P;
  const float f = 525.0f * w / 640.0f, cx = w / 2 - 0.5f, cy = h / 2 - 0.5f;

  uwt_ctx* ctx = nullptr;
  uwt_params p;
  uwt_default_params(&p, w, h, f, f, cx, cy);
  if (ref_sched) {
    levels = 5;
    p.n_levels = 5; p.first_level = 4; p.last_level = 1; p.max_iters = 50; p.early_exit = 1;
  } else {
    p.n_levels = levels; p.first_level = levels - 1; p.last_level = 0; p.max_iters = iters; p.early_exit = 0;
  }
  p.has_depth = depth ? 1 : 0;
  p.max_frames = 2 * P;
  p.max_pairs = P;
  CHK(uwt_create(&p, &ctx));

  // U distinct pairs, tiled over the P resident pairs (slot 2i = reference, 2i+1 = target)
  std::vector<std::vector<uint8_t>> refs(U), tgts(U);
  std::vector<std::vector<uint16_t>> deps(U);
  for (int u = 0; u < U; u++) {
    Rng rng(1000 + u);
    const double z = 0.8 + 0.4 * ((u * 7) % 11) / 10.0;
    refs[u] = texture(w, h, u);
    tgts[u] = render_target(refs[u], w, h, f, f, cx, cy, z, rng);
    if (depth) {
      deps[u].assign((size_t)w * h, (uint16_t)std::lrint(z / 0.0002));
      for (auto& d : deps[u])
        if (rng.uniform() < 0.01) d = 0;
    }
  }
  for (int i = 0; i < P; i++) {
    const int u = i % U;
    CHK(uwt_set_frame(ctx, 2 * i, refs[u].data(), (size_t)w, depth ? deps[u].data() : nullptr, depth ? (size_t)w * 2 : 0));
    CHK(uwt_set_frame(ctx, 2 * i + 1, tgts[u].data(), (size_t)w, depth ? deps[u].data() : nullptr, depth ? (size_t)w * 2 : 0));
  }
  std::vector<int32_t> ref_slots(P), tgt_slots(P);
  for (int i = 0; i < P; i++) { ref_slots[i] = 2 * i; tgt_slots[i] = 2 * i + 1; }
  float* d_poses = nullptr;
  if (hipMalloc((void**)&d_poses, sizeof(float) * 7 * P) != 0) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }

  for (int i = 0; i < warmup; i++) CHK(uwt_track_batch_async(ctx, 0, 2 * P, 1, P, ref_slots.data(), tgt_slots.data(), d_poses, nullptr));
  CHK(uwt_sync(ctx));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < steps; i++) CHK(uwt_track_batch_async(ctx, 0, 2 * P, 1, P, ref_slots.data(), tgt_slots.data(), d_poses, nullptr));
  CHK(uwt_sync(ctx));
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  std::vector<float> poses((size_t)7 * P);
  if (hipMemcpy(poses.data(), d_poses, sizeof(float) * 7 * P, 2 /* hipMemcpyDeviceToHost */) != 0) return 1;
  double tmax = 0;
  int finite = 1;
  for (int i = 0; i < P; i++) {
    const float* q = &poses[7 * i];
    for (int k = 0; k < 7; k++) finite &= std::isfinite(q[k]) ? 1 : 0;
    tmax = std::fmax(tmax, std::sqrt((double)q[4] * q[4] + (double)q[5] * q[5] + (double)q[6] * q[6]));
  }
  // tiled pairs must give identical poses (same inputs, same arithmetic)
  int tiled_equal = 1;
  for (int i = U; i < P; i++) tiled_equal &= !std::memcmp(&poses[7 * i], &poses[7 * (i % U)], sizeof(float) * 7);
  std::printf("{\"metric\": \"frame-pair alignments/sec (%dx%d, %d pyr lvls)\", \"value\": %.2f, \"unit\": \"alignments/s\", "
              "\"n_gpus\": 1, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"host\": \"C++ over the C ABI\", "
              "\"config\": {\"workload\": \"%s, %d pairs resident (%d distinct)%s\"}, "
              "\"poses_finite\": %s, \"tiled_pairs_identical\": %s, \"max_translation_m\": %.6f}\n",
              w, h, levels, (double)P * steps / sec, steps, warmup, sec * 1e3 / steps,
              ref_sched ? "reference schedule (levels 4..1, <= 50 iterations, early exit)" : "fixed iterations, no early exit", P, U,
              depth ? ", u16 depth plane" : "", finite ? "true" : "false", tiled_equal ? "true" : "false", tmax);
  (void)hipFree(d_poses);
  uwt_destroy(ctx);
  return (finite && tiled_equal) ? 0 : 1;
}
