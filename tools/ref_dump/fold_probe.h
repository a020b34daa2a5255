// fold_probe.h — one point whose warped z tells HOW a build's cv::gemm folds the four partial sums of the 4-term rigid
// product rigid * points.t() (src/Tracker.cpp:1450; GEMMSingleMul<float,double>, "A * Bt" branch, len 4):
//     published source   "s0 += s1 + s2 + s3;"   ->  s0 + ((s1 + s2) + s3)      (the oracle's and the HIP kernels' fold 0)
//     left to right                               ->  ((s0 + s1) + s2) + s3      (uwo_set_gemm_fold(1))
// Both are sums of the same four exact products in double, rounded to float once; they differ in about one stored value in
// 10^9, so no dense dump can separate them.  This header CONSTRUCTS the separating point for a given row (r0, r1, r2) of the
// rigid matrix (any generic rotation: r2 in (0.5, 1], r0 and r1 non-zero), for a camera with fx = fy = 1, cx = cy = 0 (the
// unprojection is then the identity in either arithmetic set) and w = 0 (s3 = 0):
//   z    makes s2 = r2 * z fall within 2^18 double-ulps of a float midpoint m whose lower neighbour `lo` is even;
//   y    makes s1 = r1 * fl(y * z) cancel that distance up to +0.3 ulp:   s1 + s2 rounds to m exactly (in double);
//   x    makes s0 = r0 * fl(x * z) = +0.4 ulp:                            s0 + m   rounds to m  -> float: tie -> `lo`;
//   left to right, s0 + s1 is formed first and carries +0.7 ulp into s2:  the double lands on m + 1 ulp -> float `hi`.
// Plain C++ (no OpenCV, no Eigen): the reference-side driver (ref_dump.cpp) calls it with the reference's own matrix entries;
// tests/test_ref_vectors.py compiles it here and checks the construction against the oracle under both folds.
#pragma once
#include <cmath>

namespace uw_ref_dump {

struct FoldProbe {
  bool found;
  float x, y, z;     // the point is (x, y, z, w = 0)
  float lo, hi;      // warped z under "s0 += s1 + s2 + s3" (lo) and under the left-to-right fold (hi = next float up)
};

inline double fold_published(double s0, double s1, double s2, double s3) { double t = s1 + s2; t = t + s3; return s0 + t; }
inline double fold_left_to_right(double s0, double s1, double s2, double s3) { return ((s0 + s1) + s2) + s3; }

inline FoldProbe find_fold_probe(float r0, float r1, float r2) {
  FoldProbe fp = {false, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (!(r2 > 0.5f && r2 <= 1.0f) || r0 == 0.0f || r1 == 0.0f) return fp;
  for (long zi = 1; zi < (1L << 23); zi++) {
    const float z = 1.0f + (float)zi * 0x1p-23f;
    const double p = (double)r2 * (double)z;                 // exact: 24 x 24 bits
    int e;
    (void)std::frexp(p, &e);
    e -= 1;                                                  // p in [2^e, 2^(e+1))
    const double s32 = std::ldexp(1.0, e - 23), u = std::ldexp(1.0, e - 52);
    const double k = std::floor(p / s32);
    if (std::fmod(k, 2.0) != 0.0) continue;                  // the tie must round DOWN (to the even neighbour)
    const double m = (k + 0.5) * s32, delta = p - m;
    if (std::fabs(delta) > std::ldexp(u, 18)) continue;
    // y: s1 = r1 * fl(y * z) in [-delta + 0.25 u, -delta + 0.35 u]
    float y = (float)((-delta + 0.3 * u) / ((double)r1 * (double)z));
    bool ok = false;
    for (int step = 0; step < 256 && !ok; step++) {
      const float Y = y * z;
      const double eps = delta + (double)r1 * (double)Y;   // = s1 + s2 - m; (p - m) first: p + s1 would round to the grid of p
      if (eps >= 0.25 * u && eps <= 0.35 * u) { ok = true; break; }
      const bool up = (eps < 0.25 * u) == ((double)r1 * (double)z > 0.0);   // s1 grows with y when r1 * z > 0
      y = std::nextafter(y, up ? INFINITY : -INFINITY);
    }
    if (!ok) continue;
    const float x = (float)((0.4 * u) / ((double)r0 * (double)z));
    const float X = x * z, Y = y * z;
    const double s0 = (double)r0 * (double)X, s1 = (double)r1 * (double)Y;
    const float a = (float)fold_published(s0, s1, p, 0.0), b = (float)fold_left_to_right(s0, s1, p, 0.0);
    const float lo = (float)(k * s32), hi = (float)((k + 1.0) * s32);
    if (a == lo && b == hi) {
      fp.found = true; fp.x = x; fp.y = y; fp.z = z; fp.lo = lo; fp.hi = hi;
      return fp;
    }
  }
  return fp;
}

}  // namespace uw_ref_dump
