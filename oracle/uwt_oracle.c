/*
 * uwt_oracle.c — CPU ORACLE (test infrastructure; see uwt_oracle.h for the contract and the
 * arithmetic sets: G1..G4 = OpenCV 3.x's published generic paths, the default; S1/S3/S4 = the legacy set).  Plain C restatement of the UW-SLAM direct-tracking path.
 * Reference citations are relative to /root/reference.  PARITY UNPINNED (no reference goldens exist).
 *
 * Compile with -ffp-contract=off: every FMA in here is an explicit fmaf().
 */
#include "uwt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* parameters                                                                                   */
/* ------------------------------------------------------------------------------------------ */

void uwo_default_params(uwo_params* p, int width, int height, float fx, float fy, float cx, float cy) {
  memset(p, 0, sizeof(*p));
  p->width = width;
  p->height = height;
  p->fx = fx; p->fy = fy; p->cx = cx; p->cy = cy;
  p->n_levels = 5;        /* Options.cpp:26 */
  p->first_level = 4;     /* Tracker.cpp:368 */
  p->last_level = 1;      /* Tracker.cpp:369 */
  p->max_iters = 50;      /* Tracker.cpp:366 */
  p->epsilon = 0.001f;    /* Tracker.cpp:364 */
  p->gain = 50.0f;        /* Tracker.cpp:559 */
  p->z_factor = 1.0f;     /* Tracker.cpp:371 */
  p->angle_factor = 1.0f; /* Tracker.cpp:372 */
  p->depth_scale = 0.0002f; /* Tracker.cpp:1261 */
  p->initial_error = 50000.0f; /* Tracker.cpp:393 */
  p->early_exit = 1;
  p->has_depth = 0;
  p->handoff_scale_t = 0;
  p->weights = UWO_WEIGHTS_IDENTITY;
  p->sampler = UWO_SAMPLER_NEAREST;
  p->arith = UWO_ARITH_OPENCV;
}

/* Tracker::InitializePyramid, Tracker.cpp:297-340.  fx halves in double then narrows
 * (:317 "fx_[lvl-1] * 0.5"), cx_l = (cx0 + 0.5) / 2^l - 0.5 in double then narrows (:319). */
int uwo_level_intrinsics(const uwo_params* p, int lvl, uwo_level* out) {
  if (lvl < 0 || lvl >= p->n_levels || lvl >= UWO_MAX_LEVELS) return UWO_ERR_INVALID_ARG;
  float fx = p->fx, fy = p->fy;
  for (int l = 1; l <= lvl; l++) {
    fx = (float)((double)fx * 0.5);
    fy = (float)((double)fy * 0.5);
  }
  out->w = p->width >> lvl;   /* :312-313 "_width >> lvl": the POINT GRID of ObtainAllPoints (:1267-1268) */
  out->h = p->height >> lvl;
  /* the size of images_[lvl] itself: the cv::resize(.., Size(), 0.5, 0.5) chain of System.cpp:246-251 */
  out->iw = p->width;
  out->ih = p->height;
  for (int l = 1; l <= lvl; l++) {
    out->iw = uwo_half_size(out->iw);
    out->ih = uwo_half_size(out->ih);
  }
  out->fx = fx;
  out->fy = fy;
  if (lvl == 0) {
    out->cx = p->cx;
    out->cy = p->cy;
  } else {
    out->cx = (float)(((double)p->cx + 0.5) / (double)(1 << lvl) - 0.5);
    out->cy = (float)(((double)p->cy + 0.5) / (double)(1 << lvl) - 0.5);
  }
  out->invfx = 1.0f / fx; /* :328 */
  out->invfy = 1.0f / fy;
  return UWO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* pyramid + gradients                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* System.cpp:247 cv::resize(src, dst, Size(), 0.5, 0.5).  OpenCV 3.x imgproc/imgwarp.cpp (restated from memory of the published
 * source; to be confirmed by tools/ref_dump or by someone holding the 3.2 tree):
 *   dsize = Size(saturate_cast<int>(ssize.width * inv_scale_x), ...) — saturate_cast<int>(double) = cvRound = round half to even
 *   (733 -> 366, 735 -> 368);
 *   scale_x = 1 / inv_scale_x = 2 exactly, so is_area_fast holds whatever the parity of the source size, and INTER_LINEAR with
 *   iscale 2 x 2 is switched to INTER_AREA -> resizeAreaFast_Invoker<T, WT, ResizeAreaFastVec<..>>:
 *     dwidth1 = ssize.width / 2 (integer): the columns with a whole 2 x 2 cell;
 *     a row with sy0 + 2 <= ssize.height: columns dx < dwidth1 take the vector op's fast mode (scale 2 x 2, cn == 1)
 *       D[dx] = (S[2dx] + S[2dx+1] + nextS[2dx] + nextS[2dx+1] + 2) >> 2                      (u8 and u16 alike, in int);
 *     every other cell — the partial last column of such a row (2 dsize.width > ssize.width) and EVERY column of a partial last
 *     row (w = 0 there: the vector op is not called) — goes through the generic tail: the sum of the source pixels of the cell
 *     that exist, count of them, D[dx] = saturate_cast<T>((float)sum / count) — cvRound(float): round half to even
 *     (a + b = 5 -> 2, = 7 -> 4), where the fast mode rounds half up.
 * Even sizes: every cell is whole, the 2 x 2 mean (a+b+c+d+2)>>2 of rounds 1-5. */
int uwo_half_size(int n) {
  double v = (double)n * 0.5;
  return (int)lrint(v); /* cvRound(double): SSE2 cvtsd2si / lrint, round half to even */
}

#define UWO_RESIZE_HALF(NAME, T)                                                                  \
  void NAME(const T* src, int sw, int sh, T* dst) {                                              \
    const int dw = uwo_half_size(sw), dh = uwo_half_size(sh), dwidth1 = sw / 2;                  \
    for (int dy = 0; dy < dh; dy++) {                                                            \
      const int sy0 = 2 * dy;                                                                    \
      const int wfast = sy0 + 2 <= sh ? dwidth1 : 0;                                             \
      T* D = dst + (size_t)dy * dw;                                                              \
      int dx = 0;                                                                                \
      for (; dx < wfast; dx++) {                                                                 \
        const T* S = src + (size_t)sy0 * sw + 2 * dx;                                            \
        D[dx] = (T)(((int)S[0] + (int)S[1] + (int)S[sw] + (int)S[sw + 1] + 2) >> 2);             \
      }                                                                                          \
      for (; dx < dw; dx++) {                                                                    \
        const int sx0 = 2 * dx;                                                                  \
        float sum = 0.0f; /* WT = int (u8) / float (u16): the same value, sums stay below 2^24 */ \
        int count = 0;                                                                           \
        for (int sy = 0; sy < 2 && sy0 + sy < sh; sy++)                                          \
          for (int sx = 0; sx < 2 && sx0 + sx < sw; sx++) { sum += (float)src[(size_t)(sy0 + sy) * sw + sx0 + sx]; count++; } \
        D[dx] = count ? (T)lrintf(sum / (float)count) : (T)0;                                    \
      }                                                                                          \
    }                                                                                            \
  }
UWO_RESIZE_HALF(uwo_resize_half_u8, uint8_t)
UWO_RESIZE_HALF(uwo_resize_half_u16, uint16_t)

/* The whole-cell part alone, dst (w >> 1) x (h >> 1): what rounds 1-5 called the pyramid step (sizes divisible by 2). */
void uwo_halve_u8(const uint8_t* src, int w, int h, uint8_t* dst) {
  int w2 = w >> 1, h2 = h >> 1;
  for (int y = 0; y < h2; y++) {
    const uint8_t* r0 = src + (size_t)(2 * y) * w;
    const uint8_t* r1 = r0 + w;
    for (int x = 0; x < w2; x++)
      dst[(size_t)y * w2 + x] = (uint8_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2) >> 2);
  }
}

/* System.cpp:249 same call on the 16-bit depth image (invalid zeros are averaged in). */
void uwo_halve_u16(const uint16_t* src, int w, int h, uint16_t* dst) {
  int w2 = w >> 1, h2 = h >> 1;
  for (int y = 0; y < h2; y++) {
    const uint16_t* r0 = src + (size_t)(2 * y) * w;
    const uint16_t* r1 = r0 + w;
    for (int x = 0; x < w2; x++)
      dst[(size_t)y * w2 + x] =
          (uint16_t)(((uint32_t)r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2u) >> 2);
  }
}

static inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  return i;
}

/* Tracker.cpp:1133-1134.  The literal 3 binds to Scharr's `scale`; kernels stay integral
 * ([3 10 3]·3 smoothing x [-1 0 1] derivative), so the result is exact in int and never
 * saturates int16 (|g| <= 48·255). Correlation (not convolution): gx > 0 where I grows with x. */
void uwo_scharr3(const uint8_t* src, int w, int h, int16_t* gx, int16_t* gy) {
  for (int y = 0; y < h; y++) {
    int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    const uint8_t* rm = src + (size_t)ym * w;
    const uint8_t* r0 = src + (size_t)y * w;
    const uint8_t* rp = src + (size_t)yp * w;
    for (int x = 0; x < w; x++) {
      int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      int sx = 3 * (rm[xp] - rm[xm]) + 10 * (r0[xp] - r0[xm]) + 3 * (rp[xp] - rp[xm]);
      int sy = 3 * (rp[xm] - rm[xm]) + 10 * (rp[x] - rm[x]) + 3 * (rp[xp] - rm[xp]);
      sx *= 3;
      sy *= 3;
      if (sx > 32767) sx = 32767;
      if (sx < -32768) sx = -32768;
      if (sy > 32767) sy = 32767;
      if (sy < -32768) sy = -32768;
      gx[(size_t)y * w + x] = (int16_t)sx;
      gy[(size_t)y * w + x] = (int16_t)sy;
    }
  }
}

/* Tracker.cpp:1139-1142: convertScaleAbs (saturate |g| to u8) then addWeighted(0.5, 0.5)
 * (cvRound = round-half-to-even). */
void uwo_gradient_mag(const int16_t* gx, const int16_t* gy, int n, uint8_t* out) {
  for (int i = 0; i < n; i++) {
    int ax = abs((int)gx[i]), ay = abs((int)gy[i]);
    if (ax > 255) ax = 255;
    if (ay > 255) ay = 255;
    double v = 0.5 * ax + 0.5 * ay;
    long r = lrint(v); /* default rounding mode: to nearest even */
    out[i] = (uint8_t)(r > 255 ? 255 : r);
  }
}

/* Tracker::ObtainAllPoints, Tracker.cpp:1259-1310.  Depth is read through at<short> (signed). */
void uwo_dense_points(const uint16_t* depth, int w, int h, int lvl, float depth_scale, float* pts) {
  uwo_dense_points_ex(depth, w, w, h, lvl, depth_scale, pts);
}

/* the same loop over the w x h grid (w_[lvl] x h_[lvl], :1267-1268) of a depth image whose rows are `stride` elements long
 * (depths_[lvl].cols — the resize chain's size, >= w) */
void uwo_dense_points_ex(const uint16_t* depth, int stride, int w, int h, int lvl, float depth_scale, float* pts) {
  float factor_lvl = (float)((double)depth_scale / pow(2.0, (double)lvl)); /* :1266 */
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      float* p = pts + 4 * ((size_t)y * w + x);
      if (depth) {
        int16_t d = (int16_t)depth[(size_t)y * stride + x];
        if (d > 0) {
          p[0] = (float)x; p[1] = (float)y; p[2] = (float)d * factor_lvl; p[3] = 1.0f; /* :1276-1278 */
        } else {
          p[0] = 0.0f; p[1] = 0.0f; p[2] = 1.0f; p[3] = 0.0f; /* :1290-1291 */
        }
      } else {
        p[0] = (float)x; p[1] = (float)y; p[2] = 1.0f; p[3] = 1.0f; /* :1299-1302 */
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* SE(3) (Sophus SE3f / SO3f, Eigen quaternion kernels; S5, S6)                                */
/* ------------------------------------------------------------------------------------------ */

/* The arithmetic set, the gemm fold and the sine / cosine travel WITH THE CALL (uwo_ar): uwo_estimate_pose* take them from
 * uwo_params and hand them down, so that a thread pool may run alignments of different sets side by side.  The per-stage entry
 * points that have no parameter block (uwo_warp, uwo_residual_jacobian*, uwo_error, uwo_normal_equations, uwo_solve_delta,
 * uwo_se3_exp) read the CALLING THREAD's defaults, which uwo_set_arith / uwo_set_gemm_fold / uwo_set_trig change for that thread
 * alone (thread-local: nothing process-wide is left). */
typedef struct uwo_ar { int arith, fold, trig; } uwo_ar;
static _Thread_local uwo_ar t_ar = {UWO_ARITH_OPENCV, 0, UWO_TRIG_ROUNDED};

/* S5.  Sophus calls std::sin / std::cos on float (so3.hpp:538-558, se3.hpp:738-740): libm's sinf / cosf, whose last bit is the
 * libm build's (glibc 2.35's sinf is one ulp off the correctly rounded value at 0.063 % of the floats of [0, 0.5], its cosf at
 * 0.0015 %: tools/trig/trig_sweep.c, exhaustive; the reference's Ubuntu 16.04 glibc 2.23 has a different sinf again).
 * UWO_TRIG_ROUNDED (default, and what the HIP library computes): (float)sin((double)x), the correctly rounded value in all but
 * double-rounding corner cases — a function of x alone, the same on every host.  UWO_TRIG_LIBM: THIS host's sinf / cosf, for
 * measuring how much of a pose rides on that bit (tests/test_oracle.py). */
static inline float sin32(uwo_ar ar, float x) { return ar.trig == UWO_TRIG_LIBM ? sinf(x) : (float)sin((double)x); }
static inline float cos32(uwo_ar ar, float x) { return ar.trig == UWO_TRIG_LIBM ? cosf(x) : (float)cos((double)x); }
int uwo_set_trig(int trig) {
  int prev = t_ar.trig;
  t_ar.trig = trig == UWO_TRIG_LIBM ? UWO_TRIG_LIBM : UWO_TRIG_ROUNDED;
  return prev;
}

void uwo_se3_identity(float pose[7]) {
  /* Tracker.cpp:385: SE3(SO3::exp(0), 0) — Taylor branch of expAndTheta gives (0,0,0,1). */
  pose[0] = 0.0f; pose[1] = 0.0f; pose[2] = 0.0f; pose[3] = 1.0f;
  pose[4] = 0.0f; pose[5] = 0.0f; pose[6] = 0.0f;
}

/* Eigen Quaternion::toRotationMatrix (generic). q = x y z w; R row-major 3x3. so3.hpp:286-288. */
static void quat_to_R(const float q[4], float R[9]) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  float twx = tx * w, twy = ty * w, twz = tz * w;
  float txx = tx * x, txy = ty * x, txz = tz * x;
  float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
}

/* Eigen generic quaternion product (Hamilton), coefficient order x y z w. */
static void quat_mul(const float a[4], const float b[4], float o[4]) {
  float ax = a[0], ay = a[1], az = a[2], aw = a[3];
  float bx = b[0], by = b[1], bz = b[2], bw = b[3];
  float w = aw * bw - ax * bx - ay * by - az * bz;
  float x = aw * bx + ax * bw + ay * bz - az * by;
  float y = aw * by + ay * bw + az * bx - ax * bz;
  float z = aw * bz + az * bw + ax * by - ay * bx;
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

/* Eigen QuaternionBase::_transformVector: uv = 2·(q.vec × v); v + w·uv + q.vec × uv.  so3.hpp:320-322. */
static void quat_rotate(const float q[4], const float v[3], float o[3]) {
  float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  float ux = qy * v[2] - qz * v[1];
  float uy = qz * v[0] - qx * v[2];
  float uz = qx * v[1] - qy * v[0];
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  float cx = qy * uz - qz * uy;
  float cy = qz * ux - qx * uz;
  float cz = qx * uy - qy * ux;
  o[0] = (v[0] + qw * ux) + cx;
  o[1] = (v[1] + qw * uy) + cy;
  o[2] = (v[2] + qw * uz) + cz;
}

/* SO3::expAndTheta, so3.hpp:534-566.  epsilon<float> = 1e-5 (common.hpp:154-158). */
static void so3_exp_theta(uwo_ar ar, const float om[3], float q[4], float* theta_out) {
  float theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  float theta = sqrtf(theta_sq);
  float half_theta = 0.5f * theta;
  float imag, real;
  if (theta < 1e-5f) {
    float theta_po4 = theta_sq * theta_sq;
    imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
    real = 1.0f - (float)(1.0 / 8.0) * theta_sq + (float)(1.0 / 384.0) * theta_po4;
  } else {
    float s = sin32(ar, half_theta);
    imag = s / theta;
    real = cos32(ar, half_theta);
  }
  q[0] = imag * om[0]; q[1] = imag * om[1]; q[2] = imag * om[2]; q[3] = real;
  *theta_out = theta;
}

/* SE3::exp, se3.hpp:723-744.  xi = [upsilon(3), omega(3)]. */
static void se3_exp_ar(uwo_ar ar, const float xi[6], float pose[7]) {
  const float* up = xi;
  const float* om = xi + 3;
  float q[4], theta;
  so3_exp_theta(ar, om, q, &theta);
  float Om[9] = {0.0f, -om[2], om[1], om[2], 0.0f, -om[0], -om[1], om[0], 0.0f}; /* so3.hpp:618-627 */
  float Om2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Om2[3 * i + j] = (Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j]) + Om[3 * i + 2] * Om[6 + j];
  float V[9];
  if (theta < 1e-5f) {
    quat_to_R(q, V); /* se3.hpp:734 "V = so3.matrix()" */
  } else {
    float theta_sq = theta * theta;
    float c1 = (1.0f - cos32(ar, theta)) / theta_sq;
    float c2 = (theta - sin32(ar, theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; i++) {
      float id = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
      V[i] = (id + c1 * Om[i]) + c2 * Om2[i];
    }
  }
  pose[0] = q[0]; pose[1] = q[1]; pose[2] = q[2]; pose[3] = q[3];
  for (int i = 0; i < 3; i++) pose[4 + i] = (V[3 * i] * up[0] + V[3 * i + 1] * up[1]) + V[3 * i + 2] * up[2];
}
void uwo_se3_exp(const float xi[6], float pose[7]) { se3_exp_ar(t_ar, xi, pose); }

/* SE3::operator*= (se3.hpp:317-321) + SO3::operator*= with first-order renormalisation (so3.hpp:338-354). */
void uwo_se3_mul(const float a[7], const float b[7], float out[7]) {
  float rt[3];
  quat_rotate(a, b + 4, rt);
  float t0 = a[4] + rt[0], t1 = a[5] + rt[1], t2 = a[6] + rt[2];
  float q[4];
  quat_mul(a, b, q);
  float sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sn != 1.0f) {
    float s = 2.0f / (1.0f + sn);
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
  }
  out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  out[4] = t0; out[5] = t1; out[6] = t2;
}

/* SE3::matrix(), se3.hpp:253-268. Row-major 4x4. */
void uwo_se3_matrix(const float pose[7], float T[16]) {
  float R[9];
  quat_to_R(pose, R);
  T[0] = R[0]; T[1] = R[1]; T[2] = R[2];  T[3] = pose[4];
  T[4] = R[3]; T[5] = R[4]; T[6] = R[5];  T[7] = pose[5];
  T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = pose[6];
  T[12] = 0.0f; T[13] = 0.0f; T[14] = 0.0f; T[15] = 1.0f;
}

/* Level hand-off, Tracker.cpp:580-590: q.xyz *= 2, SE3(q, t) re-normalises (se3.hpp:446-448 →
 * so3.hpp:431-439 → normalize() so3.hpp:270-276).  EstimatePoseFeatures also doubles t (:856). */
int uwo_se3_handoff(float pose[7], int scale_t) {
  float x = pose[0] * 2.0f, y = pose[1] * 2.0f, z = pose[2] * 2.0f, w = pose[3];
  float len = sqrtf(x * x + y * y + z * z + w * w);
  if (!(len >= 1e-5f)) return UWO_ERR_INVALID_ARG; /* SOPHUS_ENSURE would abort */
  pose[0] = x / len; pose[1] = y / len; pose[2] = z / len; pose[3] = w / len;
  if (scale_t) { pose[4] *= 2.0f; pose[5] *= 2.0f; pose[6] *= 2.0f; }
  return UWO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* warp + per-point terms                                                                       */
/* ------------------------------------------------------------------------------------------ */

int uwo_set_arith(int arith) {
  int prev = t_ar.arith;
  t_ar.arith = arith == UWO_ARITH_LEGACY ? UWO_ARITH_LEGACY : UWO_ARITH_OPENCV;
  return prev;
}

/* How GEMMSingleMul's "A * Bt" branch folds its four partial sums.  matmul.cpp writes "s0 += s1 + s2 + s3;" ahead of the
 * store "d_data[j] = T(s0*alpha)": C++ evaluates the right-hand side first, so the stored sum is s0 + ((s1 + s2) + s3) —
 * fold 0, the default.  Fold 1, ((s0 + s1) + s2) + s3, is what a left-to-right reading of "(s0+s1+s2+s3)" gives; kept
 * selectable so that a dump of a real build (tools/ref_dump) can say which of the two it follows.  The two differ only where
 * the double-precision sums round differently AND the float rounding of the result falls between them: about one stored
 * value in 10^9. */
int uwo_set_gemm_fold(int fold) {
  int prev = t_ar.fold;
  t_ar.fold = fold ? 1 : 0;
  return prev;
}
static inline double fold4(uwo_ar ar, double s0, double s1, double s2, double s3) {
  if (ar.fold) return ((s0 + s1) + s2) + s3;
  double t = s1 + s2;
  t = t + s3;
  return s0 + t;
}

/* "(col - c) * inv" of Tracker.cpp:1439 / :1443 on one element.
 * G3: operator-(Mat, Scalar) makes MatOp_AddEx(a, alpha = 1, s = -(double)c); operator*(MatExpr, double) is
 * MatOp_AddEx::multiply: alpha *= inv, s *= inv — both in double; Mat::operator=(MatExpr) runs MatOp_AddEx::assign:
 * |alpha| != 1  ->  a.convertTo(m, type, alpha, s)  ->  cvtScale32f: x * (float)alpha + (float)s;
 * alpha == 1 -> cv::add(a, s); alpha == -1 -> cv::subtract(s, a) (scalar narrowed to f32 by arithm_op).
 * Legacy: the expression as written. */
static inline float unproject_scaled(uwo_ar ar, float x, float c, float inv) {
  if (ar.arith == UWO_ARITH_LEGACY) return (x - c) * inv;
  double alpha = (double)inv;
  double sc = -(double)c * (double)inv;
  if (alpha == 1.0) return x + (float)sc;
  if (alpha == -1.0) return (float)sc - x;
  float m = x * (float)alpha;
  return m + (float)sc;
}

/* Tracker::WarpFunction, Tracker.cpp:1417-1471.  The 4x4·4xN product follows G1 (legacy: S1). */
static void warp_ar(uwo_ar ar, const float* pts, int n, const float pose[7], const uwo_level* L, float* warped) {
  float T[16];
  uwo_se3_matrix(pose, T);
  float fx = L->fx, fy = L->fy, cx = L->cx, cy = L->cy, invfx = L->invfx, invfy = L->invfy;
  for (int i = 0; i < n; i++) {
    const float* p = pts + 4 * (size_t)i;
    float z = p[2], w = p[3];
    float X = unproject_scaled(ar, p[0], cx, invfx); /* :1439 */
    X = X * z;                                   /* :1440 cv::multiply */
    float Y = unproject_scaled(ar, p[1], cy, invfy); /* :1443 */
    Y = Y * z;                                   /* :1444 */
    float o[4];
    for (int k = 0; k < 4; k++) {  /* :1450 rigid * P^T */
      if (ar.arith != UWO_ARITH_LEGACY) {
        /* G1: gemm(rigid, P, 1, GEMM_2_T) -> GEMMSingleMul<float,double>, "A * Bt" branch, n = 4: the unrolled loop
         * runs once, s0 = a0*b0, s1 = a1*b1, s2 = a2*b2, s3 = a3*b3 (exact in double), "s0 += s1 + s2 + s3" (fold4),
         * d = T(s0*alpha), alpha = 1 */
        double s0 = (double)T[4 * k] * (double)X, s1 = (double)T[4 * k + 1] * (double)Y;
        double s2 = (double)T[4 * k + 2] * (double)z, s3 = (double)T[4 * k + 3] * (double)w;
        o[k] = (float)fold4(ar, s0, s1, s2, s3);
        continue;
      }
      float s = T[4 * k] * X;
      s = fmaf(T[4 * k + 1], Y, s);
      s = fmaf(T[4 * k + 2], z, s);
      s = fmaf(T[4 * k + 3], w, s);
      o[k] = s;
    }
    float u = o[0] * fx; u = u / o[2]; u = u + cx; /* :1454-1456 */
    float v = o[1] * fy; v = v / o[2]; v = v + cy; /* :1459-1461 */
    u = u * o[3];                                   /* :1466 */
    v = v * o[3];                                   /* :1467 */
    float* q = warped + 4 * (size_t)i;
    q[0] = u; q[1] = v; q[2] = o[2]; q[3] = o[3];
  }
}

void uwo_warp(const float* pts, int n, const float pose[7], const uwo_level* L, float* warped) { warp_ar(t_ar, pts, n, pose, L, warped); }

/* EXTENSION (north star; the reference samples nearest-neighbour only): bilinear interpolation, f32, fixed order. */
float uwo_bilinear_u8(const uint8_t* img, int w, int h, float x, float y) {
  float x0 = floorf(x), y0 = floorf(y);
  float ax = x - x0, ay = y - y0;
  int ix0 = (int)x0, iy0 = (int)y0;
  if (ix0 > w - 1) ix0 = w - 1;
  if (iy0 > h - 1) iy0 = h - 1;
  int ix1 = ix0 + 1 > w - 1 ? w - 1 : ix0 + 1;
  int iy1 = iy0 + 1 > h - 1 ? h - 1 : iy0 + 1;
  float a = (float)img[(size_t)iy0 * w + ix0], b = (float)img[(size_t)iy0 * w + ix1];
  float c = (float)img[(size_t)iy1 * w + ix0], d = (float)img[(size_t)iy1 * w + ix1];
  float top = fmaf(ax, b - a, a);
  float bot = fmaf(ax, d - c, c);
  return fmaf(ay, bot - top, top);
}

static int residual_jacobian_ar(uwo_ar ar, const uint8_t* img1, const uint8_t* img2, const int16_t* gx1, const int16_t* gy1,
                                const float* pts, const float* warped, int n, const uwo_level* L,
                                float zf, float af, int sampler, float* J, float* r, int32_t* idx);

int uwo_residual_jacobian(const uint8_t* img1, const uint8_t* img2, const int16_t* gx1, const int16_t* gy1,
                          const float* pts, const float* warped, int n, const uwo_level* L,
                          float zf, float af, float* J, float* r, int32_t* idx) {
  return residual_jacobian_ar(t_ar, img1, img2, gx1, gy1, pts, warped, n, L, zf, af, UWO_SAMPLER_NEAREST, J, r, idx);
}

int uwo_residual_jacobian_ex(const uint8_t* img1, const uint8_t* img2, const int16_t* gx1, const int16_t* gy1,
                             const float* pts, const float* warped, int n, const uwo_level* L,
                             float zf, float af, int sampler, float* J, float* r, int32_t* idx) {
  return residual_jacobian_ar(t_ar, img1, img2, gx1, gy1, pts, warped, n, L, zf, af, sampler, J, r, idx);
}

/* Tracker.cpp:432-490. Returns the number of valid rows written.  The planes are the level's IMAGES (L->iw x L->ih, tight rows):
 * the bounds test reads "image2.rows / image2.cols" (:450) and Mat::at indexes the Mat — the size of the resize chain, which is
 * larger than the point grid (L->w x L->h = w_[lvl] x h_[lvl]) where a level-0 size is not divisible by 2^lvl. */
static int residual_jacobian_ar(uwo_ar ar, const uint8_t* img1, const uint8_t* img2, const int16_t* gx1, const int16_t* gy1,
                                const float* pts, const float* warped, int n, const uwo_level* L,
                                float zf, float af, int sampler, float* J, float* r, int32_t* idx) {
  int w = L->iw, h = L->ih;
  float fx = L->fx, fy = L->fy;
  int nv = 0;
  for (int i = 0; i < n; i++) {
    float x1 = pts[4 * (size_t)i], y1 = pts[4 * (size_t)i + 1];
    float x2 = warped[4 * (size_t)i], y2 = warped[4 * (size_t)i + 1], z2 = warped[4 * (size_t)i + 2];
    float iz = 1.0f / z2; /* :447 */
    if (y2 > 0.0f && y2 < (float)h && x2 > 0.0f && x2 < (float)w) { /* :450 */
      if (z2 != 0.0f) {                                              /* :451 */
        /* A table row whose reference position lies outside the level: Mat::at(y1, x1) (:474-477) would read outside the
         * image there — undefined in the reference (its own producers never emit such a row).  Defined here: the row is
         * dropped, like a row that fails :450-451 (the HIP path does the same; INTEGRATION.md "deviations"). */
        if ((int)x1 < 0 || (int)x1 >= w || (int)y1 < 0 || (int)y1 >= h) continue;
        if (iz < 0.0f) iz = 0.0f;                                    /* :452-453 */
        float Jw[2][6];
        Jw[0][0] = fx * iz;                               /* :455 */
        Jw[0][1] = 0.0f;
        Jw[0][2] = -(fx * x2 * iz * iz) * zf;             /* :457 */
        Jw[0][3] = -(fx * x2 * y2 * iz * iz) * af;        /* :458 */
        Jw[0][4] = (fx * (1.0f + x2 * x2 * iz * iz)) * af; /* :459 */
        Jw[0][5] = -fx * y2 * iz * af;                    /* :460 */
        Jw[1][0] = 0.0f;
        Jw[1][1] = fy * iz;                               /* :463 */
        Jw[1][2] = -(fy * y2 * iz * iz) * zf;             /* :464 */
        Jw[1][3] = -(fy * (1.0f + y2 * y2 * iz * iz)) * af; /* :465 */
        Jw[1][4] = fy * x2 * y2 * iz * iz * af;           /* :466 */
        Jw[1][5] = fy * x2 * iz * af;                     /* :467 */

        int ix1 = (int)x1, iy1 = (int)y1; /* Mat::at(float,float) truncates */
        int ix2 = (int)roundf(x2), iy2 = (int)roundf(y2); /* :472 */
        if (ix2 > w - 1) ix2 = w - 1; /* S7 */
        if (iy2 > h - 1) iy2 = h - 1;
        int i1 = img1[(size_t)iy1 * w + ix1];
        int i2 = img2[(size_t)iy2 * w + ix2];
        float res = (float)(i2 - i1); /* :474 */
        if (sampler == UWO_SAMPLER_BILINEAR) res = uwo_bilinear_u8(img2, w, h, x2, y2) - (float)i1; /* extension */
        float jl0 = (float)gx1[(size_t)iy1 * w + ix1]; /* :476 */
        float jl1 = (float)gy1[(size_t)iy1 * w + ix1]; /* :477 */
        float* Jr = J + 6 * (size_t)nv;
        for (int k = 0; k < 6; k++) { /* :479 Jl * Jw */
          if (ar.arith != UWO_ARITH_LEGACY) {
            /* G1: gemm(Jl 1x2, Jw 2x6): flags 0, len 2, d_size 6x1 -> no inline special case (needs len == width|height);
             * GEMMSingleMul<float,double>, "d_size.width * sizeof <= 1600" branch: WT s(0); for k: s += WT(a[k]) * WT(b[k][j]);
             * d[j] = T(s * alpha) */
            double sj = 0.0;
            sj += (double)jl0 * (double)Jw[0][k];
            sj += (double)jl1 * (double)Jw[1][k];
            Jr[k] = (float)sj;
            continue;
          }
          float s = jl0 * Jw[0][k];
          s = fmaf(jl1, Jw[1][k], s);
          Jr[k] = s;
        }
        r[nv] = res;
        if (idx) idx[nv] = i;
        nv++;
      }
    }
  }
  return nv;
}

/* ------------------------------------------------------------------------------------------ */
/* weights (Tracker.cpp:1571-1654)                                                              */
/* ------------------------------------------------------------------------------------------ */

/* Tracker::MedianMat, :1571-1594: saturating convert to u8 (cvRound; negatives -> 0), 256-bin histogram. */
float uwo_median_mat(const float* v, int n) {
  int hist[256];
  memset(hist, 0, sizeof(hist));
  for (int i = 0; i < n; i++) {
    long q = lrintf(v[i]);
    if (q < 0) q = 0;
    if (q > 255) q = 255;
    hist[q]++;
  }
  float m = (float)(n / 2);
  int bin = 0;
  float med = -1.0f;
  for (int i = 0; i < 256 && med < 0.0f; ++i) {
    bin += hist[i];
    if ((float)bin > m && med < 0.0f) med = (float)i;
  }
  return med;
}

/* Tracker::MedianAbsoluteDeviation, :1607-1619 */
float uwo_mad(const float* v, int n) {
  float c = 1.4826f;
  float median = uwo_median_mat(v, n);
  float* dev = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) dev[i] = fabsf(v[i] - median);
  float mad = uwo_median_mat(dev, n);
  free(dev);
  return c * mad;
}

/* Tracker::TukeyFunctionWeights, :1626-1654 */
void uwo_tukey_weights(const float* r, int n, float* w) {
  float b = 4.6851f;
  float MAD = uwo_mad(r, n);
  if (MAD == 0.0f) MAD = 1.0f;
  float inv_MAD = (float)(1.0 / (double)MAD);
  float inv_b2 = (float)(1.0 / (double)(b * b));
  for (int i = 0; i < n; i++) {
    float x = r[i] * inv_MAD;
    if (fabsf(x) <= b) {
      float tukey = (float)(1.0 - (double)((x * x) * inv_b2));
      w[i] = tukey * tukey;
    } else {
      w[i] = 0.0f;
    }
  }
}

/* EXTENSION: Huber weights (see header). */
static float signed_hist_median(const int* q, int n, int lo, int hi) { /* first bin whose cumulative count > n/2 */
  int nb = hi - lo + 1;
  int* hist = (int*)calloc((size_t)nb, sizeof(int));
  for (int i = 0; i < n; i++) {
    int v = q[i] < lo ? lo : (q[i] > hi ? hi : q[i]);
    hist[v - lo]++;
  }
  float m = (float)(n / 2);
  int bin = 0;
  float med = (float)hi;
  for (int i = 0; i < nb; i++) {
    bin += hist[i];
    if ((float)bin > m) { med = (float)(lo + i); break; }
  }
  free(hist);
  return med;
}

void uwo_huber_weights(const float* r, int n, float* w) {
  const float k = 1.345f;
  int* q = (int*)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
  for (int i = 0; i < n; i++) q[i] = (int)lrintf(r[i]);
  float med = signed_hist_median(q, n, -255, 255);
  for (int i = 0; i < n; i++) q[i] = abs(q[i] - (int)med);
  float MAD = 1.4826f * signed_hist_median(q, n, 0, 510);
  free(q);
  if (MAD == 0.0f) MAD = 1.0f;
  float inv_MAD = (float)(1.0 / (double)MAD);
  for (int i = 0; i < n; i++) {
    float ax = fabsf(r[i] * inv_MAD);
    w[i] = ax <= k ? 1.0f : k / ax;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* error, normal equations, solve                                                               */
/* ------------------------------------------------------------------------------------------ */

/* A cv::gemm on CV_32F whose result is one column wide, len = n: row `a` (n floats, stride sa) times the vector b.
 * matmul.cpp gemmImpl: "(d_size.width == 1 || len == 1) && !(flags & GEMM_2_T) && B.isContinuous()" sets b_step = 0 and
 * GEMM_2_T, so both code paths below run their "A * Bt" branch.  `rows` = d_size.height (6 for Jᵀr, 1 for rᵀr).
 *  - single pass ("(d_size.height <= 64 || d_size.width <= 64) && len <= 10000"): GEMMSingleMul<float,double>, four partial
 *    sums over k, k+1, k+2, k+3 (CV_ENABLE_UNROLLED is 1 outside ICC / CV_DISABLE_OPTIMIZATION), the tail into s0, then
 *    "s0 += s1 + s2 + s3" (fold4), s0 * alpha;
 *  - otherwise the block algorithm: dm0 = min(128, rows), dn0 = 1, dk0 = min(16384 / dm0, 16384 / dn0, len); per block
 *    [k, k + dk): GEMMBlockMul<float,double>: s0 = (first block ? 0 : d_buf), s1 = 0, pairs into s0 / s1, an odd last term
 *    into s0, d_buf = s0 + s1; a block absorbs the remainder when "k + dk >= len || 8*(k + dk) + dk > 8*len";
 *    GEMMStore: alpha * d_buf.
 * Returns the double before the final (float) store.  Legacy set: one sequential sum. */
static double gemm_dot_width1(uwo_ar ar, const float* a, size_t sa, const float* b, int n, int rows, double alpha) {
  if (ar.arith == UWO_ARITH_LEGACY) {
    double s = 0.0;
    for (int k = 0; k < n; k++) s += (double)a[sa * (size_t)k] * (double)b[k];
    return s * alpha;
  }
  if (n <= 10000) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k <= n - 4; k += 4) {
      s0 += (double)a[sa * (size_t)k] * (double)b[k];
      s1 += (double)a[sa * (size_t)(k + 1)] * (double)b[k + 1];
      s2 += (double)a[sa * (size_t)(k + 2)] * (double)b[k + 2];
      s3 += (double)a[sa * (size_t)(k + 3)] * (double)b[k + 3];
    }
    for (; k < n; k++) s0 += (double)a[sa * (size_t)k] * (double)b[k];
    return fold4(ar, s0, s1, s2, s3) * alpha;
  }
  const int block_size = 128 * 128;
  int dm0 = rows < 128 ? rows : 128;
  int dk0 = block_size / dm0;
  if (dk0 > block_size) dk0 = block_size; /* block_size / dn0, dn0 = 1 */
  if (dk0 > n) dk0 = n;
  double d = 0.0;
  int dk;
  for (int k = 0; k < n; k += dk) {
    dk = dk0;
    if (k + dk >= n || 8 * (long long)(k + dk) + dk > 8 * (long long)n) dk = n - k;
    double s0 = d, s1 = 0.0; /* do_acc: the first block starts from 0 */
    int q = 0;
    for (; q <= dk - 2; q += 2) {
      s0 += (double)a[sa * (size_t)(k + q)] * (double)b[k + q];
      s1 += (double)a[sa * (size_t)(k + q + 1)] * (double)b[k + q + 1];
    }
    for (; q < dk; q++) s0 += (double)a[sa * (size_t)(k + q)] * (double)b[k + q];
    d = s0 + s1;
  }
  return alpha * d;
}

/* Tracker.cpp:499-502 (S2, S8): errorMat = inv_num_residuals * Residuals.t() * ResidualsW — MatOp_T::multiply folds the
 * scalar into the transpose's alpha, MatOp::matmul makes one gemm(R, RW, alpha = inv_n, GEMM_1_T) with a 1x1 result.
 * w may be NULL (identity). */
static float error_ar(uwo_ar ar, const float* r, const float* w, int n, int64_t* sum_r2_out) {
  int64_t si = 0;
  float* rw = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    rw[i] = w ? r[i] * w[i] : r[i]; /* :500 Residuals.mul(W) */
    si += (int64_t)lrintf(r[i]) * (int64_t)lrintf(r[i]);
  }
  if (sum_r2_out) *sum_r2_out = si;
  float inv_n = (float)(1.0 / (double)n); /* :499 */
  float e = (float)gemm_dot_width1(ar, r, 1, rw, n, 1, (double)inv_n);
  free(rw);
  return e;
}
float uwo_error(const float* r, const float* w, int n, int64_t* sum_r2_out) { return error_ar(t_ar, r, w, n, sum_r2_out); }

/* Tracker.cpp:554-561 (S2).  J <- w∘J ; r <- gain·r ; A = JᵀJ ; b = -Jᵀ(r∘w).
 * A: gemm(J, J, 1, GEMM_1_T), 6x6, len N — GEMMSingleMul's "d_size.width * sizeof <= 1600" branch (N <= 10000) and
 * GEMMBlockMul's non-transposed branch (N > 10000, do_acc carrying d_buf) both add the N products of an entry one after
 * the other in double.  b: "-Jacobians.t()" is MatOp::subtract(Scalar(0), T) = the materialised transpose scaled by -1,
 * times the materialised Residuals.mul(W): gemm(Jt, RW, alpha = -1) with a 6x1 result — gemm_dot_width1. */
static void normal_equations_ar(uwo_ar ar, const float* J, const float* r, const float* w, int n, float gain, float A[36], float b[6]) {
  double Ad[36];
  memset(Ad, 0, sizeof(Ad));
  float* Jw = (float*)malloc(sizeof(float) * 6 * (size_t)(n > 0 ? n : 1));
  float* rw = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    float wi = w ? w[i] : 1.0f;
    float* Jr = Jw + 6 * (size_t)i;
    for (int k = 0; k < 6; k++) Jr[k] = wi * J[6 * (size_t)i + k]; /* :556 */
    float rg = r[i] * gain;                                        /* :559 */
    rw[i] = rg * wi;                                               /* :561 Residuals.mul(W) */
    for (int a = 0; a < 6; a++)
      for (int c = 0; c < 6; c++) Ad[6 * a + c] += (double)Jr[a] * (double)Jr[c];
  }
  for (int k = 0; k < 36; k++) A[k] = (float)Ad[k];
  for (int k = 0; k < 6; k++) b[k] = (float)gemm_dot_width1(ar, Jw + k, 6, rw, n, 6, -1.0);
  free(Jw); free(rw);
}
void uwo_normal_equations(const float* J, const float* r, const float* w, int n, float gain, float A[36], float b[6]) {
  normal_equations_ar(t_ar, J, r, w, n, gain, A, b);
}

/* cv::Mat::inv() default DECOMP_LU on CV_32F (Tracker.cpp:564; S3): OpenCV 3.x hal LU —
 * partial pivoting on |A[j][i]|, singular if |pivot| < 10·FLT_EPSILON (result zeroed),
 * d = -1/pivot, row updates A[j][k] += alpha·A[i][k], the pivot slot keeps 1/pivot, back
 * substitution multiplies by it. */
int uwo_inv6(const float Ain[36], float X[36]) {
  const int m = 6;
  float A[36];
  memcpy(A, Ain, sizeof(A));
  for (int i = 0; i < 36; i++) X[i] = 0.0f;
  for (int i = 0; i < m; i++) X[7 * i] = 1.0f;
  const float eps = FLT_EPSILON * 10;
  for (int i = 0; i < m; i++) {
    int k = i;
    for (int j = i + 1; j < m; j++)
      if (fabsf(A[j * m + i]) > fabsf(A[k * m + i])) k = j;
    if (fabsf(A[k * m + i]) < eps) {
      for (int q = 0; q < 36; q++) X[q] = 0.0f;
      return 0;
    }
    if (k != i) {
      for (int j = i; j < m; j++) { float t = A[i * m + j]; A[i * m + j] = A[k * m + j]; A[k * m + j] = t; }
      for (int j = 0; j < m; j++) { float t = X[i * m + j]; X[i * m + j] = X[k * m + j]; X[k * m + j] = t; }
    }
    float d = -1.0f / A[i * m + i];
    for (int j = i + 1; j < m; j++) {
      float alpha = A[j * m + i] * d;
      for (int q = i + 1; q < m; q++) A[j * m + q] = A[j * m + q] + alpha * A[i * m + q];
      for (int q = 0; q < m; q++) X[j * m + q] = X[j * m + q] + alpha * X[i * m + q];
    }
    A[i * m + i] = -d;
  }
  for (int i = m - 1; i >= 0; i--)
    for (int j = 0; j < m; j++) {
      float s = X[i * m + j];
      for (int q = i + 1; q < m; q++) s = s - A[i * m + q] * X[q * m + j];
      X[i * m + j] = s * A[i * m + i];
    }
  return 1;
}

/* cv::solve(A, b, x, DECOMP_LU) on CV_32F, 6x6 with one right-hand side (G2): src.copyTo(a), src2.copyTo(dst),
 * hal::LU32f(a, astep, 6, dst, dstep, 1) = LUImpl<float>(.., eps = FLT_EPSILON*10); "if (!result) dst = Scalar(0)".
 * The same elimination as uwo_inv6 with the 6x1 right-hand side in place of the identity.  Returns 0 when singular. */
int uwo_solve6(const float Ain[36], const float bin[6], float x[6]) {
  const int m = 6;
  float A[36], b[6];
  memcpy(A, Ain, sizeof(A));
  memcpy(b, bin, sizeof(b));
  const float eps = FLT_EPSILON * 10;
  for (int i = 0; i < m; i++) {
    int k = i;
    for (int j = i + 1; j < m; j++)
      if (fabsf(A[j * m + i]) > fabsf(A[k * m + i])) k = j;
    if (fabsf(A[k * m + i]) < eps) {
      for (int q = 0; q < m; q++) x[q] = 0.0f;
      return 0;
    }
    if (k != i) {
      for (int j = i; j < m; j++) { float t = A[i * m + j]; A[i * m + j] = A[k * m + j]; A[k * m + j] = t; }
      float t = b[i]; b[i] = b[k]; b[k] = t;
    }
    float d = -1.0f / A[i * m + i];
    for (int j = i + 1; j < m; j++) {
      float alpha = A[j * m + i] * d;
      for (int q = i + 1; q < m; q++) A[j * m + q] = A[j * m + q] + alpha * A[i * m + q];
      b[j] = b[j] + alpha * b[i];
    }
    A[i * m + i] = -d;
  }
  for (int i = m - 1; i >= 0; i--) {
    float s = b[i];
    for (int q = i + 1; q < m; q++) s = s - A[i * m + q] * b[q];
    b[i] = s * A[i * m + i];
  }
  memcpy(x, b, sizeof(b));
  return 1;
}

/* Tracker.cpp:564 deltaMat = A.inv() * b.  G2: MatOp_Invert::matmul turns "A.inv() * b" into MatOp_Solve — cv::solve, no
 * inverse, no product (the MatExpr class documentation lists it: "A.inv([method]) * B (~ X: AX = B)").
 * Legacy (S3 + S4): the inverse formed first, then the 6x6·6x1 product accumulated in double. */
static void solve_delta_ar(uwo_ar ar, const float A[36], const float b[6], float delta[6]) {
  if (ar.arith != UWO_ARITH_LEGACY) {
    uwo_solve6(A, b, delta);
    return;
  }
  float Ai[36];
  uwo_inv6(A, Ai);
  for (int i = 0; i < 6; i++) {
    double s = 0.0;
    for (int j = 0; j < 6; j++) s += (double)Ai[6 * i + j] * (double)b[j];
    delta[i] = (float)s;
  }
}
void uwo_solve_delta(const float A[36], const float b[6], float delta[6]) { solve_delta_ar(t_ar, A, b, delta); }

/* ------------------------------------------------------------------------------------------ */
/* Tracker::EstimatePose, Tracker.cpp:362-597                                                   */
/* ------------------------------------------------------------------------------------------ */

int uwo_estimate_pose(const uwo_params* p, const uwo_frame* prev, const uwo_frame* cur,
                      float pose_out[7], uwo_trace* trace, int32_t* n_trace) {
  return uwo_estimate_pose_points(p, prev, cur, NULL, NULL, pose_out, trace, n_trace);
}

/* Same loop over explicit per-level point tables (Frame::candidatePoints_[lvl], N x 4): what EstimatePose /
 * EstimatePoseFeatures consume whichever producer filled them (ObtainAllPoints, ObtainCandidatePoints,
 * ObtainPatchesPoints).  tables == NULL: dense tables (ObtainAllPoints). */
int uwo_estimate_pose_points(const uwo_params* p, const uwo_frame* prev, const uwo_frame* cur,
                             const float* const* tables, const int32_t* n_points,
                             float pose_out[7], uwo_trace* trace, int32_t* n_trace) {
  /* the call's arithmetic: the set from the parameters; the gemm fold from the parameters, or the calling thread's default */
  const uwo_ar ar = {p->arith == UWO_ARITH_LEGACY ? UWO_ARITH_LEGACY : UWO_ARITH_OPENCV, (p->gemm_fold || t_ar.fold) ? 1 : 0,
                     (p->trig == UWO_TRIG_LIBM || t_ar.trig == UWO_TRIG_LIBM) ? UWO_TRIG_LIBM : UWO_TRIG_ROUNDED};
  if (p->first_level >= p->n_levels || p->last_level < 0 || p->last_level > p->first_level) return UWO_ERR_INVALID_ARG;
  int cap = (trace && n_trace) ? *n_trace : 0;
  int nt = 0;
  int status = UWO_OK;
  float pose[7];
  uwo_se3_identity(pose); /* :385 */

  uwo_level L0;
  uwo_level_intrinsics(p, p->last_level, &L0);
  size_t nmax = (size_t)L0.w * L0.h;
  if (tables)
    for (int l = p->last_level; l <= p->first_level; l++)
      if ((size_t)n_points[l] > nmax) nmax = (size_t)n_points[l];
  if (nmax == 0) nmax = 1;
  float* pts = (float*)malloc(sizeof(float) * 4 * nmax);
  float* warped = (float*)malloc(sizeof(float) * 4 * nmax);
  float* J = (float*)malloc(sizeof(float) * 6 * nmax);
  float* r = (float*)malloc(sizeof(float) * nmax);
  float* wts = (float*)malloc(sizeof(float) * nmax);

  for (int lvl = p->first_level; lvl >= p->last_level && status == UWO_OK; lvl--) { /* :389 */
    uwo_level L;
    uwo_level_intrinsics(p, lvl, &L);
    int n = L.w * L.h;
    float last_error = p->initial_error; /* :393 */
    if (tables) {
      n = n_points[lvl];
      if (n > 0) memcpy(pts, tables[lvl], sizeof(float) * 4 * (size_t)n); /* :401 candidatePoints_[lvl].clone() */
    } else {
      uwo_dense_points_ex(p->has_depth ? prev->depth[lvl] : NULL, L.iw, L.w, L.h, lvl, p->depth_scale, pts); /* :401 */
    }

    for (int k = 0; k < p->max_iters; k++) { /* :414 */
      warp_ar(ar, pts, n, pose, &L, warped); /* :422 */
      int nv = residual_jacobian_ar(ar, prev->img[lvl], cur->img[lvl], prev->gx[lvl], prev->gy[lvl], pts, warped, n, &L,
                                        p->z_factor, p->angle_factor, p->sampler, J, r, NULL);
      if (nv == 0) { status = UWO_ERR_NO_VALID_POINTS; break; } /* reference: cv::Exception on empty Mat product */
      const float* W = NULL;
      if (p->weights == UWO_WEIGHTS_TUKEY_REFERENCE) { uwo_tukey_weights(r, nv, wts); W = wts; } /* :495-496 */
      else if (p->weights == UWO_WEIGHTS_HUBER) { uwo_huber_weights(r, nv, wts); W = wts; }        /* extension */
      int64_t sr2 = 0;
      float error = error_ar(ar, r, W, nv, &sr2); /* :499-502 */

      uwo_trace* tr = (nt < cap) ? &trace[nt] : NULL;
      if (tr) {
        memset(tr, 0, sizeof(*tr));
        tr->level = lvl; tr->iter = k; tr->n_valid = nv; tr->sum_r2 = sr2; tr->error = error;
      }
      int exit_now = 0;
      if (p->early_exit) { /* :508 */
        if (error >= last_error || k == p->max_iters - 1 || fabsf(error - last_error) < p->epsilon) exit_now = 1;
      }
      if (exit_now) {
        if (tr) { tr->exited = 1; memcpy(tr->pose, pose, sizeof(pose)); }
        nt++;
        break;
      }
      last_error = error; /* :529 */

      float A[36], b[6], delta[6];
      normal_equations_ar(ar, J, r, W, nv, p->gain, A, b); /* :554-561 */
      solve_delta_ar(ar, A, b, delta);                 /* :564 */
      float dT[7], np[7];
      se3_exp_ar(ar, delta, dT);                       /* :574 */
      uwo_se3_mul(pose, dT, np);
      memcpy(pose, np, sizeof(pose));
      if (tr) {
        memcpy(tr->A, A, sizeof(A)); memcpy(tr->b, b, sizeof(b)); memcpy(tr->delta, delta, sizeof(delta));
        memcpy(tr->pose, pose, sizeof(pose));
      }
      nt++;
    }
    if (status != UWO_OK) break;
    if (lvl != 0) { /* :580 */
      if (uwo_se3_handoff(pose, p->handoff_scale_t) != UWO_OK) status = UWO_ERR_INVALID_ARG;
    }
  }
  free(pts); free(warped); free(J); free(r); free(wts);
  memcpy(pose_out, pose, sizeof(pose)); /* :595 */
  if (n_trace) *n_trace = nt < cap ? nt : cap;
  return status;
}

/* System::AddFrame pyramid loop + Tracker::ApplyGradient + EstimatePose for one pair. */
int uwo_align_pair(const uwo_params* p, const uint8_t* ref_gray, const uint8_t* tgt_gray,
                   const uint16_t* ref_depth, const uint16_t* tgt_depth,
                   float pose_out[7], uwo_trace* trace, int32_t* n_trace) {
  (void)tgt_depth; /* the tracker reads only the previous frame's depth (Tracker.cpp:401) */
  return uwo_align_pair_points(p, ref_gray, tgt_gray, ref_depth, NULL, NULL, pose_out, trace, n_trace);
}

int uwo_align_pair_points(const uwo_params* p, const uint8_t* ref_gray, const uint8_t* tgt_gray,
                          const uint16_t* ref_depth, const float* const* tables, const int32_t* n_points,
                          float pose_out[7], uwo_trace* trace, int32_t* n_trace) {
  if (p->n_levels < 1 || p->n_levels > UWO_MAX_LEVELS) return UWO_ERR_INVALID_ARG;
  if (p->width < 1 || p->height < 1 || (p->width >> (p->n_levels - 1)) < 1 || (p->height >> (p->n_levels - 1)) < 1)
    return UWO_ERR_INVALID_ARG; /* every level needs at least one grid point */
  uwo_frame fr[2];
  memset(fr, 0, sizeof(fr));
  void* owned[2][UWO_MAX_LEVELS][4];
  memset(owned, 0, sizeof(owned));
  const uint8_t* gray[2] = {ref_gray, tgt_gray};
  for (int f = 0; f < 2; f++) {
    for (int l = 0; l < p->n_levels; l++) {
      uwo_level Ll;
      uwo_level_intrinsics(p, l, &Ll);
      int w = Ll.iw, h = Ll.ih; /* images_[l].cols / rows: the resize chain (System.cpp:246-251) */
      size_t n = (size_t)w * h;
      if (l == 0) {
        fr[f].img[0] = gray[f];
        if (f == 0 && p->has_depth) fr[f].depth[0] = ref_depth;
      } else {
        uint8_t* im = (uint8_t*)malloc(n);
        uwo_level Lp;
        uwo_level_intrinsics(p, l - 1, &Lp);
        uwo_resize_half_u8(fr[f].img[l - 1], Lp.iw, Lp.ih, im);
        fr[f].img[l] = im; owned[f][l][0] = im;
        if (f == 0 && p->has_depth) {
          uint16_t* d = (uint16_t*)malloc(n * 2);
          uwo_resize_half_u16(fr[f].depth[l - 1], Lp.iw, Lp.ih, d);
          fr[f].depth[l] = d; owned[f][l][1] = d;
        }
      }
      if (f == 0) { /* gradients of the previous frame only are consumed (Tracker.cpp:407-408) */
        int16_t* gx = (int16_t*)malloc(n * 2);
        int16_t* gy = (int16_t*)malloc(n * 2);
        uwo_scharr3(fr[f].img[l], w, h, gx, gy);
        fr[f].gx[l] = gx; fr[f].gy[l] = gy; owned[f][l][2] = gx; owned[f][l][3] = gy;
      }
    }
  }
  int st = uwo_estimate_pose_points(p, &fr[0], &fr[1], tables, n_points, pose_out, trace, n_trace);
  for (int f = 0; f < 2; f++)
    for (int l = 0; l < UWO_MAX_LEVELS; l++)
      for (int k = 0; k < 4; k++) free(owned[f][l][k]);
  return st;
}

/* Tracker::ObtainPatchesPoints, Tracker.cpp:1178-1257 (level 0 only): 11x11 patches ("patch_size_ - 1 / 2" = 5, :1190)
 * around at most 200 keypoints, x-major inside a patch; with depth the whole patch takes the key point's depth
 * (at<short>, != 0 test, :1202-1204).  kp: n_kp x 2 (x, y).  Returns the number of points written (<= cap). */
int uwo_patch_points(const float* kp, int n_kp, const uint16_t* depth0, int w, int h, float* pts, int cap) {
  const float factor_depth = 0.0002f;
  const float factor_lvl = (float)(1.0 / pow(2.0, 0.0));
  const int start_point = 5 - 1 / 2;
  int n = 0;
  for (int q = 0; q < (n_kp < 200 ? n_kp : 200); q++) {
    const float x = kp[2 * q], y = kp[2 * q + 1];
    float z = 1.0f;
    if (depth0) {
      const int16_t d = (int16_t)depth0[(size_t)(int)y * w + (int)x];
      if (d == 0) continue;
      z = (float)d * factor_depth * factor_lvl; /* :1204 */
    }
    for (int i = (int)(x - (float)start_point); (float)i <= x + (float)start_point; i++)
      for (int j = (int)(y - (float)start_point); (float)j <= y + (float)start_point; j++)
        if (i > 0 && i < w && j > 0 && j < h) {
          if (n < cap) { pts[4 * n] = (float)i; pts[4 * n + 1] = (float)j; pts[4 * n + 2] = z; pts[4 * n + 3] = 1.0f; }
          n++;
        }
  }
  return n;
}

/* Tracker::AddPatchPointsFeatures, Tracker.cpp:599-629: clone of the table (:601), then per point the patch cells around
 * (round(x), round(y)) (:607-608), x outer / y inner (:611-612), strictly inside the level and not the centre (:613), with the
 * point's z and w = 1 (Mat::ones, :614-617).  start_point = (patch_size_ - 1) / 2 (:602; patch_size_ = 5, :274).
 * Returns the full count; at most cap rows are written. */
int uwo_add_patch_points(const float* pts_in, int n, int w, int h, int patch_size, float* pts, int cap) {
  const int start_point = (patch_size - 1) / 2;
  int m = 0;
  for (int q = 0; q < n; q++, m++)
    if (m < cap) memcpy(pts + 4 * (size_t)m, pts_in + 4 * (size_t)q, 16);
  for (int q = 0; q < n; q++) {
    const float x = roundf(pts_in[4 * q]), y = roundf(pts_in[4 * q + 1]), z = pts_in[4 * q + 2];
    for (int i = (int)(x - (float)start_point); (float)i <= x + (float)start_point; i++)
      for (int j = (int)(y - (float)start_point); (float)j <= y + (float)start_point; j++)
        if (i > 0 && i < w && j > 0 && j < h && !((float)i == x && (float)j == y)) {
          if (m < cap) { pts[4 * m] = (float)i; pts[4 * m + 1] = (float)j; pts[4 * m + 2] = z; pts[4 * m + 3] = 1.0f; }
          m++;
        }
  }
  return m;
}

/* Tracker::ObtainCandidatePoints, Tracker.cpp:1314-1398, one level: mask = gradient_ > mean(gradient_) + threshold
 * (cuda::meanStdDev + cuda::threshold THRESH_BINARY, :1324-1329), points pushed x-major (x outer, y inner, :1334-1335),
 * z = 1 without depth.  With depth the reference indexes the 16-bit image through at<uchar> (:1339, :1344): byte x of
 * row y, scaled by 0.0002 with no level factor — reproduced as is.  Returns the count (<= cap written). */
int uwo_candidate_points(const uint8_t* mag, const uint16_t* depth, int w, int h, double threshold, float* pts, int cap) {
  return uwo_candidate_points_ex(mag, depth, w, h, w, h, threshold, pts, cap);
}

/* the same with the level's image (iw x ih: gradient_[lvl] and depths_[lvl], whose mean cuda::meanStdDev takes over the whole
 * Mat, :1324) and its point grid (w x h = w_[lvl] x h_[lvl], the loops of :1334-1335) apart */
int uwo_candidate_points_ex(const uint8_t* mag, const uint16_t* depth, int iw, int ih, int w, int h, double threshold, float* pts,
                            int cap) {
  double sum = 0.0;
  for (size_t i = 0; i < (size_t)iw * ih; i++) sum += mag[i];
  const double thres = sum / (double)((size_t)iw * ih) + threshold;
  int n = 0;
  for (int x = 0; x < w; x++)
    for (int y = 0; y < h; y++) {
      if (!((double)mag[(size_t)y * iw + x] > thres)) continue;
      float z = 1.0f;
      if (depth) {
        const uint8_t b = ((const uint8_t*)(depth + (size_t)y * iw))[x];
        if (b == 0) continue;
        z = (float)b * 0.0002f;
      }
      if (n < cap) { pts[4 * n] = (float)x; pts[4 * n + 1] = (float)y; pts[4 * n + 2] = z; pts[4 * n + 3] = 1.0f; }
      n++;
    }
  return n;
}

/* Visualizer.cpp:304-325.  SE3(quaternion, t) normalises the quaternion (so3.hpp:431-439, 270-276). */
void uwo_accumulate_trajectory(const float* poses, int n, const float start[7], float t_scale, int reference_axes,
                               float* traj_out) {
  float prev[7];
  memcpy(prev, start, sizeof(prev));
  for (int i = 0; i < n; i++) {
    const float* p = poses + 7 * (size_t)i;
    float cur[7];
    float len = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
    cur[0] = p[0] / len; cur[1] = p[1] / len; cur[2] = p[2] / len; cur[3] = p[3] / len;
    cur[4] = t_scale * p[4]; cur[5] = t_scale * p[5]; cur[6] = t_scale * p[6]; /* :307-309 */
    float fin[7];
    uwo_se3_mul(prev, cur, fin); /* :313 */
    memcpy(prev, fin, sizeof(prev)); /* :325 */
    float* o = traj_out + 7 * (size_t)i;
    o[0] = fin[0]; o[1] = fin[1]; o[2] = fin[2]; o[3] = fin[3];
    if (reference_axes) { o[4] = -fin[6]; o[5] = -fin[4]; o[6] = -fin[5]; } /* :318-320 */
    else { o[4] = fin[4]; o[5] = fin[5]; o[6] = fin[6]; }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* frame ingest (f-2).  OpenCV 3.2: cvUndistortPoints (5 fixed iterations), icvGetRectangles,   */
/* cvGetOptimalNewCameraMatrix, initUndistortRectifyMap, remap INTER_LINEAR fixed point.        */
/* ------------------------------------------------------------------------------------------ */

static void undistort_point(double u, double v, const double K[4], const double k[4], const double P[4], float* ox, float* oy) {
  double x = (u - K[2]) / K[0], y = (v - K[3]) / K[1];
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    double r2 = x * x + y * y;
    double icdist = 1.0 / (1.0 + ((0.0 * r2 + k[1]) * r2 + k[0]) * r2);
    double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
    double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  *ox = (float)(x * P[0] + P[2]);
  *oy = (float)(y * P[1] + P[3]);
}

/* CameraModel.cpp:89: getOptimalNewCameraMatrix(K, dist, Size(in), 1.0, Size(out), nullptr, false) */
void uwo_optimal_new_camera_matrix(const float Kf[4], const float distf[4], int in_w, int in_h, double alpha, int new_w,
                                   int new_h, double newK[4]) {
  const double K[4] = {Kf[0], Kf[1], Kf[2], Kf[3]}, k[4] = {distf[0], distf[1], distf[2], distf[3]};
  const int N = 9;
  float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
  float oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
  for (int y = 0; y < N; y++)
    for (int x = 0; x < N; x++) {
      float px = (float)x * in_w / (N - 1), py = (float)y * in_h / (N - 1);
      float qx, qy;
      const double Pid[4] = {1.0, 1.0, 0.0, 0.0}; /* icvGetRectangles(K, dist, R = 0, newK = 0, ...): normalised output */
      undistort_point(px, py, K, k, Pid, &qx, &qy);
      if (qx < oX0) oX0 = qx;
      if (qx > oX1) oX1 = qx;
      if (qy < oY0) oY0 = qy;
      if (qy > oY1) oY1 = qy;
      if (x == 0 && qx > iX0) iX0 = qx;
      if (x == N - 1 && qx < iX1) iX1 = qx;
      if (y == 0 && qy > iY0) iY0 = qy;
      if (y == N - 1 && qy < iY1) iY1 = qy;
    }
  const float iw = iX1 - iX0, ih = iY1 - iY0, ow = oX1 - oX0, oh = oY1 - oY0;
  double fx0 = (float)(new_w - 1) / iw, fy0 = (float)(new_h - 1) / ih;
  double cx0 = -fx0 * iX0, cy0 = -fy0 * iY0;
  double fx1 = (float)(new_w - 1) / ow, fy1 = (float)(new_h - 1) / oh;
  double cx1 = -fx1 * oX0, cy1 = -fy1 * oY0;
  newK[0] = fx0 * (1 - alpha) + fx1 * alpha;
  newK[1] = fy0 * (1 - alpha) + fy1 * alpha;
  newK[2] = cx0 * (1 - alpha) + cx1 * alpha;
  newK[3] = cy0 * (1 - alpha) + cy1 * alpha;
}

/* CameraModel.cpp:90: initUndistortRectifyMap(K, dist, Mat(), newK, Size(out), CV_16SC2, map1, map2).
 * INTER_BITS = 5: map1 = integer source coordinates, map2 = (fy << 5) | fx sub-pixel index. */
void uwo_init_undistort_maps(const float Kf[4], const float distf[4], const double newK[4], int w, int h, int16_t* map1,
                             uint16_t* map2) {
  const double fx = Kf[0], fy = Kf[1], u0 = Kf[2], v0 = Kf[3];
  const double k1 = distf[0], k2 = distf[1], p1 = distf[2], p2 = distf[3];
  /* ir = inverse of [newfx 0 newcx; 0 newfy newcy; 0 0 1] */
  const double ir0 = 1.0 / newK[0], ir2 = -newK[2] / newK[0], ir4 = 1.0 / newK[1], ir5 = -newK[3] / newK[1];
  for (int i = 0; i < h; i++) {
    double _x = i * 0.0 + ir2, _y = i * ir4 + ir5, _w = 1.0;
    for (int j = 0; j < w; j++, _x += ir0) {
      double ww = 1.0 / _w, x = _x * ww, y = _y * ww;
      double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
      double kr = (1 + ((0.0 * r2 + k2) * r2 + k1) * r2) / (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2);
      double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
      double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
      double u = fx * xd + u0, v = fy * yd + v0;
      double su = u * 32.0, sv = v * 32.0;
      long iu = lrint(su), iv = lrint(sv); /* saturate_cast<int>(double) = cvRound */
      int su_i = (int)iu >> 5, sv_i = (int)iv >> 5;
      if (su_i > 32767) su_i = 32767;
      if (su_i < -32768) su_i = -32768;
      if (sv_i > 32767) sv_i = 32767;
      if (sv_i < -32768) sv_i = -32768;
      map1[2 * ((size_t)i * w + j)] = (int16_t)su_i;
      map1[2 * ((size_t)i * w + j) + 1] = (int16_t)sv_i;
      map2[(size_t)i * w + j] = (uint16_t)(((int)iv & 31) * 32 + ((int)iu & 31));
    }
  }
}

/* System.cpp:152 / :233 remap(src, dst, map1, map2, INTER_LINEAR) with BORDER_CONSTANT 0.  Fixed point: weights
 * (32-fx)(32-fy)·32 etc. (sum 32768, INTER_REMAP_COEF_BITS = 15), value = (Σ w·p + 2^14) >> 15. */
void uwo_remap_linear(const uint8_t* src, int sw, int sh, const int16_t* map1, const uint16_t* map2, int dw, int dh,
                      uint8_t* dst) {
  for (size_t q = 0; q < (size_t)dw * dh; q++) {
    int sx = map1[2 * q], sy = map1[2 * q + 1];
    int fxy = map2[q] & 1023, fx = fxy & 31, fy = fxy >> 5;
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    int p00 = (sx >= 0 && sx < sw && sy >= 0 && sy < sh) ? src[(size_t)sy * sw + sx] : 0;
    int p01 = (sx + 1 >= 0 && sx + 1 < sw && sy >= 0 && sy < sh) ? src[(size_t)sy * sw + sx + 1] : 0;
    int p10 = (sx >= 0 && sx < sw && sy + 1 >= 0 && sy + 1 < sh) ? src[(size_t)(sy + 1) * sw + sx] : 0;
    int p11 = (sx + 1 >= 0 && sx + 1 < sw && sy + 1 >= 0 && sy + 1 < sh) ? src[(size_t)(sy + 1) * sw + sx + 1] : 0;
    int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
    dst[q] = (uint8_t)(v > 255 ? 255 : v);
  }
}

/* System::CalculateROI, System.cpp:148-191: roi = {x, y, w, h} with Rect(p1, p2) semantics (w = p2.x - p1.x). */
void uwo_calculate_roi(const uint8_t* und, int w, int h, int32_t roi[4]) {
  int x_middle = (int)((w - 1) * 0.5), y_middle = (int)((h - 1) * 0.5);
  int p1x = 0, p1y = 0, p2x = w - 1, p2y = h - 1;
  while (p1x < w - 1 && und[(size_t)y_middle * w + p1x] == 0) p1x++;
  while (p2x > 0 && und[(size_t)y_middle * w + p2x] == 0) p2x--;
  while (p1y < h - 1 && und[(size_t)p1y * w + x_middle] == 0) p1y++;
  while (p2y > 0 && und[(size_t)p2y * w + x_middle] == 0) p2y--;
  p1x += 5; p2x -= 5; p1y += 5; p2y -= 5; /* :180-183 */
  roi[0] = p1x; roi[1] = p1y; roi[2] = p2x - p1x; roi[3] = p2y - p1y;
}

/* ------------------------------------------------------------------------------------------ */
/* LS, LeastSquares.cpp:30-209                                                                  */
/* ------------------------------------------------------------------------------------------ */

void uwo_ls_initialize(uwo_ls* ls) { memset(ls, 0, sizeof(*ls)); } /* :30-37 */

/* :204-209.  Eigen evaluates (J·Jᵀ)·weight coefficient-wise; b -= J·(res·weight). */
void uwo_ls_update(uwo_ls* ls, const float J[6], float res, float weight) {
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) ls->A[6 * i + j] = ls->A[6 * i + j] + (J[i] * J[j]) * weight;
  float rw = res * weight;
  for (int i = 0; i < 6; i++) ls->b[i] = ls->b[i] - J[i] * rw;
  ls->error = ls->error + res * res * weight;
  ls->num_constraints += 1;
}

/* :148-202.  28 accumulators x 4 lanes: (J_i·w)·J_j ; (res·w)·J_i ; (res·w)·res.  The reference adds 6
 * to num_constraints per 4 points (:201, quirk C-6); quirk_plus6 = 0 adds the correct 4. */
void uwo_ls_update4(uwo_ls* ls, const float J[6][4], const float res[4], const float weight[4], int quirk_plus6) {
  int s = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++, s++)
      for (int l = 0; l < 4; l++) ls->sse[4 * s + l] = ls->sse[4 * s + l] + (J[i][l] * weight[l]) * J[j][l];
  for (int i = 0; i < 6; i++, s++)
    for (int l = 0; l < 4; l++) ls->sse[4 * s + l] = ls->sse[4 * s + l] + (res[l] * weight[l]) * J[i][l];
  for (int l = 0; l < 4; l++) ls->sse[4 * s + l] = ls->sse[4 * s + l] + (res[l] * weight[l]) * res[l];
  ls->num_constraints += quirk_plus6 ? 6 : 4;
}

/* :39-139.  Lanes folded left to right, added into A's first-row (upper) / other-rows (lower)
 * slot, mirrored to the symmetric slot; b -= fold; error += fold. */
void uwo_ls_finish_no_divide(uwo_ls* ls) {
  int s = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++, s++) {
      const float* a = ls->sse + 4 * s;
      float f = a[0] + a[1] + a[2] + a[3];
      if (i == j) {
        ls->A[6 * i + i] = ls->A[6 * i + i] + f;
      } else if (i == 0) {
        ls->A[6 * 0 + j] = ls->A[6 * 0 + j] + f; /* A(j,0) = (A(0,j) += f) */
        ls->A[6 * j + 0] = ls->A[6 * 0 + j];
      } else {
        ls->A[6 * j + i] = ls->A[6 * j + i] + f; /* A(i,j) = (A(j,i) += f) */
        ls->A[6 * i + j] = ls->A[6 * j + i];
      }
    }
  for (int i = 0; i < 6; i++, s++) {
    const float* a = ls->sse + 4 * s;
    ls->b[i] = ls->b[i] - (a[0] + a[1] + a[2] + a[3]);
  }
  const float* a = ls->sse + 4 * s;
  ls->error = ls->error + (a[0] + a[1] + a[2] + a[3]);
}

/* :141-146 */
void uwo_ls_finish(uwo_ls* ls) {
  uwo_ls_finish_no_divide(ls);
  float n = (float)ls->num_constraints;
  for (int i = 0; i < 36; i++) ls->A[i] = ls->A[i] / n;
  for (int i = 0; i < 6; i++) ls->b[i] = ls->b[i] / n;
  ls->error = ls->error / n;
}
