// uwt_math.h — device-side SE(3) / 6x6 solve used by the Gauss-Newton update kernel.
//
// Restates, for gfx950, the arithmetic the reference obtains from Sophus::SE3f / Eigen::Quaternionf and
// cv::Mat::inv(): thirdparty/sophus/se3.hpp:253-268, 317-321, 723-744; so3.hpp:270-288, 338-354, 534-566;
// src/Tracker.cpp:564, 574, 580-590.  All f32, no FMA contraction (build with -ffp-contract=off); sin/cos are
// evaluated in f64 and rounded once so the result does not depend on a libm's f32 kernels.
#pragma once

#include <hip/hip_runtime.h>

namespace uwt {

struct Pose {  // unit quaternion (x y z w) + translation, the storage of Sophus::SE3f
  float q[4];
  float t[3];
};

// sin / cos of a float, correctly rounded in practice: evaluated in f64 and rounded once.  The angles of a Gauss-Newton
// step are small, and up to |x| <= 0.5 the Taylor series to x^17 / x^16 is exact to 1e-20 relative — a dozen dependent
// f64 operations instead of the library routine's several hundred instructions (argument reduction, large-angle paths)
// on the serial tail of the update; larger angles take the library routine.
__device__ inline double sin_small(double x) {
  const double z = x * x;
  double p = -1.0 / 355687428096000.0;            // -1/17!
  p = __builtin_fma(p, z, 1.0 / 1307674368000.0);  //  1/15!
  p = __builtin_fma(p, z, -1.0 / 6227020800.0);    // -1/13!
  p = __builtin_fma(p, z, 1.0 / 39916800.0);       //  1/11!
  p = __builtin_fma(p, z, -1.0 / 362880.0);        // -1/9!
  p = __builtin_fma(p, z, 1.0 / 5040.0);           //  1/7!
  p = __builtin_fma(p, z, -1.0 / 120.0);           // -1/5!
  p = __builtin_fma(p, z, 1.0 / 6.0);              //  1/3!  (sign folded below)
  return __builtin_fma(-(x * z), p, x);            // x - x^3 (1/6 - z/120 + ...)
}
__device__ inline double cos_small(double x) {
  const double z = x * x;
  double p = 1.0 / 20922789888000.0;               //  1/16!
  p = __builtin_fma(p, z, -1.0 / 87178291200.0);   // -1/14!
  p = __builtin_fma(p, z, 1.0 / 479001600.0);      //  1/12!
  p = __builtin_fma(p, z, -1.0 / 3628800.0);       // -1/10!
  p = __builtin_fma(p, z, 1.0 / 40320.0);          //  1/8!
  p = __builtin_fma(p, z, -1.0 / 720.0);           // -1/6!
  p = __builtin_fma(p, z, 1.0 / 24.0);             //  1/4!
  p = __builtin_fma(p, z, -0.5);                   // -1/2!
  return __builtin_fma(p, z, 1.0);
}
__device__ inline float sin_r(float x) { return fabsf(x) <= 0.5f ? (float)sin_small((double)x) : (float)sin((double)x); }
__device__ inline float cos_r(float x) { return fabsf(x) <= 0.5f ? (float)cos_small((double)x) : (float)cos((double)x); }

__device__ inline void pose_identity(Pose& p) {
  p.q[0] = 0.f; p.q[1] = 0.f; p.q[2] = 0.f; p.q[3] = 1.f;
  p.t[0] = 0.f; p.t[1] = 0.f; p.t[2] = 0.f;
}

// Eigen Quaternion::toRotationMatrix, row-major 3x3 (so3.hpp:286-288)
__device__ inline void quat_to_rot(const float q[4], float R[9]) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}

// SE3f::matrix3x4 (se3.hpp:263-268): rows of [R | t]
__device__ inline void pose_to_T12(const Pose& p, float T[12]) {
  float R[9];
  quat_to_rot(p.q, R);
  T[0] = R[0]; T[1] = R[1]; T[2] = R[2];  T[3] = p.t[0];
  T[4] = R[3]; T[5] = R[4]; T[6] = R[5];  T[7] = p.t[1];
  T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = p.t[2];
}

// Hamilton product, coefficient order x y z w (Eigen generic quat_product)
__device__ inline void quat_mul(const float a[4], const float b[4], float o[4]) {
  const float ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const float bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

// QuaternionBase::_transformVector (so3.hpp:320-322)
__device__ inline void quat_rotate(const float q[4], const float v[3], float o[3]) {
  float ux = q[1] * v[2] - q[2] * v[1];
  float uy = q[2] * v[0] - q[0] * v[2];
  float uz = q[0] * v[1] - q[1] * v[0];
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const float cx = q[1] * uz - q[2] * uy;
  const float cy = q[2] * ux - q[0] * uz;
  const float cz = q[0] * uy - q[1] * ux;
  o[0] = (v[0] + q[3] * ux) + cx;
  o[1] = (v[1] + q[3] * uy) + cy;
  o[2] = (v[2] + q[3] * uz) + cz;
}

constexpr float kSophusEps = 1e-5f;  // Constants<float>::epsilon, common.hpp:154-158

// SE3f::exp (se3.hpp:723-744) with SO3f::expAndTheta (so3.hpp:534-566); xi = [upsilon, omega].
// sh, ch = sin, cos of theta / 2 and st, ct = sin, cos of theta, each evaluated in f64 and rounded once (S5).
__device__ inline void se3_exp_with(const float xi[6], float theta_sq, float theta, float sh, float ch, float st, float ct,
                                    Pose& out) {
  const float o0 = xi[3], o1 = xi[4], o2 = xi[5];
  float imag, real;
  if (theta < kSophusEps) {
    const float theta_po4 = theta_sq * theta_sq;
    imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
    real = 1.f - (float)(1.0 / 8.0) * theta_sq + (float)(1.0 / 384.0) * theta_po4;
  } else {
    imag = sh / theta;
    real = ch;
  }
  out.q[0] = imag * o0; out.q[1] = imag * o1; out.q[2] = imag * o2; out.q[3] = real;

  const float Om[9] = {0.f, -o2, o1, o2, 0.f, -o0, -o1, o0, 0.f};  // SO3::hat, so3.hpp:618-627
  float V[9];
  if (theta < kSophusEps) {
    quat_to_rot(out.q, V);  // se3.hpp:734
  } else {
    float Om2[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        Om2[3 * i + j] = (Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j]) + Om[3 * i + 2] * Om[6 + j];
    const float tsq = theta * theta;
    const float c1 = (1.f - ct) / tsq;
    const float c2 = (theta - st) / (tsq * theta);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const float id = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
      V[i] = (id + c1 * Om[i]) + c2 * Om2[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 3; i++) out.t[i] = (V[3 * i] * xi[0] + V[3 * i + 1] * xi[1]) + V[3 * i + 2] * xi[2];
}

__device__ inline void se3_exp(const float xi[6], Pose& out) {
  const float theta_sq = xi[3] * xi[3] + xi[4] * xi[4] + xi[5] * xi[5];
  const float theta = sqrtf(theta_sq);
  const float half_theta = 0.5f * theta;
  float sh = 0.f, ch = 1.f, st = 0.f, ct = 1.f;
  if (!(theta < kSophusEps)) { sh = sin_r(half_theta); ch = cos_r(half_theta); st = sin_r(theta); ct = cos_r(theta); }
  se3_exp_with(xi, theta_sq, theta, sh, ch, st, ct, out);
}

// threadIdx.x behind an opaque statement.  Whatever is derived from it (lane roles, LDS addresses of the reduction and of the
// update) is then computed where it is used: read plainly, the compiler hoists those values out of a loop that encloses an
// evaluation and its update (k_coarse), where they stay live through the residual loop — two dozen registers of the kernel's
// budget for a few shifts per evaluation.
__device__ __forceinline__ unsigned thread_here() {
  unsigned t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// The same on a full wave holding uniform values (the update's tail): the four transcendentals — the longest dependent
// stretch of the tail — are taken in one pass, lane 0 on theta / 2 and lane 1 on theta, each by the sin and the cos of its
// own argument, instead of four evaluations one after the other.  Same functions, same results.
__device__ inline void se3_exp_wave(const float xi[6], Pose& out) {
  const float theta_sq = xi[3] * xi[3] + xi[4] * xi[4] + xi[5] * xi[5];
  const float theta = sqrtf(theta_sq);
  const float half_theta = 0.5f * theta;
  const int lane = (int)(thread_here() & 63u);
  const float arg = (lane & 1) ? theta : half_theta;
  const float s = sin_r(arg), c = cos_r(arg);
  const float sh = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s), 0));
  const float ch = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(c), 0));
  const float st = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s), 1));
  const float ct = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(c), 1));
  se3_exp_with(xi, theta_sq, theta, sh, ch, st, ct, out);
}

// SE3f::operator*= (se3.hpp:317-321) and SO3f::operator*= with the 2/(1+|q|²) renormalisation (so3.hpp:338-354)
__device__ inline void se3_mul(const Pose& a, const Pose& b, Pose& out) {
  float rt[3];
  quat_rotate(a.q, b.t, rt);
  const float t0 = a.t[0] + rt[0], t1 = a.t[1] + rt[1], t2 = a.t[2] + rt[2];
  float q[4];
  quat_mul(a.q, b.q, q);
  const float sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sn != 1.f) {
    const float s = 2.f / (1.f + sn);
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
  }
  out.q[0] = q[0]; out.q[1] = q[1]; out.q[2] = q[2]; out.q[3] = q[3];
  out.t[0] = t0; out.t[1] = t1; out.t[2] = t2;
}

// src/Tracker.cpp:580-590: q.xyz *= 2; SE3(q, t) re-normalises (so3.hpp:270-276). Returns false where
// SOPHUS_ENSURE would abort (|q| < eps).
__device__ inline bool se3_handoff(Pose& p, bool scale_t) {
  const float x = p.q[0] * 2.f, y = p.q[1] * 2.f, z = p.q[2] * 2.f, w = p.q[3];
  const float len = sqrtf(x * x + y * y + z * z + w * w);
  if (!(len >= kSophusEps)) return false;
  p.q[0] = x / len; p.q[1] = y / len; p.q[2] = z / len; p.q[3] = w / len;
  if (scale_t) { p.t[0] *= 2.f; p.t[1] *= 2.f; p.t[2] *= 2.f; }
  return true;
}

// cv::Mat::inv() (DECOMP_LU, CV_32F) as used at src/Tracker.cpp:564: Gaussian elimination with partial
// pivoting, pivot threshold 10·FLT_EPSILON, singular ⇒ zero matrix.  Returns false when singular.
__device__ inline bool inv6_lu(const float Ain[36], float X[36]) {
  // Fully unrolled with compile-time indices (the pivot row is applied through predicated row swaps), so that the
  // two 6x6 matrices live in registers instead of scratch memory.
  float A[36];
#pragma unroll
  for (int i = 0; i < 36; i++) { A[i] = Ain[i]; X[i] = 0.f; }
#pragma unroll
  for (int i = 0; i < 6; i++) X[7 * i] = 1.f;
  const float eps = 1.1920929e-07f * 10;
  bool singular = false;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    int k = i;
    float best = fabsf(A[i * 6 + i]);
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      const float v = fabsf(A[j * 6 + i]);
      if (v > best) { best = v; k = j; }  // "abs(A[j][i]) > abs(A[k][i])": the first maximum wins
    }
    if (best < eps) singular = true;
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      if (k == j) {
#pragma unroll
        for (int q = i; q < 6; q++) { const float t = A[i * 6 + q]; A[i * 6 + q] = A[j * 6 + q]; A[j * 6 + q] = t; }
#pragma unroll
        for (int q = 0; q < 6; q++) { const float t = X[i * 6 + q]; X[i * 6 + q] = X[j * 6 + q]; X[j * 6 + q] = t; }
      }
    }
    const float d = -1.f / A[i * 6 + i];
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      const float alpha = A[j * 6 + i] * d;
#pragma unroll
      for (int q = i + 1; q < 6; q++) A[j * 6 + q] = A[j * 6 + q] + alpha * A[i * 6 + q];
#pragma unroll
      for (int q = 0; q < 6; q++) X[j * 6 + q] = X[j * 6 + q] + alpha * X[i * 6 + q];
    }
    A[i * 6 + i] = -d;
    if (singular) break;  // the reference returns at the first tiny pivot
  }
  if (singular) {
#pragma unroll
    for (int q = 0; q < 36; q++) X[q] = 0.f;
    return false;
  }
#pragma unroll
  for (int i = 5; i >= 0; i--)
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float s = X[i * 6 + j];
#pragma unroll
      for (int q = i + 1; q < 6; q++) s = s - A[i * 6 + q] * X[q * 6 + j];
      X[i * 6 + j] = s * A[i * 6 + i];
    }
  return true;
}

// deltaMat = A.inv() * b (src/Tracker.cpp:564), OpenCV's evaluation: MatOp_Invert::matmul makes it MatOp_Solve, i.e.
// cv::solve(A, b, x, DECOMP_LU) = hal::LU32f(A, 6, b, 1): the elimination of inv6_lu applied to the 6x1 right-hand side,
// f32 back substitution, singular => x = 0.  No inverse, no product.  Returns false when singular.
__device__ inline bool solve6_lu(const float Ain[36], const float bin[6], float x[6]) {
  float A[36], b[6];
#pragma unroll
  for (int i = 0; i < 36; i++) A[i] = Ain[i];
#pragma unroll
  for (int i = 0; i < 6; i++) b[i] = bin[i];
  const float eps = 1.1920929e-07f * 10;
  bool singular = false;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    int k = i;
    float best = fabsf(A[i * 6 + i]);
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      const float v = fabsf(A[j * 6 + i]);
      if (v > best) { best = v; k = j; }
    }
    if (best < eps) { singular = true; break; }
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      if (k == j) {
#pragma unroll
        for (int q = i; q < 6; q++) { const float t = A[i * 6 + q]; A[i * 6 + q] = A[j * 6 + q]; A[j * 6 + q] = t; }
        const float t = b[i]; b[i] = b[j]; b[j] = t;
      }
    }
    const float d = -1.f / A[i * 6 + i];
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      const float alpha = A[j * 6 + i] * d;
#pragma unroll
      for (int q = i + 1; q < 6; q++) A[j * 6 + q] = A[j * 6 + q] + alpha * A[i * 6 + q];
      b[j] = b[j] + alpha * b[i];
    }
    A[i * 6 + i] = -d;
  }
  if (singular) {
#pragma unroll
    for (int q = 0; q < 6; q++) x[q] = 0.f;
    return false;
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float s = b[i];
#pragma unroll
    for (int q = i + 1; q < 6; q++) s = s - A[i * 6 + q] * b[q];
    b[i] = s * A[i * 6 + i];
  }
#pragma unroll
  for (int q = 0; q < 6; q++) x[q] = b[q];
  return true;
}

// The legacy arithmetic set's A.inv() * b: the inverse formed first (inv6_lu), then the 6x6·6x1 product accumulated in
// f64 and rounded once.  legacy == false: solve6_lu (Ainv_out, when asked for, still receives inv6_lu's inverse).
__device__ inline bool solve_delta(const float A[36], const float b[6], float delta[6], float* Ainv_out, bool legacy) {
  float Ai[36];
  bool ok = true;
  if (legacy || Ainv_out) ok = inv6_lu(A, Ai);
  if (legacy) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 6; j++) s += (double)Ai[6 * i + j] * (double)b[j];
      delta[i] = (float)s;
    }
  } else {
    ok = solve6_lu(A, b, delta);
  }
  if (Ainv_out) {
#pragma unroll
    for (int i = 0; i < 36; i++) Ainv_out[i] = Ai[i];
  }
  return ok;
}

// The same A.inv() * b on a wave.  OpenCV's set (legacy == false): the 7 columns of [A | b] over 7 lanes (lane c < 6 owns
// column c of A, lanes 6.. the right-hand side): hal::LU32f(A, 6, b, 1), delta = the back-substituted b.  Legacy set: the 12
// columns of [A | X] over 12 lanes (lane 6 + j column j of X = I), then delta = X * b accumulated in f64.  Every lane of
// the wave must be active and hold the same `sums` / `b`.  Each element goes
// through exactly the operations of inv6_lu — the elimination updates every column of a row alike — so the result is
// bit-identical; what changes is that a row operation is one instruction instead of twelve, and the pivot, the
// multipliers and the upper triangle travel through v_readlane.  Entries the serial code never reads again (below the
// diagonal of eliminated columns) are left with other junk here.
__device__ __forceinline__ float lane_f(float v, int lane) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
}

// sums: the 21 upper-triangle entries of A in row-major order (as accumulated); returns false when singular (delta = 0)
__device__ inline bool solve_delta_wave(const double* sums, const float b[6], float delta[6], bool legacy) {
  const int lane = (int)(thread_here() & 63u);
  const int c = lane < 12 ? lane : 11;  // lanes >= 12 shadow lane 11
  float col[6];
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int i = r < c ? r : c, j = r < c ? c : r;          // (min, max) of (r, c) for the A lanes
    const int idx = i * 6 - (i * (i - 1)) / 2 + (j - i);
    const float a_rc = (float)sums[c < 6 ? idx : 0];
    const float rhs = legacy ? (r == c - 6 ? 1.f : 0.f) : b[r];   // X = I (legacy) or the right-hand side itself
    col[r] = c < 6 ? a_rc : rhs;
  }
  const float eps = 1.1920929e-07f * 10;
  bool singular = false;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    int k = i;
    float best = fabsf(col[i]);
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      const float v = fabsf(col[j]);
      if (v > best) { best = v; k = j; }  // "abs(A[j][i]) > abs(A[k][i])": the first maximum wins
    }
    k = __builtin_amdgcn_readlane(k, i);          // lane i owns column i
    best = lane_f(best, i);
    if (best < eps) { singular = true; break; }   // wave-uniform
#pragma unroll
    for (int j = i + 1; j < 6; j++)
      if (k == j) { const float t = col[i]; col[i] = col[j]; col[j] = t; }   // wave-uniform branch
    const float d = lane_f(-1.f / col[i], i);
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      const float alpha = lane_f(col[j] * d, i);
      col[j] = col[j] + alpha * col[i];
    }
    if (lane == i) col[i] = -d;
  }
  if (singular) {
#pragma unroll
    for (int i = 0; i < 6; i++) delta[i] = 0.f;   // cv::Mat::inv() returns zeros, zeros * b = 0
    return false;
  }
  // the upper triangle and the (inverted) diagonal, out of the A lanes before the X lanes' columns are overwritten
  float U[6][6];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int q = i; q < 6; q++) U[i][q] = lane_f(col[i], q);
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float sacc = col[i];
#pragma unroll
    for (int q = i + 1; q < 6; q++) sacc = sacc - U[i][q] * col[q];
    col[i] = sacc * U[i][i];
  }
  if (!legacy) {   // wave-uniform: lane 6 holds x = the back-substituted right-hand side (cv::solve's dst)
#pragma unroll
    for (int i = 0; i < 6; i++) delta[i] = lane_f(col[i], 6);
    return true;
  }
  // delta = X * b, f64 accumulation in column order, rounded once (X[i][j] lives in lane 6 + j)
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 6; j++) acc += (double)lane_f(col[i], 6 + j) * (double)b[j];
    delta[i] = (float)acc;
  }
  return true;
}

}  // namespace uwt
