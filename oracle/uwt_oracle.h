/*
 * uwt_oracle.h — CPU ORACLE for the UW-SLAM direct SE(3) tracking hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (uw-slam_amd/csrc,
 * libuwt_hip.so) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (MecatronicaUSB/uw-slam) ships no tests, golden vectors
 * or fixtures for this path and cannot be compiled in this image (needs OpenCV 3.2 + contrib
 * + CUDA, Eigen3, Ceres, ROS — all absent).  This file is a plain-C restatement
 * of the reference algorithm; every function cites the reference file:line it follows.
 * Third-party arithmetic that is not under /root/reference (OpenCV 3.2 resize / Scharr /
 * gemm / Mat::inv, Eigen3 quaternion kernels) is restated from the published algorithms;
 * the choices are listed below (G1..G4, S1..S8) and in DESIGN.md.
 *
 * Third-party arithmetic, two selectable sets (uwo_params::arith).  All IEEE-754 binary32 unless stated.
 *
 * UWO_ARITH_OPENCV (0, the default): OpenCV 3.x's published generic (non-BLAS, non-IPP) code paths, restated from
 * modules/core/src/matop.cpp (MatExpr algebra), matmul.cpp (gemmImpl / GEMMSingleMul / GEMMBlockMul / GEMMStore),
 * lapack.cpp + hal (cv::solve / LUImpl), convert.cpp (cvtScale32f), arithm.cpp (multiply / divide / add with a scalar):
 *  G1  every cv::gemm on CV_32F runs GEMMSingleMul<float,double> / GEMMBlockMul<float,double>: operands widened to
 *      double, products exact, partial sums in double, ONE rounding to float on store:
 *      - rigid * points.t() (Tracker.cpp:1450; GEMM_2_T, len 4): s0..s3 take one product each, stored value
 *        (float)(T0*X + ((T1*Y + T2*Z) + T3*w)) — the source folds the partial sums with "s0 += s1 + s2 + s3;", whose
 *        right-hand side is evaluated first (uwo_set_gemm_fold(1) selects the left-to-right fold ((s0+s1)+s2)+s3 instead);
 *      - Jl * Jw (:479; flags 0, len 2, 1x6 result — the inline len-2 case needs len == d_size.width|height, so the
 *        generic branch runs): (float)((0 + g0*Jw0k) + g1*Jw1k);
 *      - Jacobians.t() * Jacobians (:560; GEMM_1_T, 6x6, len N): one sequential double sum per entry (also through
 *        the block algorithm for N > 10000, whose d_buf carries the sum from block to block);
 *      - 1-wide results (-Jacobians.t() * Residuals.mul(W) :561, inv_n * Residuals.t() * ResidualsW :501): gemmImpl
 *        turns them into A*Bt with b_step 0; N <= 10000: four interleaved partial sums folded the same way (CV_ENABLE_UNROLLED);
 *        N > 10000: blocks of dk0 = min(16384/rows, N) terms, two interleaved partial sums per block, the total carried
 *        in double across blocks; alpha (-1, inv_n) applied in double before the store.
 *  G2  A.inv() * b (:564) is MatOp_Invert::matmul -> MatOp_Solve -> cv::solve(A, b, x, DECOMP_LU) -> hal::LU32f(A, 6, b, 1):
 *      f32 Gaussian elimination with partial pivoting applied to the 6x1 right-hand side, back substitution in f32
 *      (pivot slot holds 1/pivot), |pivot| < 10*FLT_EPSILON => x = 0.  No inverse is formed, no product follows.
 *  G3  ((col - cx) * invfx) (:1439, :1443) is MatOp_AddEx::multiply folding the scalar into alpha and s (in double),
 *      assigned through convertTo(alpha, beta) -> cvtScale32f: x*(float)alpha + (float)beta, beta = (float)(-(double)cx *
 *      (double)invfx): one f32 multiply and one f32 add.
 *  G4  row *= fx, row /= row2, row += cx, .mul(): separate f32 operations (convertTo scale / cv::divide / cv::add /
 *      cv::multiply); cv::divide yields 0 for a zero divisor (OpenCV 3.x), such a point fails "z2 != 0" (:451) either way.
 *
 * UWO_ARITH_LEGACY (1): the set rounds 1-3 pinned (kept selectable; it is cheaper on the GPU and NOT what OpenCV computes):
 *  S1  small products (4x4·4xN warp, 1x2·2x6 Jacobian row): k-sequential f32 FMA chain s = a0*b0; s = fma(a1,b1,s); ...
 *  S3  cv::Mat::inv() as its own step: f32 LU on [A | I] (same elimination as G2), then
 *  S4  the 6x6·6x1 product with f64 accumulation, rounded to f32;
 *      unprojection as written, (x - cx) * invfx; N-long sums sequential in double (the 1-wide ones too).
 *
 * Common to both sets:
 *  S2  N-long reductions accumulate exact f32 products in double and round to f32 once; Σr² is an exact integer
 *      (residuals are integer differences of u8).
 *  S5  sinf/cosf := (float)sin/cos((double)x) — the correctly rounded value, a function of x alone; the reference calls libm's
 *      sinf / cosf, whose last bit is the libm build's (glibc 2.35: 1 ulp off at 0.063 % / 0.0015 % of the floats of [0, 0.5],
 *      tools/trig/trig_sweep.c); uwo_params::trig = UWO_TRIG_LIBM evaluates this host's instead (diagnosis).  sqrtf, 1/x, a/b
 *      correctly rounded (sqrtf == (float)sqrt((double)x) on every float: the same sweep).
 *  S6  Sophus / Eigen quaternion formulas: generic (non-SIMD) left-to-right f32, no FMA.
 *  S7  nearest-neighbour index round(x2) may equal the dimension (reference reads out of
 *      bounds, Tracker.cpp:450,472); the oracle clamps the index to dim-1.
 *  S8  error = (float)((double)(float)(1.0/N) * (double)Σ w r²)  (scaled-gemm form of Tracker.cpp:499-502).
 *
 * Build with -ffp-contract=off (oracle/Makefile does).
 */
#ifndef UWT_ORACLE_H
#define UWT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UWO_MAX_LEVELS 8

/* status codes (mirrors include/uwt.h) */
enum {
  UWO_OK = 0,
  UWO_ERR_INVALID_ARG = 1,
  UWO_ERR_NO_VALID_POINTS = 2
};

enum { UWO_WEIGHTS_IDENTITY = 0, UWO_WEIGHTS_TUKEY_REFERENCE = 1, UWO_WEIGHTS_HUBER = 2 };
enum { UWO_SAMPLER_NEAREST = 0, UWO_SAMPLER_BILINEAR = 1 };
enum { UWO_ARITH_OPENCV = 0, UWO_ARITH_LEGACY = 1 };
enum { UWO_TRIG_ROUNDED = 0, UWO_TRIG_LIBM = 1 };   /* S5: (float)sin((double)x) (default) / this host's sinf, cosf */

typedef struct uwo_params {
  int32_t width, height;     /* level-0 size */
  float fx, fy, cx, cy;      /* level-0 intrinsics (K) */
  int32_t n_levels;          /* PYRAMID_LEVELS (Options.cpp:26) */
  int32_t first_level;       /* coarsest level iterated (Tracker.cpp:368) */
  int32_t last_level;        /* finest level iterated   (Tracker.cpp:369) */
  int32_t max_iters;         /* Tracker.cpp:366 */
  float epsilon;             /* Tracker.cpp:364 */
  float gain;                /* Tracker.cpp:559 */
  float z_factor;            /* Tracker.cpp:371 */
  float angle_factor;        /* Tracker.cpp:372 */
  float depth_scale;         /* Tracker.cpp:1261 */
  float initial_error;       /* Tracker.cpp:393 */
  int32_t early_exit;        /* 1: reference termination rule; 0: exactly max_iters updates/level */
  int32_t has_depth;
  int32_t handoff_scale_t;   /* 0: EstimatePose (:580-590); 1: EstimatePoseFeatures (:856) */
  int32_t weights;           /* UWO_WEIGHTS_* */
  int32_t sampler;           /* UWO_SAMPLER_* (bilinear is a north-star extension, not in the reference) */
  int32_t arith;             /* UWO_ARITH_OPENCV (0, default: G1..G4 above) or UWO_ARITH_LEGACY (1: S1, S3, S4) */
  int32_t gemm_fold;         /* 0 (default): GEMMSingleMul's partial sums folded s0 + ((s1 + s2) + s3); 1: ((s0 + s1) + s2) + s3 */
  int32_t trig;              /* UWO_TRIG_ROUNDED (0, default) or UWO_TRIG_LIBM (1: the host libm's sinf / cosf; diagnosis, S5) */
} uwo_params;

typedef struct uwo_level {
  int32_t w, h;              /* w_[lvl], h_[lvl] = width >> lvl, height >> lvl (Tracker.cpp:312-313): the point grid of ObtainAllPoints */
  float fx, fy, cx, cy, invfx, invfy;
  int32_t iw, ih;            /* images_[lvl].cols / rows: the cv::resize(.., 0.5, 0.5) chain (System.cpp:246-251), cvRound per level;
                                equal to w, h when the level-0 size is divisible by 2^lvl, larger otherwise */
} uwo_level;

/* one row per GN iteration (fixtures / parity traces) */
typedef struct uwo_trace {
  int32_t level, iter, n_valid, exited;   /* exited: 1 if the exit test fired at this iteration */
  int64_t sum_r2;                         /* Σ r² (exact; identity weights) */
  float error;
  float A[36];
  float b[6];
  float delta[6];
  float pose[7];                          /* qx qy qz qw tx ty tz after this iteration */
} uwo_trace;

void uwo_default_params(uwo_params* p, int width, int height, float fx, float fy, float cx, float cy);

/* Tracker::InitializePyramid, Tracker.cpp:297-340 */
int uwo_level_intrinsics(const uwo_params* p, int lvl, uwo_level* out);

/* System::AddFrame pyramid loop, System.cpp:246-251 (cv::resize ½ == 2x2 mean, round half up) */
void uwo_halve_u8(const uint8_t* src, int w, int h, uint8_t* dst);
void uwo_halve_u16(const uint16_t* src, int w, int h, uint16_t* dst);
/* the same call on ANY source size: dst is uwo_half_size(sw) x uwo_half_size(sh) (cvRound: half to even), whole cells as
 * above, the partial last column / row by resizeAreaFast's generic tail (mean of the existing pixels, cvRound) */
int  uwo_half_size(int n);
void uwo_resize_half_u8(const uint8_t* src, int sw, int sh, uint8_t* dst);
void uwo_resize_half_u16(const uint16_t* src, int sw, int sh, uint16_t* dst);

/* Tracker::ApplyGradient, Tracker.cpp:1133-1134: cv::Scharr(.., CV_16S, dx, dy, scale=3, delta=0, BORDER_REFLECT_101) */
void uwo_scharr3(const uint8_t* src, int w, int h, int16_t* gx, int16_t* gy);
/* gradient_ magnitude image, Tracker.cpp:1136-1142 (convertScaleAbs + addWeighted 0.5/0.5), u8 */
void uwo_gradient_mag(const int16_t* gx, const int16_t* gy, int n, uint8_t* out);

/* Tracker::ObtainAllPoints, Tracker.cpp:1259-1310: dense N x 4 table [x y z w] */
void uwo_dense_points(const uint16_t* depth_or_null, int w, int h, int lvl, float depth_scale, float* pts);
/* the w x h grid over a depth image with rows of `stride` elements (the level's image is wider than its grid) */
void uwo_dense_points_ex(const uint16_t* depth_or_null, int stride, int w, int h, int lvl, float depth_scale, float* pts);

/* SE3 helpers; pose = qx qy qz qw tx ty tz */
void uwo_se3_identity(float pose[7]);
void uwo_se3_exp(const float xi[6], float pose[7]);                 /* sophus/se3.hpp:723-744 */
void uwo_se3_mul(const float a[7], const float b[7], float out[7]); /* se3.hpp:317-321, so3.hpp:338-354 */
void uwo_se3_matrix(const float pose[7], float T[16]);              /* se3.hpp:253-268 (row-major 4x4) */
int  uwo_se3_handoff(float pose[7], int scale_t);                   /* Tracker.cpp:580-590 */

/* Tracker::WarpFunction, Tracker.cpp:1417-1471 */
void uwo_warp(const float* pts, int n, const float pose[7], const uwo_level* L, float* warped);
/* Arithmetic set of the per-stage functions that take no parameter block (uwo_warp, uwo_residual_jacobian*,
 * uwo_normal_equations, uwo_error, uwo_solve_delta) FOR THE CALLING THREAD (thread-local; nothing process-wide).
 * uwo_estimate_pose* / uwo_align_pair* do not read it: they take the set from uwo_params::arith and hand it down with the call,
 * so alignments of different sets may run side by side in a thread pool.  Returns the thread's previous one. */
int uwo_set_arith(int arith);
/* fold of GEMMSingleMul's four partial sums for the calling thread: 0 (default) s0 + ((s1 + s2) + s3), 1 ((s0 + s1) + s2) + s3;
 * returns the previous.  uwo_estimate_pose* use fold 1 when uwo_params::gemm_fold or the calling thread's default says so. */
int uwo_set_gemm_fold(int fold);
/* sine / cosine of the SE(3) exponential for the calling thread: UWO_TRIG_ROUNDED (default) or UWO_TRIG_LIBM; returns the previous */
int uwo_set_trig(int trig);

/* per-point loop of Tracker::EstimatePose, Tracker.cpp:432-490.
 * J (n x 6) and r (n) receive the valid rows compacted; idx (n, optional) their point index. */
int uwo_residual_jacobian(const uint8_t* img1, const uint8_t* img2, const int16_t* gx1, const int16_t* gy1,
                          const float* pts, const float* warped, int n, const uwo_level* L,
                          float z_factor, float angle_factor, float* J, float* r, int32_t* idx);
/* same with a sampler choice (UWO_SAMPLER_*) */
int uwo_residual_jacobian_ex(const uint8_t* img1, const uint8_t* img2, const int16_t* gx1, const int16_t* gy1,
                             const float* pts, const float* warped, int n, const uwo_level* L,
                             float z_factor, float angle_factor, int sampler, float* J, float* r, int32_t* idx);
/* EXTENSION (not in the reference): bilinear sample of a u8 image at (x, y), 0 < x < w, 0 < y < h; neighbours clamped */
float uwo_bilinear_u8(const uint8_t* img, int w, int h, float x, float y);
/* EXTENSION: Huber weights, k = 1.345, scale = 1.4826 * median|q - median q| over q = lrint(r) (signed 511-bin histograms,
 * the reference's "first bin whose cumulative count exceeds n/2" rule) */
void  uwo_huber_weights(const float* r, int n, float* w);

/* weights, Tracker.cpp:1571-1654 */
float uwo_median_mat(const float* v, int n);
float uwo_mad(const float* v, int n);
void  uwo_tukey_weights(const float* r, int n, float* w);

/* error + normal equations + solve, Tracker.cpp:495-564 */
float uwo_error(const float* r, const float* w, int n, int64_t* sum_r2_out);
void  uwo_normal_equations(const float* J, const float* r, const float* w, int n, float gain, float A[36], float b[6]);
int   uwo_inv6(const float A[36], float Ainv[36]);  /* returns 0 when singular (Ainv zeroed) */
int   uwo_solve6(const float A[36], const float b[6], float x[6]);  /* cv::solve(A, b, x, DECOMP_LU); 0 when singular (x zeroed) */
void  uwo_solve_delta(const float A[36], const float b[6], float delta[6]);  /* "A.inv() * b" under the current set */

/* one frame's pyramid data as the tracker consumes it */
typedef struct uwo_frame {
  const uint8_t* img[UWO_MAX_LEVELS];
  const uint16_t* depth[UWO_MAX_LEVELS]; /* NULL if !has_depth */
  const int16_t* gx[UWO_MAX_LEVELS];
  const int16_t* gy[UWO_MAX_LEVELS];
} uwo_frame;

/* Tracker::EstimatePose, Tracker.cpp:362-597.  trace may be NULL; *n_trace in: capacity, out: rows. */
int uwo_estimate_pose(const uwo_params* p, const uwo_frame* prev, const uwo_frame* cur,
                      float pose_out[7], uwo_trace* trace, int32_t* n_trace);

/* the same loop over explicit per-level point tables (candidatePoints_[lvl]); tables == NULL: dense */
int uwo_estimate_pose_points(const uwo_params* p, const uwo_frame* prev, const uwo_frame* cur,
                             const float* const* tables, const int32_t* n_points,
                             float pose_out[7], uwo_trace* trace, int32_t* n_trace);
/* Tracker::ObtainPatchesPoints (Tracker.cpp:1178-1257) and ObtainCandidatePoints (:1314-1398, one level) */
int uwo_patch_points(const float* kp, int n_kp, const uint16_t* depth0_or_null, int w, int h, float* pts, int cap);
int uwo_add_patch_points(const float* pts_in, int n, int w, int h, int patch_size, float* pts, int cap);  /* Tracker.cpp:599-629 */
int uwo_candidate_points(const uint8_t* mag, const uint16_t* depth_or_null, int w, int h, double threshold, float* pts, int cap);
int uwo_candidate_points_ex(const uint8_t* mag, const uint16_t* depth_or_null, int iw, int ih, int w, int h, double threshold,
                            float* pts, int cap);

/* convenience: level-0 images in, pyramid + gradients + EstimatePose; (the CPU-baseline unit of work) */
int uwo_align_pair(const uwo_params* p, const uint8_t* ref_gray, const uint8_t* tgt_gray,
                   const uint16_t* ref_depth, const uint16_t* tgt_depth,
                   float pose_out[7], uwo_trace* trace, int32_t* n_trace);
/* same with explicit per-level point tables for the reference frame (NULL: dense) */
int uwo_align_pair_points(const uwo_params* p, const uint8_t* ref_gray, const uint8_t* tgt_gray,
                          const uint16_t* ref_depth, const float* const* tables, const int32_t* n_points,
                          float pose_out[7], uwo_trace* trace, int32_t* n_trace);

/* Visualizer::UpdateMessages pose accumulation, Visualizer.cpp:304-325: final = previous * SE3(q, t_scale * t);
 * reference_axes != 0 also applies the published position permutation (-z, -x, -y).  traj_out: n x 7.
 * The reference uses t_scale = 40 and the permutation; (1, 0) is the plain SE(3) prefix product. */
void uwo_accumulate_trajectory(const float* poses, int n, const float start[7], float t_scale, int reference_axes,
                               float* traj_out);

/* ---- frame ingest: CameraModel::GetCameraModel (CameraModel.cpp:84-90), System::CalculateROI
 * (System.cpp:148-191), System::AddFrame remap + crop (System.cpp:231-235).  OpenCV 3.2 calib3d / imgproc algorithms
 * restated: getOptimalNewCameraMatrix(alpha = 1), initUndistortRectifyMap(CV_16SC2), remap(INTER_LINEAR, border 0). */
void uwo_optimal_new_camera_matrix(const float K[4], const float dist[4], int in_w, int in_h, double alpha, int new_w,
                                   int new_h, double newK[4]);
void uwo_init_undistort_maps(const float K[4], const float dist[4], const double newK[4], int w, int h, int16_t* map1,
                             uint16_t* map2);
void uwo_remap_linear(const uint8_t* src, int sw, int sh, const int16_t* map1, const uint16_t* map2, int dw, int dh,
                      uint8_t* dst);
void uwo_calculate_roi(const uint8_t* undistorted, int w, int h, int32_t roi[4]);

/* LS, LeastSquares.cpp:30-209 */
typedef struct uwo_ls {
  float A[36];
  float b[6];
  float error;
  int32_t num_constraints;
  float sse[4 * 28];
} uwo_ls;
void uwo_ls_initialize(uwo_ls* ls);
void uwo_ls_update(uwo_ls* ls, const float J[6], float res, float weight);
void uwo_ls_update4(uwo_ls* ls, const float J[6][4], const float res[4], const float weight[4], int quirk_plus6);
void uwo_ls_finish_no_divide(uwo_ls* ls);
void uwo_ls_finish(uwo_ls* ls);

#ifdef __cplusplus
}
#endif
#endif
