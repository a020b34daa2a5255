/*
 * uwt.h — C ABI of the MI355X-native direct SE(3) tracker (libuwt_hip.so).
 *
 * Drop-in boundary for UW-SLAM's per-frame direct-tracking hot path.  Each entry point names the reference
 * interface it replaces (paths relative to the reference repo).  Plain pointers and sizes only; every call
 * returns an int status (UWT_OK == 0).  The caller owns all host buffers; a uwt_ctx owns its device buffers
 * and its HIP stream.  One ctx per host thread per GPU.  Calls are synchronous unless named *_async.
 *
 * The library is HIP-only: there is no CPU fallback.  uwt_create() fails with UWT_ERR_NO_DEVICE when no
 * gfx950 device is visible.
 */
#ifndef UWT_H
#define UWT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UWT_MAX_LEVELS 8
#define UWT_ABI_VERSION 4   /* 2: uwt_params::arith; 3: uwt_tuning (no environment variables are read any more); 4: any frame size
                               (uwt_level::img_w / img_h / pitch, uwt_resize_half_*) */

enum uwt_status_code {
  UWT_OK = 0,
  UWT_ERR_INVALID_ARG = 1,
  UWT_ERR_NO_VALID_POINTS = 2, /* reference: cv::Exception from the empty Mat product, src/Tracker.cpp:501 */
  UWT_ERR_HIP = 3,
  UWT_ERR_NO_DEVICE = 4,
  UWT_ERR_CAPACITY = 5,
  UWT_ERR_PAIR_FAILED = 6      /* batch ran; at least one pair has a non-zero status in its uwt_stats */
};

enum uwt_plane { UWT_PLANE_IMAGE = 0, UWT_PLANE_DEPTH = 1, UWT_PLANE_GRADX = 2, UWT_PLANE_GRADY = 3 };

/* Arithmetic of the OpenCV steps on the path (the reference pins no OpenCV build; "3.2", README.md:15).
 * UWT_ARITH_OPENCV: what OpenCV 3.x's generic (non-BLAS, non-IPP) code computes for the reference's expressions —
 *   - every cv::gemm on CV_32F accumulates in double and rounds to float once (matmul.cpp GEMMSingleMul<float,double>):
 *     rigid * points.t() (src/Tracker.cpp:1450) = (float)(T0*X + ((T1*Y + T2*Z) + T3*w)) — the A*Bt branch folds its four
 *     partial sums with "s0 += s1 + s2 + s3;" —, Jl * Jw (:479) = (float)((0 + g0*Jw0k) + g1*Jw1k), and the N-long JtJ / Jtr /
 *     rtr sums (:501, :560-561);
 *   - "(col - cx) * invfx" (:1439, :1443) is folded by the MatExpr algebra (matop.cpp MatOp_AddEx::multiply) into one scaled
 *     convert: x * invfx + (float)(-(double)cx * invfx);
 *   - "A.inv() * b" (:564) is MatOp_Invert::matmul -> cv::solve(A, b, DECOMP_LU) = hal::LU32f(A, 6, b, 1): elimination on
 *     the right-hand side and f32 back substitution, no inverse and no product.
 * UWT_ARITH_LEGACY: rounds 1-3 of this library: the 4-/2-term products as f32 FMA chains, (x - cx) * invfx as written, the
 *   inverse formed and multiplied with f64 accumulation.  Cheaper; differs from the above by <= 1 ulp per pixel term. */
enum uwt_arith { UWT_ARITH_OPENCV = 0, UWT_ARITH_LEGACY = 1 };

/* All solver constants the reference hard-codes as locals of Tracker::EstimatePose* (src/Tracker.cpp:364-372,
 * 634-640) and as link-time globals (src/Options.cpp:26-28), as one POD. */
typedef struct uwt_params {
  int32_t width, height;     /* level-0 size (w_, h_ of src/System.cpp:123: the ROI size of :186-190 when the camera is
                                distorted — any size, as long as (size >> (n_levels-1)) >= 1)                             */
  float fx, fy, cx, cy;      /* level-0 intrinsics = K passed to Tracker::InitializePyramid              */
  int32_t n_levels;          /* PYRAMID_LEVELS, src/Options.cpp:26 (5)                                   */
  int32_t first_level;       /* coarsest level iterated, src/Tracker.cpp:368 (4)                         */
  int32_t last_level;        /* finest level iterated,   src/Tracker.cpp:369 (1)                         */
  int32_t max_iters;         /* src/Tracker.cpp:366 (50)                                                 */
  float epsilon;             /* src/Tracker.cpp:364 (1e-3)                                               */
  float gain;                /* residual gain, src/Tracker.cpp:559 (50)                                  */
  float z_factor;            /* src/Tracker.cpp:371 (1)                                                  */
  float angle_factor;        /* src/Tracker.cpp:372 (1)                                                  */
  float depth_scale;         /* src/Tracker.cpp:1261 (0.0002)                                            */
  float initial_error;       /* last_error seed, src/Tracker.cpp:393 (50000)                             */
  int32_t early_exit;        /* 1: reference exit test src/Tracker.cpp:508; 0: exactly max_iters updates */
  int32_t has_depth;         /* Tracker(bool _depth_available)                                           */
  int32_t handoff_scale_t;   /* 0: EstimatePose (:580-590); 1: EstimatePoseFeatures (:856)               */
  int32_t accumulate_f64;    /* 1 (default): JᵀJ/Jᵀr summed in f64 like cv::gemm on CV_32F (:560-561);  */
                             /* 0: f32 per-thread partial sums (faster, not bit-reproducing the oracle)  */
  int32_t sampler;           /* 0: nearest neighbour, round() (:472, reference); 1: bilinear (north-star extension)  */
  int32_t weights;           /* 0: IdentityWeights (:495, :1621); 1: TukeyFunctionWeights with the reference's      */
                             /* histogram medians (:496, :1571-1654); 2: Huber, k = 1.345 (extension)                */
  int32_t max_frames;        /* frame-slot capacity of the context                                       */
  int32_t max_pairs;         /* largest batch of pairs per call                                          */
  int32_t device;            /* HIP device ordinal                                                       */
  int32_t arith;             /* uwt_arith: UWT_ARITH_OPENCV (default) or UWT_ARITH_LEGACY                */
} uwt_params;

/* per-level camera model = the vectors Tracker::InitializePyramid fills (include/Tracker.h:516-528), and the size of the
 * level's images.  The two sizes differ where the level-0 size is not divisible by 2^lvl (the reference's EUROC path crops to a
 * data-dependent ROI, src/System.cpp:148-191):
 *   w, h          w_[lvl], h_[lvl] = size >> lvl (src/Tracker.cpp:312-313): the point grid ObtainAllPoints walks (:1267-1268);
 *   img_w, img_h  images_[lvl].cols / rows: the chain of cv::resize(.., Size(), 0.5, 0.5) (src/System.cpp:246-251), each step
 *                 cvRound(size * 0.5) — half to even: 733 -> 366, 735 -> 368; what the bounds test of src/Tracker.cpp:450 reads
 *                 and what uwt_get_plane returns; img_w >= w, img_h >= h;
 *   pitch         elements per row of the level's planes in device memory (img_w rounded up to a multiple of 4): what a producer
 *                 that writes frames in place through uwt_plane_device_ptr must honour — a slot is pitch * img_h elements.  Equal to
 *                 the width for every width that is a multiple of 4. */
typedef struct uwt_level {
  int32_t w, h;
  float fx, fy, cx, cy, invfx, invfy;
  int32_t img_w, img_h;
  int32_t pitch;
} uwt_level;

typedef struct uwt_stats {
  int32_t status;     /* per-pair uwt_status_code */
  int32_t iterations; /* residual evaluations over all levels */
  int32_t n_valid;    /* valid points of the last evaluation */
  float error;        /* error of the last evaluation (src/Tracker.cpp:499-502) */
} uwt_stats;

/* normal-equation accumulators of one residual evaluation (the 28 LS accumulators, src/LeastSquares.cpp:151-199,
 * plus the constraint count): A upper triangle row-major (A00 A01 .. A05 A11 .. A55), jtr = +Σ J·r (un-gained). */
typedef struct uwt_accum {
  double A[21];
  double jtr[6];
  int64_t sum_r2;
  int32_t n_valid;
  int32_t pad;
} uwt_accum;

typedef struct uwt_ctx uwt_ctx;

/* Launch-shape switches of a context: HOW the same arithmetic is laid out in launches, never WHAT is computed — every
 * setting gives the same poses bit for bit (tests/test_gpu_production.py runs the forms against each other).  The library reads
 * no environment variable; a context starts with the defaults below (uwt_get_tuning returns them), and the parity suite and the
 * A/B tools change them through uwt_set_tuning.  No reference counterpart (the reference has one form of everything). */
typedef struct uwt_tuning {
  int32_t split;             /* parts a large fixed-schedule batch is cut into, each on a stream of its own (1..4) [2]      */
  int32_t split_min;         /* pairs per part at least [8]                                                                 */
  int64_t split_min_px;      /* level-0 pixels of the whole batch from which the split pays [32 * 640 * 480]                */
  int64_t stream_bytes;      /* a level whose planes over the whole batch exceed this is read non-temporally [200 MiB]      */
  int32_t tail_update;       /* Gauss-Newton update in the tail of the evaluation's own launch: 0 never, 1 split batches,
                                2 always [1]                                                                                */
  int32_t target_blocks;     /* blocks per residual launch the batch-dependent slicing aims at; 0: automatic [0]            */
  int32_t coarse;            /* a few pairs: the coarsest levels in one launch (k_coarse) [1]                               */
  int32_t coarse_batch_px;   /* batches: levels of up to this many pixels run one block per pair, one launch per level;
                                0: never [6144]                                                                             */
  int32_t coarse_weighted;   /* the same for robust weights over the nearest sampler (k_coarse_weighted) [1]                */
  int32_t overlap_gradients; /* uwt_track_batch_async: finer levels' gradients on a side stream beside the first, coarse
                                iterations [1]                                                                              */
  int32_t first_poll;        /* early-exit schedules: evaluations of a level before the host first looks [3]                */
  int32_t chained;           /* -1: update chained into the next evaluation's launch for a few pairs; 1 / 0: always / never
                                [-1]                                                                                        */
  int32_t speculation;       /* one or two pairs, early exit: launch without read-backs, run again if cut short [1]         */
  int32_t fused_stages;      /* a few frames: whole pyramid / all gradient levels in one launch each [1]                    */
  int32_t pyramid_batch;     /* batches: pyramid levels 1..3 in one pass over level 0 [1]                                   */
  int32_t typed_loads;       /* the dominant kernel reads gradients and depth through typed buffer loads (the texture path
                                converts int16 to float: three vector conversions per pixel less) [1]                       */
  int32_t reserved[4];
} uwt_tuning;

/* ---- lifecycle -------------------------------------------------------------------------------------------- */

/* Fills the reference's EstimatePose constants (src/Tracker.cpp:364-372) for a w x h camera. */
int uwt_default_params(uwt_params* p, int32_t width, int32_t height, float fx, float fy, float cx, float cy);

/* new Tracker(depth) + Tracker::InitializePyramid(w, h, K)  (src/System.cpp:121-122, src/Tracker.cpp:272, 297-340) */
int uwt_create(const uwt_params* p, uwt_ctx** out);
/* Changes the solver constants of a live context — the locals the reference re-declares at the top of each
 * EstimatePose* variant (src/Tracker.cpp:364-372, 634-640, 877-885): first/last level, max_iters, epsilon, gain, z_factor,
 * angle_factor, initial_error, early_exit, handoff_scale_t, accumulate_f64, sampler, weights, arith.  Geometry and capacity
 * (size, intrinsics, n_levels, has_depth, max_frames, max_pairs, device) must equal the context's. */
int uwt_update_params(uwt_ctx* ctx, const uwt_params* p);
/* the context's current parameters */
int uwt_get_params(const uwt_ctx* ctx, uwt_params* out);
/* the context's launch-shape switches (uwt_tuning); uwt_set_tuning waits for the context's work in flight and applies to every
 * later call.  A value outside its range (split 1..4, split_min >= 1, split_min_px >= 1, stream_bytes >= 0, tail_update 0..2,
 * target_blocks 0..2^20, coarse_batch_px 0..2^24, first_poll 1..2^20, chained -1..1) returns UWT_ERR_INVALID_ARG and changes
 * nothing; the on / off switches take any non-zero value as 1. */
int uwt_get_tuning(const uwt_ctx* ctx, uwt_tuning* out);
int uwt_set_tuning(uwt_ctx* ctx, const uwt_tuning* t);
/* Tracker::~Tracker (src/Tracker.cpp:280-293) */
int uwt_destroy(uwt_ctx* ctx);
/* reads back w_/h_/fx_/fy_/cx_/cy_/invfx_/invfy_[lvl] (include/Tracker.h:516-526) */
int uwt_level_info(const uwt_ctx* ctx, int32_t lvl, uwt_level* out);
const char* uwt_status_string(int status);
const char* uwt_last_error(const uwt_ctx* ctx);
int uwt_abi_version(void);
/* sha256 (hex) of the sources and compiler flags this library was built from (csrc/Makefile puts it in at build time; hipcc's
 * output is not byte-reproducible, the sources are).  No reference counterpart: measurement hygiene — bench.py quotes the
 * counter-derived facts under profiles/ only while the loaded library reports the id they were collected on. */
const char* uwt_source_id(void);

/* ---- frames (the Frame data the tracker borrows: images_, depths_, gradientX_, gradientY_; include/System.h:85-89) */

/* Frame::images_[0] / depths_[0] of one frame from host memory with row strides in BYTES (cv::Mat::step).
 * Replaces the imread result handed to the pyramid loop in System::AddFrame (src/System.cpp:228, 243).  A strided image (a view
 * into a wider one: images_[0] = distortion(ROI), src/System.cpp:235) crosses as ONE copy of the span its rows cover — first byte
 * of the first row to last byte of the last, the bytes between the rows included: they must be readable, as they are inside a
 * parent image — while the stride is at most four times the row; beyond that as a 2-D copy (slow: issued row by row). */
int uwt_set_frame(uwt_ctx* ctx, int32_t slot, const uint8_t* gray, size_t row_stride,
                  const uint16_t* depth_or_null, size_t depth_row_stride);
/* n tightly packed frames (w*h elements each) into slots first_slot .. first_slot+n-1 */
int uwt_upload_frames(uwt_ctx* ctx, int32_t first_slot, int32_t n, const uint8_t* gray, const uint16_t* depth_or_null);
/* The same copy, asynchronous, on the context's copy stream: the per-frame ingest of System::AddFrame
 * (src/System.cpp:225-251) overlapped with the tracking of the frames already on the device.  `gray` / `depth` must be
 * page-locked (uwt_host_alloc) and stay untouched until a later uwt_sync, or until the tracker call that consumes these
 * slots has been followed by a uwt_sync.  The context orders the copy behind the tracker work still in flight on the
 * same slots and every later tracker call on these slots behind the copy; nothing else waits.  depth may be null even
 * on a context with a depth plane (the tracker reads depth of reference frames only, src/Tracker.cpp:1266-1272). */
int uwt_upload_frames_async(uwt_ctx* ctx, int32_t first_slot, int32_t n, const uint8_t* gray, const uint16_t* depth_or_null);
/* page-locked host memory for uwt_upload_frames_async (hipHostMalloc / hipHostFree) */
int uwt_host_alloc(size_t bytes, void** out);
int uwt_host_free(void* p);
/* device pointer of a plane of a slot (so a producer can write level-0 frames in place, inputs resident in HBM): img_h rows of
 * uwt_level::pitch elements, the first img_w of each row the image (tight rows whenever the width is a multiple of 4) */
int uwt_plane_device_ptr(uwt_ctx* ctx, int32_t slot, int32_t lvl, int32_t plane, void** out);
/* copy one plane of one slot back to the host: img_h x img_w elements, tight rows */
int uwt_get_plane(uwt_ctx* ctx, int32_t slot, int32_t lvl, int32_t plane, void* host_out);

/* the resize loop of System::AddFrame for levels 1..n_levels-1 (src/System.cpp:246-251), n frames at once */
int uwt_build_pyramids(uwt_ctx* ctx, int32_t first_slot, int32_t n);
/* Tracker::ApplyGradient(Frame*) (src/Tracker.cpp:1127-1134) for n frames at once */
int uwt_apply_gradient(uwt_ctx* ctx, int32_t first_slot, int32_t n);

/* ---- tracking --------------------------------------------------------------------------------------------- */

/* Tracker::EstimatePose(previous, current) (src/Tracker.cpp:362-597) for n_pairs independent pairs.
 * poses_out: n_pairs x 7 floats (qx qy qz qw tx ty tz) = previous_frame->rigid_transformation_ (:595).
 * Dense points (Tracker::ObtainAllPoints, :1259-1310) are implicit: the pixel grid is never materialised.
 * Synchronous.  How the launches are laid out follows the batch (same results either way): a few pairs per call take
 * the chained flow (one launch per evaluation, the coarsest levels in one launch, results written straight into
 * page-locked memory; one or two pairs of an early-exit schedule are launched without read-backs, each level with a budget of
 * evaluations, and run again with twice the budget — then with read-backs — if a level was cut short), batches take one
 * residual and one update launch per evaluation (coarse levels one launch per level, one block per pair), large fixed-schedule
 * batches run as two halves on two streams with the update in the tail of the residual launch. */
int uwt_estimate_pose_batch(uwt_ctx* ctx, int32_t n_pairs, const int32_t* ref_slots, const int32_t* tgt_slots,
                            float* poses_out, uwt_stats* stats_out_or_null);

/* Whole per-frame path for a resident batch: pyramids of slots [first_slot, first_slot+n_frames), gradients of
 * the same slots, then EstimatePose for the pairs — enqueued on the context stream, results written to DEVICE
 * memory (d_poses_out: n_pairs x 7 floats, d_stats_out_or_null: n_pairs uwt_stats).  uwt_sync() to wait.
 * grad_refs_only != 0: the planes the tracker reads of the reference frame only — gradients (src/Tracker.cpp:407-408)
 * and the depth pyramid levels 1.. (:1266-1272) — are built for the pairs' ref_slots alone (wherever they lie);
 * the image pyramids still cover the whole slot range.  0 prepares every frame of the range fully, as
 * System::AddFrame / System::Tracking do for each new frame.
 * Asynchronous with early_exit = 0 (fixed iteration counts: nothing in the call waits for the device).  With
 * early_exit = 1 — the reference's schedule, uwt_default_params' setting — the call itself waits for the device a few
 * times per level above 6144 pixels (after evaluations 3, 6, 12, ... it reads back how many pairs are still iterating — each
 * look taken while the next evaluation already runs — and stops launching for a level every pair has left; smaller levels
 * decide on the device), so it returns only once the last level's first iterations are queued.
 * Ordering against uwt_upload_frames_async covers every slot the call touches: the prepared range AND every slot the pair
 * lists name (they may lie outside the range). */
int uwt_track_batch_async(uwt_ctx* ctx, int32_t first_slot, int32_t n_frames, int32_t grad_refs_only,
                          int32_t n_pairs, const int32_t* ref_slots, const int32_t* tgt_slots,
                          float* d_poses_out, uwt_stats* d_stats_out_or_null);
/* The same call with the results copied behind it into host buffers (page-locked: uwt_host_alloc) and a ticket to wait
 * on: the streaming form — while this batch is aligned the caller uploads the next one into other slots
 * (uwt_upload_frames_async) and only waits for the batch before. */
int uwt_track_batch_host_async(uwt_ctx* ctx, int32_t first_slot, int32_t n_frames, int32_t grad_refs_only,
                               int32_t n_pairs, const int32_t* ref_slots, const int32_t* tgt_slots,
                               float* h_poses_out, uwt_stats* h_stats_out_or_null, int64_t* ticket_out);
/* blocks until the call that returned `ticket` (and its result copy) has completed; later calls keep running */
int uwt_wait_ticket(uwt_ctx* ctx, int64_t ticket);
int uwt_sync(uwt_ctx* ctx);
/* Deferred stage calls (off by default).  on = 1: uwt_build_pyramids and uwt_apply_gradient return once their kernels are
 * enqueued on the context's stream; every later call of the context runs behind them, and the calls that hand results to
 * the host (uwt_estimate_pose_*, uwt_get_plane, uwt_sync, ...) wait as before — the per-frame sequence of
 * System::AddFrame + System::Tracking (src/System.cpp:193-251) then waits once per frame, in EstimatePose, instead of
 * three times.  A kernel failure of a deferred call is reported by the next waiting call.  A consumer that reads
 * uwt_plane_device_ptr planes on a stream of its own must uwt_sync first. */
int uwt_set_deferred(uwt_ctx* ctx, int32_t on);
/* the HIP stream (hipStream_t) the context launches on, for event timing by the caller */
int uwt_stream(uwt_ctx* ctx, void** out);
/* average device time in ms of the residual/Jacobian/reduction kernel launches and their count since the last
 * reset (HIP events on the context stream; only recorded while profiling is enabled).
 * `on` is a bit mask: 1 = record the events; 2 (diagnostic) = the dense residual launches run their compute-only twin —
 * the same instruction stream with every memory operation of the loop replaced by register arithmetic.  Poses computed
 * with bit 2 set are meaningless; the launch durations are the kernel's own instruction-issue floor (bench.py's
 * roofline.valu).  0 switches both off. */
int uwt_profile_enable(uwt_ctx* ctx, int32_t on);
int uwt_profile_read(uwt_ctx* ctx, double* residual_ms_total, int64_t* residual_launches, int64_t* residual_pixels);
/* the same durations and launch counts by pyramid level (arrays of n_levels entries) */
int uwt_profile_read_levels(uwt_ctx* ctx, double* ms_by_level, int64_t* launches_by_level, int32_t n_levels);
/* Shader clock (GHz) the chip held inside the last profiled k_residual launch: blocks of a profiled launch leave their
 * s_memtime (shader cycles) and s_memrealtime (100 MHz) deltas in their records.  Diagnostic; synchronises. */
int uwt_profile_clock(uwt_ctx* ctx, double* shader_ghz);

/* ---- per-stage entry points (each kernel parity-testable alone; host buffers, synchronous) ------------------ */

/* cv::resize(src, dst, Size(), 0.5, 0.5) as used at src/System.cpp:247 / :249, even sizes (dst: w/2 x h/2) */
int uwt_halve_u8(uwt_ctx* ctx, const uint8_t* src, int32_t w, int32_t h, uint8_t* dst);
int uwt_halve_u16(uwt_ctx* ctx, const uint16_t* src, int32_t w, int32_t h, uint16_t* dst);
/* the same call on ANY size: dst is uwt_half_size(w) x uwt_half_size(h) = cvRound(size * 0.5) (half to even); whole 2 x 2 cells
 * (a + b + c + d + 2) >> 2, the half cells of a partial last column and every cell of a partial last row by resizeAreaFast's
 * generic tail: the mean of the pixels that exist, rounded half to even */
int uwt_half_size(int32_t n);
int uwt_resize_half_u8(uwt_ctx* ctx, const uint8_t* src, int32_t w, int32_t h, uint8_t* dst);
int uwt_resize_half_u16(uwt_ctx* ctx, const uint16_t* src, int32_t w, int32_t h, uint16_t* dst);
/* cv::Scharr(src, dst, CV_16S, 1|0, 0|1, scale=3, 0, BORDER_DEFAULT) as used at src/Tracker.cpp:1133-1134 */
int uwt_scharr3(uwt_ctx* ctx, const uint8_t* src, int32_t w, int32_t h, int16_t* gx, int16_t* gy);
/* Tracker::WarpFunction(points, T, lvl) (src/Tracker.cpp:1417-1471): n x 4 in, n x 4 out */
int uwt_warp(uwt_ctx* ctx, int32_t lvl, const float* pts, int32_t n, const float pose[7], float* warped_out);
/* one pass of the per-point loop of EstimatePose (src/Tracker.cpp:432-490) + the reduction, for one pair at one
 * level under `pose`.  Optional per-point dumps (the level's w x h point grid, row-major): J_out (x6), r_out, valid_out. */
int uwt_residual_jacobian(uwt_ctx* ctx, int32_t ref_slot, int32_t tgt_slot, int32_t lvl, const float pose[7],
                          uwt_accum* acc_out, float* J_out_or_null, float* r_out_or_null, uint8_t* valid_out_or_null);
/* Same under the context's sampler / weights (general path): additionally returns the per-pixel robust weights
 * (Tracker::TukeyFunctionWeights, src/Tracker.cpp:1626-1654), the scale 1/MAD, the error numerator Σ r·(r·w) and the
 * accumulators of the weighted system (J <- w·J, r <- gain·r; :554-561): acc_out->jtr then holds Σ(wJ)·(gain·r·w). */
int uwt_residual_jacobian_weighted(uwt_ctx* ctx, int32_t ref_slot, int32_t tgt_slot, int32_t lvl, const float pose[7],
                                   uwt_accum* acc_out, double* err_num_out, float* inv_mad_out, float* J_out_or_null,
                                   float* r_out_or_null, uint8_t* valid_out_or_null, float* w_out_or_null);
/* LS::initialize + n x LS::update(J, r, w) + LS::finishNoDivide / finish  (src/LeastSquares.cpp:30-37, 204-209,
 * 39-146).  Every accumulator is the f32 chain the reference's loop forms, in call order (one GPU thread per chain: the 28 chains
 * are independent) — the reference's floats bit for bit, not a re-associated sum.  A: 36 row-major, b: 6 (stored sign:
 * b = -Σ w r J), error, count. */
int uwt_ls_accumulate(uwt_ctx* ctx, const float* J, const float* r, const float* w_or_null, int32_t n, int32_t divide,
                      float A[36], float b[6], float* error, int32_t* num_constraints);
/* LS::initialize + (n/4) x LS::updateSSE + LS::finishNoDivide / finish (src/LeastSquares.cpp:148-202): the 4-wide form's
 * product association ((J_i·w)·J_j, (r·w)·J_i, (r·w)·r); n must be a multiple of 4.  count_quirk != 0 reproduces
 * "num_constraints += 6" per four points (:201); 0 counts 4. */
int uwt_ls_accumulate_sse(uwt_ctx* ctx, const float* J, const float* r, const float* w, int32_t n, int32_t divide,
                          int32_t count_quirk, float A[36], float b[6], float* error, int32_t* num_constraints);
/* Sophus::SE3f::exp (thirdparty/sophus/se3.hpp:723-744) */
int uwt_se3_exp(uwt_ctx* ctx, const float xi[6], float pose_out[7]);
/* SE3f::operator* (se3.hpp:285-321) */
int uwt_se3_mul(uwt_ctx* ctx, const float a[7], const float b[7], float out[7]);
/* SE3f::matrix() (se3.hpp:253-268), row-major 4x4 */
int uwt_se3_matrix(uwt_ctx* ctx, const float pose[7], float T_out[16]);
/* level hand-off of EstimatePose (src/Tracker.cpp:580-590) */
int uwt_se3_handoff(uwt_ctx* ctx, float pose_inout[7], int32_t scale_t);
/* deltaMat = A.inv() * b (src/Tracker.cpp:564); Ainv_out_or_null receives cv::Mat::inv()'s result */
int uwt_solve_delta(uwt_ctx* ctx, const float A[36], const float b[6], float delta_out[6], float* Ainv_out_or_null,
                    int32_t* nonsingular_out_or_null);

/* ---- sparse point tables (Frame::candidatePoints_[lvl]) -------------------------------------------- */

/* Tracker::EstimatePose / EstimatePoseFeatures over explicit per-level point tables (src/Tracker.cpp:401, 669): tables[l]
 * is an n_points[l] x 4 host array [x y z w] for every level l in [last_level, first_level] (other entries ignored).
 * With the EstimatePoseFeatures constants (first = last = 0, max_iters 10, gain 1, z_factor 0.002, handoff_scale_t 1;
 * src/Tracker.cpp:634-640, 834, 856) this is the reference's live tracking call.  uwt_params::weights and ::sampler apply as in
 * the dense call.  A row whose reference position ((int)y, (int)x) lies outside the level is dropped (the reference's
 * Mat::at would read outside the image, :474-477). */
int uwt_estimate_pose_points(uwt_ctx* ctx, int32_t ref_slot, int32_t tgt_slot, const float* const* tables,
                             const int32_t* n_points, float pose_out[7], uwt_stats* stats_out_or_null);
/* Tracker::MedianMat (src/Tracker.cpp:1571-1594), MedianAbsoluteDeviation (:1607-1619), IdentityWeights (:1621-1624) and
 * TukeyFunctionWeights (:1626-1654) on an explicit N x 1 residual vector (host pointers).  kind: 0 identity, 1 Tukey.
 * median_out = MedianMat(residuals) (256-bin histogram of the rounded values saturated to u8, first bin whose cumulative
 * count exceeds float(n / 2)); mad_out = 1.4826 * MedianMat(|residuals - median|); weights_out (n floats, or NULL when
 * only the statistics are wanted): ones, or (1 - (x / 4.6851)^2)^2 for |x| <= 4.6851 and 0 beyond, x = r / MAD (MAD = 1
 * when it is 0). */
int uwt_robust_weights(uwt_ctx* ctx, const float* residuals, int32_t n, int32_t kind, float* weights_out,
                       float* median_out_or_null, float* mad_out_or_null);

/* frame->gradient_[lvl] (src/Tracker.cpp:1136-1142): u8 plane copied to the host */
int uwt_gradient_magnitude(uwt_ctx* ctx, int32_t slot, int32_t lvl, uint8_t* mag_out);
/* Tracker::ObtainCandidatePoints for one level (src/Tracker.cpp:1314-1362): gradient_ > mean + threshold
 * (GRADIENT_THRESHOLD = 20, src/Options.cpp:27), x-major order; needs uwt_apply_gradient on the slot first.
 * Writes min(count, cap) points; *count_out is the full count. */
int uwt_obtain_candidate_points(uwt_ctx* ctx, int32_t slot, int32_t lvl, double threshold, float* pts_out, int32_t cap,
                                int32_t* count_out);
/* The same for frames first_slot .. first_slot + n_frames - 1 in one call (many blocks per frame: per-column counts by
 * row band, an exclusive scan in the reference's x-major order, ordered write).  pts_out: n_frames x cap x 4 floats (frame
 * f's points start at f * cap * 4), counts_out: n_frames full counts. */
int uwt_obtain_candidate_points_batch(uwt_ctx* ctx, int32_t first_slot, int32_t n_frames, int32_t lvl, double threshold,
                                      float* pts_out, int32_t cap, int32_t* counts_out);
/* Tracker::ObtainPatchesPoints (src/Tracker.cpp:1178-1257): 11x11 level-0 patches around <= 200 key points (x, y). */
int uwt_obtain_patch_points(uwt_ctx* ctx, int32_t slot, const float* keypoints_xy, int32_t n_keypoints, float* pts_out,
                            int32_t cap, int32_t* count_out);
/* Tracker::AddPatchPointsFeatures(candidatePoints, lvl) (src/Tracker.cpp:599-629; include/Tracker.h:126; its only call,
 * :672, is commented out in the reference): the N x 4 table followed, point by point, by the cells of the patch_size x
 * patch_size patch around each rounded point that lie inside the level (i > 0, j > 0) and are not the centre, with the
 * point's z and w = 1.  The reference's patch_size_ is 5 (:274).  count_out: the full count (n_pts + cells), of which at
 * most cap rows are written. */
int uwt_add_patch_points(uwt_ctx* ctx, int32_t lvl, const float* pts, int32_t n_pts, int32_t patch_size, float* pts_out,
                         int32_t cap, int32_t* count_out);

/* ---- next to the path: frame ingest ---------------------------------------------------------------- */

typedef struct uwt_ingest uwt_ingest;

/* CameraModel::GetCameraModel's rectification setup (src/CameraModel.cpp:84-90): getOptimalNewCameraMatrix(K, dist,
 * Size(in), 1.0, Size(out)) and initUndistortRectifyMap(K, dist, Mat(), newK, Size(out), CV_16SC2, map1, map2), computed
 * once on the host and kept on the device.  K and newK_out are (fx, fy, cx, cy); dist is (k1, k2, p1, p2). */
int uwt_ingest_create(const float K[4], const float dist[4], int32_t in_w, int32_t in_h, int32_t out_w, int32_t out_h,
                      int32_t device, uwt_ingest** out, float newK_out[4]);
int uwt_ingest_destroy(uwt_ingest* ing);
/* the fixed-point maps (out_h x out_w x 2 int16, out_h x out_w uint16), for inspection */
int uwt_ingest_maps(uwt_ingest* ing, int16_t* map1_out, uint16_t* map2_out);
/* remap(raw, undistorted, map1, map2, INTER_LINEAR) of a whole frame (src/System.cpp:152, :233) to host memory */
int uwt_ingest_undistort(uwt_ingest* ing, const uint8_t* raw, size_t row_stride, uint8_t* undistorted_out);
/* System::CalculateROI (src/System.cpp:148-191) on a raw first frame: roi_out = x, y, w, h */
int uwt_ingest_calculate_roi(uwt_ingest* ing, const uint8_t* raw_first, size_t row_stride, int32_t roi_out[4]);
/* remap + crop of System::AddFrame (src/System.cpp:231-235) fused on the GPU: the ctx->width x ctx->height window of the
 * undistorted frame starting at (x0, y0) lands directly in frame slot `slot` of `ctx` (level 0, image plane). */
int uwt_ingest_frame(uwt_ingest* ing, uwt_ctx* ctx, int32_t slot, const uint8_t* raw, size_t row_stride, int32_t x0,
                     int32_t y0);

/* ---- next to the path: trajectory accumulation (Visualizer::UpdateMessages, src/Visualizer.cpp:304-325) --------- */

/* final_i = final_{i-1} * SE3(q_i, t_scale * t_i), start = previous_pose_ (identity or the ground-truth start,
 * src/Visualizer.cpp:240-258).  reference_axes != 0 additionally publishes position as (-z, -x, -y) (:318-320).
 * The reference uses t_scale = 40, reference_axes = 1; (1, 0) is the plain SE(3) prefix product.
 * poses: n x 7 host floats (per-pair poses from uwt_estimate_pose_batch); traj_out: n x 7 host floats. */
int uwt_accumulate_trajectory(uwt_ctx* ctx, const float* poses, int32_t n, const float start_pose[7], float t_scale,
                              int32_t reference_axes, float* traj_out);
/* The same accumulation as a parallel prefix product (one block: per-thread runs, a scan over the run products, replay).
 * SE(3) composition is associative; in floats the grouping shows in the last bits, so results agree with the sequential
 * form to rounding, not bit for bit — the clean default for long trajectories; the reference-visualiser mode
 * (t_scale 40, reference_axes 1) that has to reproduce src/Visualizer.cpp:304-325 exactly stays sequential. */
int uwt_accumulate_trajectory_scan(uwt_ctx* ctx, const float* poses, int32_t n, const float start_pose[7], float t_scale,
                                   int32_t reference_axes, float* traj_out);

#ifdef __cplusplus
}
#endif
#endif
