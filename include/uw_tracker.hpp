// uw_tracker.hpp — header-only C++ mirror of the reference's direct-tracking class surface over the C ABI (uwt.h).
//
// Same class / method names, argument meaning and call order as the reference's include/Tracker.h:90-170,
// include/LeastSquares.h:26-50 and the Frame members of include/System.h:63-103 that the path touches, so that
// System::InitializeSystem / System::Tracking (src/System.cpp:121-123, 193-223) read the same against this header:
//
//     tracker_ = new uw::Tracker(depth_available_);
//     tracker_->InitializePyramid(w_, h_, K_);
//     ...
//     tracker_->ApplyGradient(previous_frame_);  tracker_->ApplyGradient(current_frame_);
//     tracker_->ObtainAllPoints(previous_frame_);
//     tracker_->EstimatePose(previous_frame_, current_frame_);   // -> previous_frame_->rigid_transformation_
//
// OpenCV / Eigen / Sophus are not required: images are passed as uw::ImageView (data, rows, cols, step — the four
// cv::Mat fields the path reads); define UW_WITH_OPENCV before including to get the cv::Mat overloads, UW_WITH_EIGEN to make
// uw::Mat61f / uw::Mat66f (LS::A, LS::b, LS::update's Jacobian) the reference's Eigen types instead of the stand-ins below.
// Neither branch has been compiled in the image this repository is developed in (it has no OpenCV and no Eigen).
// Every numeric step runs in libuwt_hip.so on the GPU; a non-zero status becomes a std::runtime_error (the reference
// surfaces misuse as cv::Exception / SOPHUS_ENSURE aborts).
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "uwt.h"

#ifdef __SSE__
#include <xmmintrin.h>
#endif

#ifdef UW_WITH_OPENCV
#include <opencv2/core.hpp>
#endif
#ifdef UW_WITH_EIGEN
#include <Eigen/Cholesky>
#include <Eigen/Core>
#endif

namespace uw {

constexpr int PYRAMID_LEVELS = 5;  // src/Options.cpp:26

// Storage of Sophus::SE3f: unit quaternion (x y z w) + translation.
struct SE3 {
  float q[4] = {0.f, 0.f, 0.f, 1.f};
  float t[3] = {0.f, 0.f, 0.f};
  const float* data() const { return q; }  // 7 contiguous floats
  float* data() { return q; }
};
static_assert(sizeof(SE3) == 7 * sizeof(float), "SE3 must be 7 packed floats");

namespace detail {
// The context a default-constructed LS folds on ("LS ls;", src/Tracker.cpp:537): LS::bind(ctx), the first Tracker's
// context, or — when neither exists — a minimal context created for the reductions alone.
inline uwt_ctx*& ls_default_ctx() { static uwt_ctx* c = nullptr; return c; }
inline uwt_ctx* ls_context() {
  uwt_ctx*& d = ls_default_ctx();
  if (!d) {   // nothing bound and no tracker yet: a minimal context for the reductions (lives for the process)
    uwt_params p;
    int st = uwt_default_params(&p, 64, 48, 64.f, 64.f, 31.5f, 23.5f);
    if (st == UWT_OK) {
      p.n_levels = 1; p.first_level = 0; p.last_level = 0; p.max_frames = 2; p.max_pairs = 1;
      st = uwt_create(&p, &d);
    }
    if (st != UWT_OK) throw std::runtime_error(std::string("LS: no context: ") + uwt_status_string(st));
  }
  return d;
}
}  // namespace detail

// include/Options.h:146-147.  With UW_WITH_EIGEN these ARE the reference's Eigen types; without Eigen, stand-ins with the
// members the path's code uses (operator(), data(), unary minus, ldlt().solve()).  The stand-in's solve() is the library's
// 6x6 solve on the GPU (uwt_solve_delta: the LU inverse of the live path, src/Tracker.cpp:564) — no host factorisation.
#ifdef UW_WITH_EIGEN
typedef Eigen::Matrix<float, 6, 1> Mat61f;
typedef Eigen::Matrix<float, 6, 6> Mat66f;
#else
struct Mat61f {
  float v[6] = {0, 0, 0, 0, 0, 0};
  float& operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
  float* data() { return v; }
  const float* data() const { return v; }
  Mat61f operator-() const { Mat61f r; for (int i = 0; i < 6; i++) r.v[i] = -v[i]; return r; }
};
struct Mat66f {
  float v[36] = {};   // row-major (A is symmetric: reads the same column-major)
  float& operator()(int r, int c) { return v[6 * r + c]; }
  float operator()(int r, int c) const { return v[6 * r + c]; }
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
  float* data() { return v; }
  const float* data() const { return v; }
  struct Solver {
    const Mat66f* A;
    Mat61f solve(const Mat61f& b) const {
      Mat61f x;
      const int st = uwt_solve_delta(detail::ls_context(), A->v, b.v, x.v, nullptr, nullptr);
      if (st != UWT_OK) throw std::runtime_error(std::string("Mat66f::ldlt().solve: ") + uwt_status_string(st));
      return x;
    }
  };
  Solver ldlt() const { return Solver{this}; }
};
#endif

struct ImageView {
  const void* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;  // bytes per row (cv::Mat::step)
  ImageView() = default;
  ImageView(const void* d, int r, int c, size_t s) : data(d), rows(r), cols(c), step(s) {}
#ifdef UW_WITH_OPENCV
  ImageView(const cv::Mat& m) : data(m.data), rows(m.rows), cols(m.cols), step(m.step) {}
#endif
};

struct Size {  // cv::Size() as System::AddFrame passes it to resize (src/System.cpp:247)
  int width = 0, height = 0;
};

// cv::resize(src, dst, Size(), 0.5, 0.5) of System::AddFrame's pyramid loop (src/System.cpp:246-251).  The pixels of
// levels 1.. are produced on the GPU when the tracker binds the frame (uwt_build_pyramids: the same 2x2 round-half-up
// mean); here the destination view only takes its shape, with no host pixels (data == nullptr), so the loop compiles
// and reads as it does in the reference.
inline void resize(const ImageView& src, ImageView& dst, Size /*dsize*/, double fx, double fy) {
  dst = ImageView(nullptr, (int)(src.rows * fy + 0.5), (int)(src.cols * fx + 0.5), 0);
}

class Tracker;

// include/System.h:63-103 — the members the tracker reads or writes.
class Frame {
 public:
  Frame() = default;
  ~Frame();                       // releases the device slot the frame holds, if any
  Frame(const Frame&) = delete;   // a bound frame is known to its tracker by address
  Frame& operator=(const Frame&) = delete;

  int idFrame_ = 0;
  std::vector<ImageView> images_ = std::vector<ImageView>(PYRAMID_LEVELS);  // [0]: CV_8UC1 host pixels; [1..]: shapes only
  std::vector<ImageView> depths_ = std::vector<ImageView>(PYRAMID_LEVELS);  // [0]: CV_16UC1 host pixels (optional)
  bool depth_available_ = false;
  bool obtained_gradients_ = false;
  bool obtained_candidatePoints_ = false;
  SE3 rigid_transformation_;
  std::vector<float> keypoints_;                         // x0 y0 x1 y1 ... (cv::KeyPoint::pt of Frame::keypoints_)
  std::vector<float> candidatePoints_[PYRAMID_LEVELS];   // N x 4 [x y z w] per level when a sparse producer ran
  int slot_ = -1;                 // device frame slot while bound; -1 again once the slot has gone to another frame
  Tracker* tracker_ = nullptr;    // the tracker that holds the slot
};

class Tracker {
 public:
  // include/Tracker.h:97.  `overrides` lets a caller change the constants the reference hard-codes as locals of
  // EstimatePose (src/Tracker.cpp:364-372); by default they are exactly those.
  explicit Tracker(bool _depth_available, int max_frames = 16, int device = 0)
      : depth_available_(_depth_available), max_frames_(max_frames < 2 ? 2 : max_frames), device_(device),
        owner_((size_t)(max_frames < 2 ? 2 : max_frames), nullptr), last_use_((size_t)(max_frames < 2 ? 2 : max_frames), 0) {}
  ~Tracker() {
    for (Frame* f : owner_)
      if (f) { f->slot_ = -1; f->tracker_ = nullptr; f->obtained_gradients_ = false; }
    if (ctx_ && detail::ls_default_ctx() == ctx_) detail::ls_default_ctx() = nullptr;
    if (ctx_) uwt_destroy(ctx_);
  }
  // a frame that goes away (System::FreeFrames, src/System.cpp:352-355) gives its slot back
  void forget(Frame* f) {
    if (f->slot_ >= 0 && f->slot_ < (int)owner_.size() && owner_[(size_t)f->slot_] == f) owner_[(size_t)f->slot_] = nullptr;
    f->slot_ = -1;
    f->tracker_ = nullptr;
  }
  Tracker(const Tracker&) = delete;
  Tracker& operator=(const Tracker&) = delete;

  uwt_params& params() { return params_; }  // valid after InitializePyramid's defaults; edit before first use

  // include/Tracker.h:112; K row-major 3x3 (fx 0 cx; 0 fy cy; 0 0 1)
  void InitializePyramid(int _width, int _height, const float K[9]) {
    check(uwt_default_params(&params_, _width, _height, K[0], K[4], K[2], K[5]), "uwt_default_params");
    params_.n_levels = PYRAMID_LEVELS;
    params_.has_depth = depth_available_ ? 1 : 0;
    params_.max_frames = max_frames_;
    params_.max_pairs = max_frames_ > 1 ? max_frames_ / 2 : 1;
    params_.device = device_;
  }
#ifdef UW_WITH_OPENCV
  void InitializePyramid(int _width, int _height, const cv::Mat& _K) {
    const float K[9] = {_K.at<float>(0, 0), 0, _K.at<float>(0, 2), 0, _K.at<float>(1, 1), _K.at<float>(1, 2), 0, 0, 1};
    InitializePyramid(_width, _height, K);
  }
#endif
  void InitializeMasks() {}  // src/Tracker.cpp:342-359 builds masks nothing reads

  void ApplyGradient(Frame* _frame) {  // include/Tracker.h:137
    const int slot = bind(_frame);
    check(uwt_apply_gradient(ctx(), slot, 1), "uwt_apply_gradient");
    _frame->obtained_gradients_ = true;
  }
  void ObtainAllPoints(Frame* _frame) {  // include/Tracker.h:153 — the dense table is implicit on the GPU
    bind(_frame);
    _frame->obtained_candidatePoints_ = true;
  }
  // include/Tracker.h:170: N x 4 points in, N x 4 out
  std::vector<float> WarpFunction(const std::vector<float>& _points2warp, const SE3& _rigid_transformation, int _lvl) {
    std::vector<float> out(_points2warp.size());
    check(uwt_warp(ctx(), _lvl, _points2warp.data(), (int)(_points2warp.size() / 4), _rigid_transformation.data(), out.data()),
          "uwt_warp");
    return out;
  }
  // include/Tracker.h:126 (src/Tracker.cpp:599-629): the N x 4 table followed by the patch cells around every point
  // (patch_size_ = 5, :274).  Its only call in the reference is commented out (:672).
  std::vector<float> AddPatchPointsFeatures(const std::vector<float>& candidatePoints, int lvl) {
    const int n = (int)(candidatePoints.size() / 4), cap = n * patch_size_ * patch_size_;
    std::vector<float> out((size_t)(cap > 0 ? cap : 1) * 4);
    int32_t count = 0;
    check(uwt_add_patch_points(ctx(), lvl, candidatePoints.data(), n, patch_size_, out.data(), cap, &count), "uwt_add_patch_points");
    out.resize((size_t)(count < cap ? count : cap) * 4);
    return out;
  }
  // include/Tracker.h:178 (src/Tracker.cpp:1596-1605): 6 x 1 [w1 w2 w3 x1 x2 x3] -> SE3(SO3::exp(w), x); the translation
  // is taken as it is (not through V(w)), so the rotation is exp's of the tangent (0, w) and t is copied.
  SE3 Mat2SE3(const float _input[6]) {
    const float xi[6] = {0.f, 0.f, 0.f, _input[0], _input[1], _input[2]};
    SE3 T;
    check(uwt_se3_exp(ctx(), xi, T.data()), "uwt_se3_exp");
    T.t[0] = _input[3]; T.t[1] = _input[4]; T.t[2] = _input[5];
    return T;
  }
#ifdef UW_WITH_OPENCV
  SE3 Mat2SE3(const cv::Mat& _input) {
    const float v[6] = {_input.at<float>(0, 0), _input.at<float>(1, 0), _input.at<float>(2, 0),
                        _input.at<float>(3, 0), _input.at<float>(4, 0), _input.at<float>(5, 0)};
    return Mat2SE3(v);
  }
#endif
  // include/Tracker.h:122
  void EstimatePose(Frame* _previous_frame, Frame* _current_frame) {
    const int32_t a = bind(_previous_frame), b = bind(_current_frame);
    if (!_previous_frame->obtained_gradients_)
      throw std::runtime_error("EstimatePose: ApplyGradient(previous) not called (or its slot was reused since)");
    check(uwt_estimate_pose_batch(ctx(), 1, &a, &b, _previous_frame->rigid_transformation_.data(), &last_stats_),
          "uwt_estimate_pose_batch");
  }
  // include/Tracker.h:124 — the vectorised prototype's schedule: levels PYRAMID_LEVELS-1 .. 0, <= 50 iterations, gain 50,
  // epsilon 1e-3, no z / angle factor applied (src/Tracker.cpp:877-885, 1082).  The reference BODY is not reproduced: its
  // residual ignores the warp (:933-944) and column 0 of Jw1 is zeroed by a typo (:986); this runs the same per-point
  // terms as EstimatePose under that schedule.
  void FastEstimatePose(Frame* _previous_frame, Frame* _current_frame) {
    const int32_t a = bind(_previous_frame), b = bind(_current_frame);
    if (!_previous_frame->obtained_gradients_)
      throw std::runtime_error("FastEstimatePose: ApplyGradient(previous) not called (or its slot was reused since)");
    uwt_params saved;
    check(uwt_get_params(ctx(), &saved), "uwt_get_params");
    uwt_params f = saved;
    f.first_level = f.n_levels - 1; f.last_level = 0; f.max_iters = 50; f.gain = 50.0f; f.epsilon = 0.001f;
    f.z_factor = 1.0f; f.angle_factor = 1.0f; f.early_exit = 1; f.handoff_scale_t = 0;
    check(uwt_update_params(ctx(), &f), "uwt_update_params");
    const int st = uwt_estimate_pose_batch(ctx(), 1, &a, &b, _previous_frame->rigid_transformation_.data(), &last_stats_);
    uwt_update_params(ctx(), &saved);
    check(st, "uwt_estimate_pose_batch");
  }
  // host copy of one level of a bound frame's planes (Frame::images_/depths_/gradientX_/gradientY_[lvl] in the reference)
  void GetFrameLevel(Frame* _frame, int _lvl, int _plane, void* _host_out) {
    check(uwt_get_plane(ctx(), bind(_frame), _lvl, _plane, _host_out), "uwt_get_plane");
  }
  // include/Tracker.h:145 — gradient_ > mean + GRADIENT_THRESHOLD (src/Options.cpp:27) on every level, x-major order
  void ObtainCandidatePoints(Frame* _frame, double gradient_threshold = 20.0) {
    const int slot = bind(_frame);
    if (!_frame->obtained_gradients_)
      throw std::runtime_error("ObtainCandidatePoints: ApplyGradient not called (or the frame's slot was reused since)");
    for (int l = 0; l < params_.n_levels && l < PYRAMID_LEVELS; l++) {
      const uwt_level L = level(l);
      std::vector<float>& t = _frame->candidatePoints_[l];
      t.resize((size_t)L.w * L.h * 4);
      int32_t n = 0;
      check(uwt_obtain_candidate_points(ctx(), slot, l, gradient_threshold, t.data(), L.w * L.h, &n), "uwt_obtain_candidate_points");
      t.resize((size_t)n * 4);
    }
    _frame->obtained_candidatePoints_ = true;
  }
  // include/Tracker.h:155 — 11x11 level-0 patches around Frame::keypoints_ (at most 200)
  void ObtainPatchesPoints(Frame* _previous_frame) {
    const int slot = bind(_previous_frame);
    std::vector<float>& t = _previous_frame->candidatePoints_[0];
    const int cap = 200 * 144;
    t.resize((size_t)cap * 4);
    int32_t n = 0;
    check(uwt_obtain_patch_points(ctx(), slot, _previous_frame->keypoints_.data(), (int)(_previous_frame->keypoints_.size() / 2),
                                  t.data(), cap, &n), "uwt_obtain_patch_points");
    t.resize((size_t)(n < cap ? n : cap) * 4);
    _previous_frame->obtained_candidatePoints_ = true;
  }
  // include/Tracker.h:128 — the reference's live variant: level 0 only, 10 iterations, gain 1, z_factor 0.002
  // (src/Tracker.cpp:634-640, 834), over previous->candidatePoints_[0]
  void EstimatePoseFeatures(Frame* _previous_frame, Frame* _current_frame) {
    const int32_t a = bind(_previous_frame), b = bind(_current_frame);
    if (!_previous_frame->obtained_gradients_)
      throw std::runtime_error("EstimatePoseFeatures: ApplyGradient(previous) not called (or its slot was reused since)");
    uwt_params saved;
    check(uwt_get_params(ctx(), &saved), "uwt_get_params");
    uwt_params f = saved;
    f.first_level = 0; f.last_level = 0; f.max_iters = 10; f.gain = 1.0f; f.z_factor = 0.002f; f.angle_factor = 1.0f;
    f.handoff_scale_t = 1; f.early_exit = 1;
    check(uwt_update_params(ctx(), &f), "uwt_update_params");
    const float* tables[UWT_MAX_LEVELS] = {};
    int32_t counts[UWT_MAX_LEVELS] = {};
    tables[0] = _previous_frame->candidatePoints_[0].data();
    counts[0] = (int32_t)(_previous_frame->candidatePoints_[0].size() / 4);
    const int st = uwt_estimate_pose_points(ctx(), a, b, tables, counts, _previous_frame->rigid_transformation_.data(), &last_stats_);
    uwt_update_params(ctx(), &saved);
    check(st, "uwt_estimate_pose_points");
  }
  // EstimatePose over the sparse tables a producer left in previous->candidatePoints_[lvl] (src/Tracker.cpp:401)
  void EstimatePoseOverCandidatePoints(Frame* _previous_frame, Frame* _current_frame) {
    const int32_t a = bind(_previous_frame), b = bind(_current_frame);
    const float* tables[UWT_MAX_LEVELS] = {};
    int32_t counts[UWT_MAX_LEVELS] = {};
    for (int l = 0; l < PYRAMID_LEVELS; l++) {
      tables[l] = _previous_frame->candidatePoints_[l].data();
      counts[l] = (int32_t)(_previous_frame->candidatePoints_[l].size() / 4);
    }
    check(uwt_estimate_pose_points(ctx(), a, b, tables, counts, _previous_frame->rigid_transformation_.data(), &last_stats_),
          "uwt_estimate_pose_points");
  }
  // include/Tracker.h:224 / :235 — selects the weighting of the following EstimatePose calls (the reference switches by
  // commenting src/Tracker.cpp:495-496): 0 IdentityWeights, 1 TukeyFunctionWeights, 2 Huber (extension)
  // include/Tracker.h:197 — gradients of a tightly packed u8 image (3 x Scharr, CV_16S)
  void ObtainGradientXY(const ImageView& _inputImage, std::vector<int16_t>& _gradientX, std::vector<int16_t>& _gradientY) {
    if (_inputImage.step != (size_t)_inputImage.cols) throw std::runtime_error("ObtainGradientXY: image rows must be contiguous");
    _gradientX.resize((size_t)_inputImage.rows * _inputImage.cols);
    _gradientY.resize(_gradientX.size());
    check(uwt_scharr3(ctx(), static_cast<const uint8_t*>(_inputImage.data), _inputImage.cols, _inputImage.rows, _gradientX.data(),
                      _gradientY.data()), "ObtainGradientXY");
  }

  // include/Tracker.h:206-235 — the weights helpers on an explicit residual vector
  float MedianMat(const std::vector<float>& _input) {
    float med = 0.f;
    check(uwt_robust_weights(ctx(), _input.data(), (int32_t)_input.size(), 0, nullptr, &med, nullptr), "MedianMat");
    return med;
  }
  float MedianAbsoluteDeviation(const std::vector<float>& x) {
    float mad = 0.f;
    check(uwt_robust_weights(ctx(), x.data(), (int32_t)x.size(), 0, nullptr, nullptr, &mad), "MedianAbsoluteDeviation");
    return mad;
  }
  std::vector<float> IdentityWeights(int _num_residuals) { return std::vector<float>((size_t)_num_residuals, 1.0f); }
  std::vector<float> TukeyFunctionWeights(const std::vector<float>& _residuals) {
    std::vector<float> w(_residuals.size());
    check(uwt_robust_weights(ctx(), _residuals.data(), (int32_t)_residuals.size(), 1, w.data(), nullptr, nullptr),
          "TukeyFunctionWeights");
    return w;
  }

  void SetWeights(int weights) {
    uwt_params p;
    check(uwt_get_params(ctx(), &p), "uwt_get_params");
    p.weights = weights;
    check(uwt_update_params(ctx(), &p), "uwt_update_params");
  }
  const uwt_stats& last_stats() const { return last_stats_; }
  uwt_level level(int lvl) {  // w_/h_/fx_/fy_/cx_/cy_/invfx_/invfy_[lvl], include/Tracker.h:516-526
    uwt_level L;
    check(uwt_level_info(ctx(), lvl, &L), "uwt_level_info");
    return L;
  }
  uwt_ctx* ctx() {
    if (!ctx_) {
      check(uwt_create(&params_, &ctx_), "uwt_create");
      // the per-frame sequence (upload + pyramid in bind(), ApplyGradient, EstimatePose) waits once, in EstimatePose
      check(uwt_set_deferred(ctx_, 1), "uwt_set_deferred");
      if (!detail::ls_default_ctx()) detail::ls_default_ctx() = ctx_;   // "LS ls;" inside the tracking loop folds here
    }
    return ctx_;
  }

 private:
  void check(int st, const char* what) {
    if (st != UWT_OK)
      throw std::runtime_error(std::string(what) + ": " + uwt_status_string(st) + (ctx_ ? std::string(" — ") + uwt_last_error(ctx_) : ""));
  }
  // System::AddFrame's pyramid loop (src/System.cpp:246-251): upload level 0, build the other levels on the GPU.
  // Slots are handed out least-recently-used first; the frame that held a reused slot is told so (slot_ = -1, its
  // gradient flag cleared): it uploads again when it is next used instead of silently reading another frame's planes.
  int bind(Frame* f) {
    if (f->slot_ >= 0 && f->tracker_ == this && owner_[(size_t)f->slot_] == f) {
      last_use_[(size_t)f->slot_] = ++use_clock_;
      return f->slot_;
    }
    if (!f->images_[0].data) throw std::runtime_error("bind: Frame::images_[0] has no pixels");
    int slot = 0;
    for (int i = 0; i < max_frames_; i++) {
      if (!owner_[(size_t)i]) { slot = i; break; }
      if (last_use_[(size_t)i] < last_use_[(size_t)slot]) slot = i;
    }
    if (Frame* old = owner_[(size_t)slot]) {
      old->slot_ = -1;
      old->tracker_ = nullptr;
      old->obtained_gradients_ = false;
    }
    owner_[(size_t)slot] = f;
    last_use_[(size_t)slot] = ++use_clock_;
    f->slot_ = slot;
    f->tracker_ = this;
    f->obtained_gradients_ = false;
    check(uwt_set_frame(ctx(), slot, (const uint8_t*)f->images_[0].data, f->images_[0].step,
                        depth_available_ ? (const uint16_t*)f->depths_[0].data : nullptr, f->depths_[0].step),
          "uwt_set_frame");
    check(uwt_build_pyramids(ctx(), slot, 1), "uwt_build_pyramids");
    return slot;
  }
  bool depth_available_;
  int patch_size_ = 5;                 // src/Tracker.cpp:274
  int max_frames_, device_;
  std::vector<Frame*> owner_;          // slot -> frame holding it
  std::vector<uint64_t> last_use_;
  uint64_t use_clock_ = 0;
  uwt_params params_{};
  uwt_ctx* ctx_ = nullptr;
  uwt_stats last_stats_{};
};

inline Frame::~Frame() {
  if (tracker_) tracker_->forget(this);
}

// Four packed floats in place of __m128 where SSE is absent (LS::updateSSE's operands, include/LeastSquares.h:42).
struct f4 {
  float v[4];
};

// include/LeastSquares.h:26-50.  update() / updateSSE() buffer rows; finish*() folds them on the GPU (uwt_ls_accumulate for
// the scalar rows, uwt_ls_accumulate_sse for the 4-wide ones — the two forms associate their products differently,
// src/LeastSquares.cpp:151-153 vs :205) and adds the two, as finishNoDivide adds the lane sums onto A, b, error
// (src/LeastSquares.cpp:39-139).  b keeps the reference's stored sign: b = -Σ w r J (:206, :117-133).
// num_constraints counts 6 per updateSSE call like the reference (:201) unless count_quirk is switched off.
class LS {
 public:
  // "LS ls;" as the reference writes it (src/Tracker.cpp:537): folds on the default context (LS::bind / the first Tracker's)
  LS() : ctx_(nullptr) { initialize(0); }
  explicit LS(uwt_ctx* ctx) : ctx_(ctx) { initialize(0); }
  static void bind(uwt_ctx* ctx) { detail::ls_default_ctx() = ctx; }
  Mat66f A;   // include/LeastSquares.h:34-35 (row-major storage; A is symmetric)
  Mat61f b;
  float error;
  int num_constraints;
  bool count_quirk = true;

  void initialize(const int /*max_num_constraints*/) {
    J_.clear(); r_.clear(); w_.clear();
    J4_.clear(); r4_.clear(); w4_.clear();
    for (int i = 0; i < 36; i++) A.data()[i] = 0.f;
    for (int i = 0; i < 6; i++) b.data()[i] = 0.f;
    error = 0.f;
    num_constraints = 0;
  }
  void update(const float J[6], const float& res, const float& weight) {
    J_.insert(J_.end(), J, J + 6);
    r_.push_back(res);
    w_.push_back(weight);
  }
  // include/LeastSquares.h:40: update(const Mat61f& J, const float& res, const float& weight)
  void update(const Mat61f& J, const float& res, const float& weight) { update(J.data(), res, weight); }
  // four points at once: J1..J6 hold Jacobian component k of the four points (include/LeastSquares.h:42-43)
  void updateSSE(const f4& J1, const f4& J2, const f4& J3, const f4& J4, const f4& J5, const f4& J6, const f4& res,
                 const f4& weight) {
    const f4* Jk[6] = {&J1, &J2, &J3, &J4, &J5, &J6};
    for (int p = 0; p < 4; p++) {
      for (int k = 0; k < 6; k++) J4_.push_back(Jk[k]->v[p]);
      r4_.push_back(res.v[p]);
      w4_.push_back(weight.v[p]);
    }
  }
#ifdef __SSE__
  void updateSSE(const __m128& J1, const __m128& J2, const __m128& J3, const __m128& J4, const __m128& J5, const __m128& J6,
                 const __m128& res, const __m128& weight) {
    f4 a[8];
    const __m128* src[8] = {&J1, &J2, &J3, &J4, &J5, &J6, &res, &weight};
    for (int i = 0; i < 8; i++) _mm_storeu_ps(a[i].v, *src[i]);
    updateSSE(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
  }
#endif
  void finishNoDivide() { fold(); }
  void finish() {
    fold();
    const float n = (float)num_constraints;   // src/LeastSquares.cpp:141-146
    for (int i = 0; i < 36; i++) A.data()[i] /= n;
    for (int i = 0; i < 6; i++) b.data()[i] /= n;
    error /= n;
  }

 private:
  void fold() {
    float A1[36] = {}, b1[6] = {}, e1 = 0.f, A2[36] = {}, b2[6] = {}, e2 = 0.f;
    int32_t n1 = 0, n2 = 0;
    uwt_ctx* c = context();
    if (!r_.empty()) chk(uwt_ls_accumulate(c, J_.data(), r_.data(), w_.data(), (int)r_.size(), 0, A1, b1, &e1, &n1), "uwt_ls_accumulate");
    if (!r4_.empty())
      chk(uwt_ls_accumulate_sse(c, J4_.data(), r4_.data(), w4_.data(), (int)r4_.size(), 0, count_quirk ? 1 : 0, A2, b2, &e2, &n2),
          "uwt_ls_accumulate_sse");
    for (int i = 0; i < 36; i++) A.data()[i] = A1[i] + A2[i];
    for (int i = 0; i < 6; i++) b.data()[i] = b1[i] + b2[i];
    error = e1 + e2;
    num_constraints = n1 + n2;
  }
  static void chk(int st, const char* what) {
    if (st != UWT_OK) throw std::runtime_error(std::string(what) + ": " + uwt_status_string(st));
  }
  uwt_ctx* context() { return ctx_ ? ctx_ : detail::ls_context(); }
  uwt_ctx* ctx_;
  std::vector<float> J_, r_, w_, J4_, r4_, w4_;
};

}  // namespace uw
