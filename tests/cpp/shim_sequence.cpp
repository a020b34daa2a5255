// Drives the C++ shim (include/uw_tracker.hpp) through System::AddFrame's pyramid loop and System::Tracking()'s call
// sequence (src/System.cpp:225-251, 193-223) on two frames read from a raw file:  <w> <h> then w*h bytes (previous) and
// w*h bytes (current).  Prints, one line each: the EstimatePose result, the feature variant, the LS mirror (scalar and
// 4-wide rows), FastEstimatePose, and what happens to a frame whose device slot has been handed to another frame.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "uw_tracker.hpp"

using namespace uw;

// System::AddFrame (src/System.cpp:225-251) with the file read replaced by a buffer
static Frame* AddFrame(int _id, const std::vector<unsigned char>& pixels, int w, int h, bool depth_available_) {
  Frame* newFrame = new Frame();
  newFrame->idFrame_ = _id;
  newFrame->images_[0] = ImageView(pixels.data(), h, w, (size_t)w);
  for (int i = 1; i < PYRAMID_LEVELS; i++) {
    resize(newFrame->images_[i - 1], newFrame->images_[i], Size(), 0.5, 0.5);
    if (depth_available_) {
      resize(newFrame->depths_[i - 1], newFrame->depths_[i], Size(), 0.5, 0.5);
    }
  }
  return newFrame;
}

// Stand-ins for the cv::Mat the DSO-way block reads (rows, row(i), at<float>(i, 0)) and for OpenCV's cv2eigen on one row.
struct RowsMat {
  int rows = 0, cols = 0;
  std::vector<float> v;
  struct Row { const float* p; int n; };
  Row row(int i) const { return Row{v.data() + (size_t)i * cols, cols}; }
  template <typename T> T at(int i, int j) const { return (T)v[(size_t)i * cols + j]; }
};
static void cv2eigen(const RowsMat::Row& src, Mat61f& dst) {
  for (int k = 0; k < 6; k++) dst(k) = src.p[k];
}
static RowsMat make_jacobians() {   // 12 rows of 6, full rank
  RowsMat m; m.rows = 12; m.cols = 6;
  for (int i = 0; i < 12; i++)
    for (int k = 0; k < 6; k++) m.v.push_back(0.5f * (float)(((i + 1) * (k + 2) * 7) % 13) - 2.0f + (i % 6 == k ? 4.0f : 0.0f));
  return m;
}
static RowsMat make_column(float a, float step) {
  RowsMat m; m.rows = 12; m.cols = 1;
  for (int i = 0; i < 12; i++) m.v.push_back(a + step * (float)(i % 5));
  return m;
}

static void print_pose(const char* tag, const SE3& T, int iterations) {
  std::printf("%s %.9g %.9g %.9g %.9g %.9g %.9g %.9g %d\n", tag, T.q[0], T.q[1], T.q[2], T.q[3], T.t[0], T.t[1], T.t[2], iterations);
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  std::vector<unsigned char> a((size_t)w * h), b((size_t)w * h);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(a.data(), 1, a.size(), f) != a.size() || std::fread(b.data(), 1, b.size(), f) != b.size()) return 3;
  std::fclose(f);
  try {
    const float fl = 525.0f * w / 640.0f;
    const float K[9] = {fl, 0, w / 2 - 0.5f, 0, fl, h / 2 - 0.5f, 0, 0, 1};
    Tracker* tracker_ = new Tracker(false, /*max_frames=*/4);
    tracker_->InitializePyramid(w, h, K);
    if (argc > 4 && !std::strcmp(argv[4], "legacy")) tracker_->params().arith = UWT_ARITH_LEGACY;   // the parity suite runs both sets
    tracker_->InitializeMasks();
    std::unique_ptr<Frame> previous_frame_(AddFrame(0, a, w, h, false)), current_frame_(AddFrame(1, b, w, h, false));
    if (previous_frame_->images_[2].rows != h / 4 || previous_frame_->images_[2].cols != w / 4) return 4;
    tracker_->ApplyGradient(previous_frame_.get());
    tracker_->ApplyGradient(current_frame_.get());
    tracker_->ObtainAllPoints(previous_frame_.get());
    tracker_->EstimatePose(previous_frame_.get(), current_frame_.get());
    const SE3 first = previous_frame_->rigid_transformation_;
    print_pose("POSE", first, tracker_->last_stats().iterations);
    // the reference's live flow (src/System.cpp:193-223): key points -> ObtainPatchesPoints -> EstimatePoseFeatures
    for (int k = 0; k < 40; k++) {
      previous_frame_->keypoints_.push_back(8.0f + (float)((k * 37) % (w - 16)));
      previous_frame_->keypoints_.push_back(8.0f + (float)((k * 23) % (h - 16)));
    }
    tracker_->ObtainPatchesPoints(previous_frame_.get());
    tracker_->EstimatePoseFeatures(previous_frame_.get(), current_frame_.get());
    std::printf("FEATURES ");
    print_pose("", previous_frame_->rigid_transformation_, tracker_->last_stats().iterations);
    std::printf("NPATCH %d\n", (int)(previous_frame_->candidatePoints_[0].size() / 4));
    // LS mirror: one scalar row, closed form A = (J J^T) w, b = -w r J
    {
      LS ls;   // default-constructed, as the reference writes it (src/Tracker.cpp:537)
      const float J[6] = {1, 2, 3, 4, 5, 6};
      ls.update(J, 2.0f, 0.5f);
      ls.finishNoDivide();
      std::printf("LS %.9g %.9g %.9g %d\n", ls.A(0, 1), ls.b(2), ls.error, ls.num_constraints);
    }
    // The "Computation of new delta (DSO-way)" block of Tracker::EstimatePose, src/Tracker.cpp:537-550 — commented out in
    // the reference; written here as it stands there, over stand-ins for the three cv::Mat it reads (test data below).
    {
      const RowsMat Jacobians = make_jacobians(), Residuals = make_column(0.75f, -1.5f), W = make_column(1.0f, 0.25f);
      Mat61f deltaVector;
      // ---- src/Tracker.cpp:537-550 ----
      LS ls;
      ls.initialize(Residuals.rows);
      for (int i=0; i<Residuals.rows; i++) {
          Mat61f jacobian;
          cv2eigen(Jacobians.row(i), jacobian);

          ls.update(jacobian, Residuals.at<float>(i,0), W.at<float>(i,0));
      }
      ls.finish();
      // Solve LS system
      float LM_lambda = 0.2;
      Mat61f b = -ls.b;
      Mat66f A = ls.A;
      deltaVector = A.ldlt().solve(b);
      // ---------------------------------
      (void)LM_lambda;
      // A * delta = b must hold for whatever factorisation solved it
      float worst = 0.f, scale = 0.f;
      for (int r = 0; r < 6; r++) {
        float acc = 0.f;
        for (int c = 0; c < 6; c++) acc += A(r, c) * deltaVector(c);
        worst = std::fmax(worst, std::fabs(acc - b(r)));
        scale = std::fmax(scale, std::fabs(b(r)));
      }
      std::printf("DSO %d %.9g %.9g %d\n", ls.num_constraints, ls.A(2, 3), ls.b(4), worst <= 1e-3f * scale ? 1 : 0);
    }
    // Tracker::Mat2SE3 (include/Tracker.h:178) and Tracker::AddPatchPointsFeatures (:126)
    {
      const float in[6] = {0.02f, -0.01f, 0.03f, 0.5f, -0.25f, 0.125f};
      const SE3 T = tracker_->Mat2SE3(in);
      std::printf("MAT2SE3 %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", T.q[0], T.q[1], T.q[2], T.q[3], T.t[0], T.t[1], T.t[2]);
      const std::vector<float> tab = {3.4f, 2.6f, 0.7f, 1.0f, 0.2f, 0.4f, 1.5f, 1.0f, 40.5f, 20.5f, 1.1f, 1.0f};
      const std::vector<float> ext = tracker_->AddPatchPointsFeatures(tab, 1);
      double sum = 0;
      for (float v : ext) sum += v;
      std::printf("ADDPATCH %d %.9g\n", (int)(ext.size() / 4), sum);
    }
    LS ls(tracker_->ctx());
    // LS::updateSSE: two calls of four points each, component-major operands (include/LeastSquares.h:42)
    ls.initialize(0);
    for (int call = 0; call < 2; call++) {
      f4 Jc[6], res, wgt;
      for (int p = 0; p < 4; p++) {
        const int q = call * 4 + p;
        for (int k = 0; k < 6; k++) Jc[k].v[p] = 0.25f * (float)((q * 7 + k * 3) % 11) - 1.0f;
        res.v[p] = (float)(q % 5) - 2.0f;
        wgt.v[p] = 0.5f + 0.125f * (float)(q % 3);
      }
      ls.updateSSE(Jc[0], Jc[1], Jc[2], Jc[3], Jc[4], Jc[5], res, wgt);
    }
    ls.finish();
    std::printf("LSSSE %.9g %.9g %.9g %.9g %.9g %d\n", ls.A(0, 0), ls.A(1, 4), ls.A(5, 5), ls.b(3), ls.error, ls.num_constraints);
    // FastEstimatePose: the 4 -> 0 / 50 iterations / gain 50 schedule (include/Tracker.h:124)
    tracker_->FastEstimatePose(previous_frame_.get(), current_frame_.get());
    print_pose("FAST", previous_frame_->rigid_transformation_, tracker_->last_stats().iterations);
    // four more frames through a four-slot tracker: the first two lose their slots and are told so
    std::vector<std::unique_ptr<Frame>> more;
    for (int i = 0; i < 4; i++) {
      more.emplace_back(AddFrame(2 + i, (i & 1) ? a : b, w, h, false));
      tracker_->ApplyGradient(more.back().get());
    }
    const int slot_after = previous_frame_->slot_;
    int thrown = 0;
    try {
      tracker_->EstimatePose(previous_frame_.get(), current_frame_.get());   // re-binds, but its gradients are gone
    } catch (const std::exception&) {
      thrown = 1;
    }
    tracker_->ApplyGradient(previous_frame_.get());
    tracker_->EstimatePose(previous_frame_.get(), current_frame_.get());
    const SE3& again = previous_frame_->rigid_transformation_;
    std::printf("EVICT %d %d %d\n", slot_after, thrown, std::memcmp(&again, &first, sizeof(SE3)) == 0 ? 1 : 0);
    more.clear();              // frames that go away give their slots back
    previous_frame_.reset();
    current_frame_.reset();
    delete tracker_;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
