// Compile-only check of include/uw_tracker.hpp's UW_WITH_OPENCV and UW_WITH_EIGEN branches against the interface stubs in
// tests/cpp/stubs/ (this image has neither library): the cv::Mat overloads and the Eigen typedefs resolve, and the reference's
// commented DSO-way block (src/Tracker.cpp:537-550) reads against Eigen-typed LS members.  Never linked, never run.
#define UW_WITH_OPENCV
#define UW_WITH_EIGEN
#include "uw_tracker.hpp"

using namespace uw;

void optional_branches(cv::Mat& K, cv::Mat& image, cv::Mat& xi, Frame* previous_frame_, Frame* current_frame_) {
  Tracker* tracker_ = new Tracker(false);
  tracker_->InitializePyramid(640, 480, K);            // include/Tracker.h:112 with the reference's cv::Mat
  previous_frame_->images_[0] = image;                 // cv::Mat -> ImageView
  current_frame_->images_[0] = ImageView(image);
  tracker_->ApplyGradient(previous_frame_);
  tracker_->EstimatePose(previous_frame_, current_frame_);
  const SE3 T = tracker_->Mat2SE3(xi);                 // include/Tracker.h:178 with a 6 x 1 cv::Mat
  (void)T;
  // LS with the reference's Eigen types (include/LeastSquares.h:34-40)
  LS ls;
  ls.initialize(4);
  Mat61f jacobian;
  for (int k = 0; k < 6; k++) jacobian(k) = (float)k;
  ls.update(jacobian, 1.0f, 0.5f);
  ls.finish();
  Mat61f b = -ls.b;
  Mat66f A = ls.A;
  Mat61f deltaVector = A.ldlt().solve(b);
  (void)deltaVector(0);
  delete tracker_;
}
