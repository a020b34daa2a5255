// Drives the C++ shim (include/uw_tracker.hpp) through System::Tracking()'s call sequence (src/System.cpp:193-223)
// on two frames read from a raw file:  <w> <h> then w*h bytes (previous) and w*h bytes (current).
// Prints the resulting pose (qx qy qz qw tx ty tz) with %.9g and the iteration count.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "uw_tracker.hpp"

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
    uw::Tracker* tracker_ = new uw::Tracker(false);
    tracker_->InitializePyramid(w, h, K);
    tracker_->InitializeMasks();
    uw::Frame previous_frame_, current_frame_;
    previous_frame_.image0_ = uw::ImageView(a.data(), h, w, (size_t)w);
    current_frame_.image0_ = uw::ImageView(b.data(), h, w, (size_t)w);
    tracker_->ApplyGradient(&previous_frame_);
    tracker_->ApplyGradient(&current_frame_);
    tracker_->ObtainAllPoints(&previous_frame_);
    tracker_->EstimatePose(&previous_frame_, &current_frame_);
    const uw::SE3& T = previous_frame_.rigid_transformation_;
    std::printf("%.9g %.9g %.9g %.9g %.9g %.9g %.9g %d\n", T.q[0], T.q[1], T.q[2], T.q[3], T.t[0], T.t[1], T.t[2],
                tracker_->last_stats().iterations);
    // the reference's live flow (src/System.cpp:193-223): key points -> ObtainPatchesPoints -> EstimatePoseFeatures
    for (int k = 0; k < 40; k++) {
      previous_frame_.keypoints_.push_back(8.0f + (float)((k * 37) % (w - 16)));
      previous_frame_.keypoints_.push_back(8.0f + (float)((k * 23) % (h - 16)));
    }
    tracker_->ObtainPatchesPoints(&previous_frame_);
    tracker_->EstimatePoseFeatures(&previous_frame_, &current_frame_);
    const uw::SE3& F = previous_frame_.rigid_transformation_;
    std::printf("FEATURES %.9g %.9g %.9g %.9g %.9g %.9g %.9g %d %d\n", F.q[0], F.q[1], F.q[2], F.q[3], F.t[0], F.t[1], F.t[2],
                tracker_->last_stats().iterations, (int)(previous_frame_.candidatePoints_[0].size() / 4));
    // LS mirror: one row, closed form A = (J J^T) w, b = -w r J
    uw::LS ls(tracker_->ctx());
    const float J[6] = {1, 2, 3, 4, 5, 6};
    ls.update(J, 2.0f, 0.5f);
    ls.finishNoDivide();
    std::printf("LS %.9g %.9g %.9g %d\n", ls.A[0 * 6 + 1], ls.b[2], ls.error, ls.num_constraints);
    delete tracker_;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
