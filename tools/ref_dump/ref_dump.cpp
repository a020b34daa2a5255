// ref_dump.cpp — driver a maintainer compiles AGAINST THE REAL REFERENCE (MecatronicaUSB/uw-slam with its OpenCV 3.2 + Eigen +
// Sophus build) to produce golden vectors for this repository's parity tests.  It is not built by this repository: the
// image this framework is developed in has neither OpenCV nor Eigen; here the file only passes a syntax check against interface
// stubs (tests/cpp/stubs/ref/).
//
// For every case directory written by export_inputs.py it reproduces what System does for two frames and dumps what the
// tracker computes, through the reference's own classes only:
//   System::AddFrame's pyramid loop              (src/System.cpp:246-251)      -> <case>_img<l>.u8, <case>_dep<l>.u16
//   Tracker::ApplyGradient                       (src/Tracker.cpp:1127-1176)   -> <case>_gx<l>.i16, <case>_gy<l>.i16
//   Tracker::ObtainAllPoints                     (src/Tracker.cpp:1259-1310)   -> <case>_pts<l>.f32 (N x 4)
//   Tracker::WarpFunction at a fixed test pose   (src/Tracker.cpp:1417-1471)   -> <case>_warp<l>.f32 (N x 4)
//   Tracker::WarpFunction at the two cyclic axis permutations q = (-+1/2, -+1/2, -+1/2, 1/2), t = 0: every entry of the rigid
//   matrix is exactly 0 or 1, so whatever cv::gemm accumulates in, column 2 of the result IS the unprojected X * z (resp.
//   Y * z) of Tracker.cpp:1439-1444, bit for bit: the quantity that tells the folded scaled convert x * invfx + beta from
//   (x - cx) * invfx                                                              -> <case>_unpx<l>.f32, <case>_unpy<l>.f32
//   Tracker::WarpFunction on ONE constructed point (fold_probe.h) through a second Tracker with fx = fy = 1, cx = cy = 0: its
//   warped z says how this build's gemm folds the four partial sums of the rigid product ("s0 += s1 + s2 + s3" against a
//   left-to-right sum) — one value in 10^9 differs, so the dense tables above cannot tell     -> dump.txt record `foldprobe`
//   Tracker::EstimatePose                        (src/Tracker.cpp:362-597)     -> dump.txt records (ref_dump_hooks.h) written
//                                                                                 by the instrumented Tracker_refdump.cpp
// Usage:  ref_dump <inputs_dir> <out_dir>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "System.h"           // the reference's: uw::Frame (include/System.h:63-103), PYRAMID_LEVELS
#include "Tracker.h"          // the reference's: uw::Tracker (include/Tracker.h:97-235)
#include "fold_probe.h"
#include "ref_dump_hooks.h"

using namespace uw;

static void write_raw(const std::string& path, const cv::Mat& m) {
  cv::Mat c = m.isContinuous() ? m : m.clone();
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(c.data), (std::streamsize)(c.total() * c.elemSize()));
}

static Frame* make_frame(const std::string& gray_path, const std::string& depth_path, bool depth) {
  Frame* f = new Frame();
  f->images_[0] = cv::imread(gray_path, CV_LOAD_IMAGE_GRAYSCALE);                  // src/System.cpp:228
  if (depth) {
    f->depth_available_ = true;                                                    // :242
    f->depths_[0] = cv::imread(depth_path, -1);                                    // :243
  }
  for (int i = 1; i < PYRAMID_LEVELS; i++) {                                       // :246-251
    cv::resize(f->images_[i - 1], f->images_[i], cv::Size(), 0.5, 0.5);
    if (depth) cv::resize(f->depths_[i - 1], f->depths_[i], cv::Size(), 0.5, 0.5);
  }
  return f;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: ref_dump <inputs_dir> <out_dir>\n"); return 2; }
  const std::string in = argv[1], out = argv[2];
  std::ifstream list(in + "/cases.txt");
  if (!list) { std::fprintf(stderr, "no cases.txt in %s (run export_inputs.py)\n", in.c_str()); return 2; }
  uw_ref_dump::open((out + "/dump.txt").c_str());
  std::string name;
  int w, h, has_depth;
  float fx, fy, cx, cy;
  while (list >> name >> w >> h >> fx >> fy >> cx >> cy >> has_depth) {
    const std::string dir = in + "/" + name, pre = out + "/" + name;
    Tracker* tracker = new Tracker(has_depth != 0);                                // src/System.cpp:121
    cv::Mat K = (cv::Mat_<float>(3, 3) << fx, 0, cx, 0, fy, cy, 0, 0, 1);
    tracker->InitializePyramid(w, h, K);                                           // :122
    tracker->InitializeMasks();                                                    // :123
    Frame* prev = make_frame(dir + "/ref.png", dir + "/depth.png", has_depth != 0);
    Frame* cur = make_frame(dir + "/tgt.png", dir + "/depth.png", has_depth != 0);
    tracker->ApplyGradient(prev);                                                  // src/System.cpp:197-213
    tracker->ObtainAllPoints(prev);
    tracker->ApplyGradient(cur);
    tracker->ObtainAllPoints(cur);
    // a fixed small test pose for WarpFunction: exp of (0.01, -0.02, 0.015, 0.004, -0.003, 0.002)
    Sophus::Vector<float, SE3::DoF> xi;
    xi << 0.01f, -0.02f, 0.015f, 0.004f, -0.003f, 0.002f;
    const SE3 test_pose = SE3::exp(xi);
    // rotations by 120 degrees about (1,1,1) and back: R = [[0,1,0],[0,0,1],[1,0,0]] (row 2 picks X) and its transpose
    // (row 2 picks Y); unit quaternions with exactly representable coefficients (Eigen order: w, x, y, z)
    const SE3 pick_x(Quaternion(0.5f, -0.5f, -0.5f, -0.5f), Mat31f::Zero());
    const SE3 pick_y(Quaternion(0.5f, 0.5f, 0.5f, 0.5f), Mat31f::Zero());
    for (int l = 0; l < PYRAMID_LEVELS; l++) {
      const std::string s = std::to_string(l);
      write_raw(pre + "_img" + s + ".u8", prev->images_[l]);
      write_raw(pre + "_tgt" + s + ".u8", cur->images_[l]);
      if (has_depth) write_raw(pre + "_dep" + s + ".u16", prev->depths_[l]);
      write_raw(pre + "_gx" + s + ".i16", prev->gradientX_[l]);
      write_raw(pre + "_gy" + s + ".i16", prev->gradientY_[l]);
      write_raw(pre + "_pts" + s + ".f32", prev->candidatePoints_[l]);
      write_raw(pre + "_warp" + s + ".f32", tracker->WarpFunction(prev->candidatePoints_[l], test_pose, l));
      write_raw(pre + "_unpx" + s + ".f32", tracker->WarpFunction(prev->candidatePoints_[l], pick_x, l));
      write_raw(pre + "_unpy" + s + ".f32", tracker->WarpFunction(prev->candidatePoints_[l], pick_y, l));
    }
    uw_ref_dump::begin_case(name.c_str());
    uw_ref_dump::pose_line("testpose", test_pose);
    {  // the fold probe: built from the matrix WarpFunction itself will see (SE3::matrix(), src/Tracker.cpp:1423)
      const Mat44f Tm = test_pose.matrix();
      const uw_ref_dump::FoldProbe fp = uw_ref_dump::find_fold_probe(Tm(2, 0), Tm(2, 1), Tm(2, 2));
      if (fp.found) {
        Tracker* unit = new Tracker(false);
        cv::Mat K1 = (cv::Mat_<float>(3, 3) << 1, 0, 0, 0, 1, 0, 0, 0, 1);
        unit->InitializePyramid(64, 64, K1);
        cv::Mat P = (cv::Mat_<float>(1, 4) << fp.x, fp.y, fp.z, 0.f);
        const cv::Mat Wp = unit->WarpFunction(P, test_pose, 0);
        uw_ref_dump::fold_probe_line(Tm(2, 0), Tm(2, 1), Tm(2, 2), fp, Wp.at<float>(0, 2));
        delete unit;
      }
    }
    tracker->EstimatePose(prev, cur);                                              // src/System.cpp:214-223
    delete tracker;
  }
  uw_ref_dump::close();
  return 0;
}
