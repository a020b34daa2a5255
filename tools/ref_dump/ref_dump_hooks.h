// ref_dump_hooks.h — what the instrumented copy of the reference's src/Tracker.cpp calls (tools/ref_dump/instrument.py
// inserts the UW_REF_DUMP_* lines; nothing else of that file changes).  Header-only; the dump goes to the FILE* the
// driver (ref_dump.cpp) opens with uw_ref_dump_open().  Floats are written as C99 hexadecimal literals (%a): exact.
//
// Text format, one record per line:
//   case <name>
//   eval <lvl> <k> <num_valid> <sum_r2> <error>          Tracker.cpp:493-502 (raw residuals, before the x50 gain)
//   exit                                                  the termination test fired at this evaluation (:508)
//   solve A <36> b <6> delta <6>                          :560-564 (row-major A); delta is the reference's own "A.inv() * b"
//   solve2 delta <6>                                      the same A and b through TWO statements, "Mat Ai = A.inv(); Mat d = Ai * b;"
//                                                         — the inverse is formed and multiplied, the MatExpr algebra cannot fold it
//                                                         into cv::solve: next to `solve` it shows which of the two the build runs
//   foldprobe row <r20 r21 r22> pt <x y z> lo <f> hi <f> out <f>   (written by the driver, once per case) row 2 of the test pose's
//                                                         rigid matrix as the build computed it, the point fold_probe.h built for
//                                                         it, and WarpFunction's z for that point: `lo` = "s0 += s1 + s2 + s3",
//                                                         `hi` = a left-to-right sum of the four partial sums
//   pose <qx qy qz qw tx ty tz>                           :574, after the update
//   final <qx qy qz qw tx ty tz>                          :595, previous_frame->rigid_transformation_
#pragma once
#include <cstdio>
#include <opencv2/core.hpp>

namespace uw_ref_dump {
inline FILE*& out() { static FILE* f = nullptr; return f; }
inline void open(const char* path) { out() = std::fopen(path, "w"); }
inline void close() { if (out()) std::fclose(out()); out() = nullptr; }
inline void begin_case(const char* name) { if (out()) std::fprintf(out(), "case %s\n", name); }

inline void floats(const cv::Mat& m) {   // CV_32F, any shape, row-major
  for (int r = 0; r < m.rows; r++)
    for (int c = 0; c < m.cols; c++) std::fprintf(out(), " %a", (double)m.at<float>(r, c));
}
inline void eval(int lvl, int k, int num_valid, const cv::Mat& residuals, float error) {
  if (!out()) return;
  const double s = residuals.rows ? residuals.dot(residuals) : 0.0;   // integers: exact in double
  std::fprintf(out(), "eval %d %d %d %.17g %a\n", lvl, k, num_valid, s, (double)error);
}
inline void exit_fired() { if (out()) std::fprintf(out(), "exit\n"); }
inline void solve(const cv::Mat& A, const cv::Mat& b, const cv::Mat& delta) {
  if (!out()) return;
  std::fprintf(out(), "solve A"); floats(A);
  std::fprintf(out(), " b"); floats(b);
  std::fprintf(out(), " delta"); floats(delta);
  std::fprintf(out(), "\n");
  cv::Mat Ai = A.inv();        // (evaluated: cv::invert, DECOMP_LU)
  cv::Mat d2 = Ai * b;         // (a plain gemm)
  std::fprintf(out(), "solve2 delta"); floats(d2);
  std::fprintf(out(), "\n");
}
template <typename ProbeT>
inline void fold_probe_line(float r0, float r1, float r2, const ProbeT& fp, float out_z) {
  if (!out()) return;
  std::fprintf(out(), "foldprobe row %a %a %a pt %a %a %a lo %a hi %a out %a\n", (double)r0, (double)r1, (double)r2, (double)fp.x,
               (double)fp.y, (double)fp.z, (double)fp.lo, (double)fp.hi, (double)out_z);
}
template <typename SE3T>
inline void pose_line(const char* tag, const SE3T& T) {
  if (!out()) return;
  const auto q = T.unit_quaternion();
  const auto t = T.translation();
  std::fprintf(out(), "%s %a %a %a %a %a %a %a\n", tag, (double)q.x(), (double)q.y(), (double)q.z(), (double)q.w(),
               (double)t(0), (double)t(1), (double)t(2));
}
}  // namespace uw_ref_dump

#define UW_REF_DUMP_EVAL(lvl, k, nv, residuals, err) uw_ref_dump::eval((lvl), (k), (nv), (residuals), (err))
#define UW_REF_DUMP_EXIT() uw_ref_dump::exit_fired()
#define UW_REF_DUMP_SOLVE(A, b, d) uw_ref_dump::solve((A), (b), (d))
#define UW_REF_DUMP_POSE(T) uw_ref_dump::pose_line("pose", (T))
#define UW_REF_DUMP_FINAL(T) uw_ref_dump::pose_line("final", (T))
