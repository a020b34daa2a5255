// TEST STUB — not the reference's header.  Declares, from the reference's public interface (include/System.h:63-103,
// include/Tracker.h:97-170, include/Options.h:128-153), only what tools/ref_dump/ref_dump.cpp touches, so that the driver goes
// through a compiler's parser and overload resolution here (tests/test_ref_vectors.py); the real build is against the
// reference's own headers with OpenCV, Eigen and Sophus.
#pragma once
#include <string>
#include <vector>

namespace cv {
struct Size {};
class Mat {
 public:
  unsigned char* data = nullptr;
  int rows = 0, cols = 0;
  bool isContinuous() const { return true; }
  Mat clone() const { return *this; }
  size_t total() const { return (size_t)rows * cols; }
  size_t elemSize() const { return 1; }
  template <typename T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[(size_t)r * cols + c]; }
  double dot(const Mat&) const { return 0.0; }
  Mat inv() const { return *this; }
};
inline Mat operator*(const Mat& a, const Mat&) { return a; }
template <typename T>
struct Mat_ : Mat {
  Mat_(int, int) {}
  Mat_& operator<<(T) { return *this; }
  Mat_& operator,(T) { return *this; }
};
inline Mat imread(const std::string&, int) { return Mat(); }
inline void resize(const Mat&, Mat&, Size, double, double) {}
}  // namespace cv
#define CV_LOAD_IMAGE_GRAYSCALE 0

namespace Sophus {
template <typename T, int N>
struct Vector {
  T v[N];
  Vector& operator<<(T) { return *this; }
  Vector& operator,(T) { return *this; }
};
struct Quat { float x() const { return 0; } float y() const { return 0; } float z() const { return 0; } float w() const { return 1; } };
struct Vec3 { float operator()(int) const { return 0; } };
struct EigenQuat { EigenQuat(float, float, float, float) {} };
struct EigenVec3 { static EigenVec3 Zero() { return EigenVec3(); } };
struct Mat4 { float operator()(int, int) const { return 0; } };
struct SE3f {
  static const int DoF = 6;
  Mat4 matrix() const { return Mat4(); }
  SE3f() {}
  SE3f(const EigenQuat&, const EigenVec3&) {}
  static SE3f exp(const Vector<float, 6>&) { return SE3f(); }
  Quat unit_quaternion() const { return Quat(); }
  Vec3 translation() const { return Vec3(); }
};
}  // namespace Sophus

namespace uw {
typedef Sophus::SE3f SE3;
typedef Sophus::EigenQuat Quaternion;   // include/Options.h:135 (Eigen::Quaternion<float>: w, x, y, z)
typedef Sophus::EigenVec3 Mat31f;       // include/Options.h:144
typedef Sophus::Mat4 Mat44f;            // include/Options.h (Eigen::Matrix<float,4,4>)
extern const int PYRAMID_LEVELS;
class Frame {
 public:
  std::vector<cv::Mat> images_ = std::vector<cv::Mat>(5), depths_ = std::vector<cv::Mat>(5), gradientX_ = std::vector<cv::Mat>(5),
                       gradientY_ = std::vector<cv::Mat>(5), candidatePoints_ = std::vector<cv::Mat>(5);
  bool depth_available_ = false;
};
class Tracker {
 public:
  explicit Tracker(bool) {}
  void InitializePyramid(int, int, cv::Mat) {}
  void InitializeMasks() {}
  void ApplyGradient(Frame*) {}
  void ObtainAllPoints(Frame*) {}
  cv::Mat WarpFunction(cv::Mat, SE3, int) { return cv::Mat(); }
  void EstimatePose(Frame*, Frame*) {}
};
}  // namespace uw
