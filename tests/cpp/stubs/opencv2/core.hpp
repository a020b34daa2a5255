// TEST STUB — not OpenCV.  Just enough of cv::Mat's interface (the members include/uw_tracker.hpp touches under
// UW_WITH_OPENCV) for tests/test_cpp_shim.py to put that branch of the header through a compiler: the image this
// repository is developed in has no OpenCV.  It checks syntax and overload resolution, nothing else.
#pragma once
#include <cstddef>
namespace cv {
struct MatStep {
  size_t v = 0;
  operator size_t() const { return v; }
};
class Mat {
 public:
  unsigned char* data = nullptr;
  int rows = 0, cols = 0;
  MatStep step;
  template <typename T> T& at(int r, int c) { return reinterpret_cast<T*>(data + (size_t)r * step.v)[c]; }
  template <typename T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data + (size_t)r * step.v)[c]; }
};
}  // namespace cv
