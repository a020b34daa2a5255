// TEST STUB — cv::Mat for tools/ref_dump/ref_dump_hooks.h comes from the System.h stub next to this directory.
#pragma once
#include "../System.h"
namespace cv {
inline double mat_dot_stub(const Mat&, const Mat&) { return 0.0; }
}
