// libuwt_gen.so — the benchmarks' input generator (uwt_gen.h) behind a C entry point, so that bench.py generates the very
// inputs tools/uwt_bench generates.  Host-side test / bench input only; not part of the product library.
#include "uwt_gen.h"

extern "C" {
// one synthetic pair: ref / tgt are w*h bytes, depth (optional) w*h u16; returns the plane depth through z_out
int uwt_gen_pair(int w, int h, double fx, double fy, double cx, double cy, int gid, uint8_t* ref, uint8_t* tgt, uint16_t* depth_or_null,
                 double* z_out) {
  if (w < 1 || h < 1 || !ref || !tgt) return 1;
  std::vector<uint8_t> r, t;
  std::vector<uint16_t> d;
  uwt_gen::gen_pair(w, h, fx, fy, cx, cy, gid, depth_or_null != nullptr, r, t, d);
  std::memcpy(ref, r.data(), r.size());
  std::memcpy(tgt, t.data(), t.size());
  if (depth_or_null) std::memcpy(depth_or_null, d.data(), d.size() * 2);
  if (z_out) *z_out = uwt_gen::plane_depth(gid);
  return 0;
}
}
