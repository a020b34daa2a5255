// uwt_kernels.h — gfx950 kernels of the direct SE(3) tracking path.
//
//   k_halve, k_pyramid_all, k_pyramid_batch   System::AddFrame pyramid loop (a launch per level / every level of a few frames in
//                               one / levels 1..3 of a batch in one pass over level 0)   (src/System.cpp:246-251)
//   k_scharr3*, k_scharr3_levels   Tracker::ApplyGradient (per level / every level of a few frames in one launch)
//                               (src/Tracker.cpp:1133-1134)
//   k_residual     WarpFunction + per-point loop + the 28-accumulator LS reduction, fused
//                  (src/Tracker.cpp:1417-1471, 432-490; src/LeastSquares.cpp:148-209)
//   k_gn_update    error / exit test / normal equations / solve / pose update (src/Tracker.cpp:495-574);
//                  tail_update_wave: the same in the tail of the residual launch (the pair's last block, by ticket)
//   k_level_end    level hand-off                           (src/Tracker.cpp:580-590)
//   k_iterate, k_finish   the same loop chained: update + hand-off at the head of the next evaluation (a few pairs per call)
//   k_coarse       the coarsest levels of such a call — and the coarsest level of a batch — run to their end in one launch,
//                  one block per pair
//   k_resid_hist_v (scale pass with the scale stage in its tail) + k_residual<.., WEIGHTS>   robust weights in the alignment loop;
//                  k_residual<.., SAMPLER = 1>: bilinear sampler
//   k_residual_points, k_points_hist, k_points_general                   explicit point tables (identity / general path)
//   k_ls_sequential   the LS mirror (src/LeastSquares.cpp): every accumulator's f32 chain in the reference's order
//   k_residual_general, k_resid_hist, k_scale_stage                      per-stage (dump) forms of the general path
//   k_grad_mag*, k_candidates_batch, k_scan_counts, k_patch_points, k_add_patch_points, k_remap_crop, k_trajectory*   the rows
//                  next to the path
//   masked_sums_*  a pixel's 28 f64 sums under an EXEC mask of the valid lanes (no select anywhere in the loop)
//   load_group_typed   the production loop's plane loads as typed buffer loads (the texture path converts int16 -> f32), every
//                  vector-memory operation of that loop hand-written with its waits (tools/check_asm_loads.py)
// The heavy templates are instantiated by the dispatchers of uwt_launch_{residual,general,flow}.hip (uwt_launch.h); the kernels
// that are not templates are `static`: each translation unit that launches one carries its own copy.
//
// Stencil + gather + reduction work with a 6-wide contraction: no MFMA.  Wave = 64 lanes, blocks of 256.
// Build with -ffp-contract=off: the per-pixel float sequence is part of the contract (every FMA is explicit).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "uwt_math.h"

namespace uwt {

constexpr int kBlock = 256;
constexpr int kAccFloats = 27;   // 21 upper-triangle JᵀJ + 6 Jᵀr
constexpr int kRecWords = 64;    // one partial record (256 B): 27 f64 | n_valid u32 @ word 54 | Σr² u64 @ words 56-57
constexpr int kMaxSlices = 160;  // slices per pair and level at most (a 1280x960 level 0 has 150 at 2 groups per thread)

// Arithmetic set of the third-party (OpenCV) steps on the path, a template parameter of every kernel that warps a point or
// forms a Jacobian row (uwt_params::arith; include/uwt.h has the published-algorithm citations):
//   kArithOpenCV  what OpenCV 3.x's generic code paths compute: cv::gemm products of CV_32F data accumulated in double and
//                 rounded to float once (GEMMSingleMul<float,double>: the 4-term rigid * points.t() of src/Tracker.cpp:1450,
//                 the 2-term Jl * Jw of :479), "(col - cx) * invfx" (:1439) folded by the MatExpr algebra into one scaled
//                 convert x * invfx + (float)(-cx * invfx), "A.inv() * b" (:564) solved by LU on the right-hand side.
//   kArithLegacy  rounds 1-3: the small products as k-sequential f32 FMA chains, the unprojection as written, the inverse
//                 formed and multiplied.  Cheaper, and not what OpenCV computes; kept selectable.
constexpr int kArithOpenCV = 0, kArithLegacy = 1;

// One pyramid level.  Three sizes (equal whenever the level-0 size is divisible by 2^lvl and the row is whole groups of four):
//   image   iw x ih   images_[lvl].cols / rows: the cv::resize(.., 0.5, 0.5) chain of src/System.cpp:246-251 (cvRound per level).
//                     What the bounds test of src/Tracker.cpp:450 reads ("image2.rows / .cols"), what Scharr runs over, what a
//                     sample index is clamped to.
//   grid    gw x gh   w_[lvl] x h_[lvl] = size >> lvl (src/Tracker.cpp:312-313): the points ObtainAllPoints walks (:1267-1268).
//                     gw <= iw, gh <= ih (733 wide: level 3 is 92 wide, its grid 91).
//   pitch             elements per plane row in memory: iw rounded up to a multiple of 4, so that every row starts on a vector
//                     boundary whatever the size.  A level's pixels are walked by ONE linear index over pitch x gh positions
//                     (index = y * pitch + x = the plane offset); positions with x >= gw carry no point and are masked out.
struct LevelK {
  int pitch;
  int iw, ih;
  int gw, gh;
  int n;            // elements per frame slot of a level plane: pitch * ih
  int ng;           // length of the linear walk: pitch * gh (a multiple of 4)
  float fx, fy, cx, cy, invfx, invfy;
  float bx, by;     // (float)(-(double)cx * (double)invfx), likewise y: beta of the folded unprojection (kArithOpenCV)
  float zscale;     // depth_scale / 2^lvl (src/Tracker.cpp:1266)
  uint32_t magic;   // ceil(2^32 / pitch): idx / pitch == umulhi(idx, magic) for idx * pitch < 2^32
};

struct PairState {
  Pose pose;
  float last_error;
  float error;
  int level_done;
  int status;
  int iters;
  int n_valid;
};

// ------------------------------------------------------------------------------------------------------------
// pyramid: 2x2 mean with round-half-up == cv::resize(.., 0.5, 0.5) on u8 / u16 (src/System.cpp:247, 249)
// One thread produces VEC horizontally adjacent outputs from two 2·VEC-wide input row segments.  Whole cells only: a source of
// 2 w_out x 2 h_out pixels (k_resize_half takes every other size).  Rows of src_pitch / dst_pitch elements.
// ------------------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_halve(const T* __restrict__ src, T* __restrict__ dst, int w_out, int h_out,
                                                  int src_pitch, int dst_pitch,
                                                  size_t src_frame_stride, size_t dst_frame_stride,
                                                  const int* __restrict__ slots, int first_slot) {
  const size_t frame = slots ? slots[blockIdx.y] : first_slot + (int)blockIdx.y;  // src / dst point at slot 0
  const int groups_per_row = w_out / VEC;
  const int g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= groups_per_row * h_out) return;
  const int y = g / groups_per_row;
  const int x = (g - y * groups_per_row) * VEC;
  const T* s0 = src + frame * src_frame_stride + (size_t)(2 * y) * src_pitch + 2 * x;
  const T* s1 = s0 + src_pitch;
  T* d = dst + frame * dst_frame_stride + (size_t)y * dst_pitch + x;
  T a[2 * VEC], b[2 * VEC], o[VEC];
  if constexpr (VEC == 4 && sizeof(T) == 1) {
    *reinterpret_cast<uint2*>(a) = *reinterpret_cast<const uint2*>(s0);
    *reinterpret_cast<uint2*>(b) = *reinterpret_cast<const uint2*>(s1);
  } else if constexpr (VEC == 4 && sizeof(T) == 2) {
    *reinterpret_cast<uint4*>(a) = *reinterpret_cast<const uint4*>(s0);
    *reinterpret_cast<uint4*>(b) = *reinterpret_cast<const uint4*>(s1);
  } else {
#pragma unroll
    for (int i = 0; i < 2 * VEC; i++) { a[i] = s0[i]; b[i] = s1[i]; }
  }
#pragma unroll
  for (int i = 0; i < VEC; i++)
    o[i] = (T)(((uint32_t)a[2 * i] + (uint32_t)a[2 * i + 1] + (uint32_t)b[2 * i] + (uint32_t)b[2 * i + 1] + 2u) >> 2);
  if constexpr (VEC == 4 && sizeof(T) == 1) {
    *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<uint32_t*>(o);
  } else if constexpr (VEC == 4 && sizeof(T) == 2) {
    *reinterpret_cast<uint2*>(d) = *reinterpret_cast<uint2*>(o);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; i++) d[i] = o[i];
  }
}

// cv::resize(src, dst, Size(), 0.5, 0.5) for ANY source size (src/System.cpp:247, 249 on the ROI-cropped frames of :148-191,
// :232-236): dst is cvRound(sw / 2) x cvRound(sh / 2) (half to even: 733 -> 366, 735 -> 368).  OpenCV's resizeAreaFast for
// scale 2 x 2: whole cells (a + b + c + d + 2) >> 2; where 2 dw > sw the last column holds half cells, where 2 dh > sh the
// last row does — and resizeAreaFast sends EVERY cell of such a row, and the half cells of the others, through its generic
// tail: the mean of the source pixels that exist, saturate_cast<T>((float)sum / count) = round half to EVEN (a + b = 5 -> 2,
// the whole cells' rule would give 3).  A thread makes four adjacent outputs; the pad columns of the pitched row get zeros.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_resize_half(const T* __restrict__ src, T* __restrict__ dst, int sw, int sh, int src_pitch,
                                                        int dw, int dh, int dst_pitch, size_t src_frame_stride,
                                                        size_t dst_frame_stride, const int* __restrict__ slots, int first_slot) {
  const size_t frame = slots ? slots[blockIdx.y] : first_slot + (int)blockIdx.y;
  const int groups_per_row = dst_pitch >> 2;
  const int g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= groups_per_row * dh) return;
  const int y = g / groups_per_row;
  const int x = (g - y * groups_per_row) * 4;
  const int fw = sw >> 1, fh = sh >> 1;   // columns / rows of whole cells
  const int r0 = 2 * y, r1 = min(2 * y + 1, sh - 1);
  const T* s0 = src + frame * src_frame_stride + (size_t)r0 * src_pitch;
  const T* s1 = src + frame * src_frame_stride + (size_t)r1 * src_pitch;
  uint32_t a[8], b[8];
  if (2 * x + 8 <= src_pitch) {   // the eight source columns lie inside the pitched row (pad columns included: never used)
    T va[8], vb[8];
    if constexpr (sizeof(T) == 1) {
      *reinterpret_cast<uint2*>(va) = *reinterpret_cast<const uint2*>(s0 + 2 * x);
      *reinterpret_cast<uint2*>(vb) = *reinterpret_cast<const uint2*>(s1 + 2 * x);
    } else {
      *reinterpret_cast<uint4*>(va) = *reinterpret_cast<const uint4*>(s0 + 2 * x);
      *reinterpret_cast<uint4*>(vb) = *reinterpret_cast<const uint4*>(s1 + 2 * x);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = va[i]; b[i] = vb[i]; }
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int cx = min(2 * x + i, sw - 1);
      a[i] = s0[cx];
      b[i] = s1[cx];
    }
  }
  T o[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int xo = x + i;
    uint32_t v = 0;
    if (y < fh && xo < fw) {
      v = (a[2 * i] + a[2 * i + 1] + b[2 * i] + b[2 * i + 1] + 2u) >> 2;
    } else if (xo < dw) {   // a cell of the partial last row, or the half cell of the last column
      const bool two_cols = 2 * xo + 1 < sw, two_rows = y < fh;
      uint32_t sum = a[2 * i];
      int count = 1;
      if (two_cols) { sum += a[2 * i + 1]; count++; }
      if (two_rows) { sum += b[2 * i]; count++; if (two_cols) { sum += b[2 * i + 1]; count++; } }
      v = (uint32_t)(int)rintf((float)sum / (float)count);   // cvRound: half to even
    }
    o[i] = (T)v;
  }
  T* d = dst + frame * dst_frame_stride + (size_t)y * dst_pitch + x;
  if constexpr (sizeof(T) == 1) *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<uint32_t*>(o);
  else *reinterpret_cast<uint2*>(d) = *reinterpret_cast<uint2*>(o);
}

// The whole pyramid of a frame in one launch (a frame or a few on their own: the live per-frame sequence, where a launch
// per level and plane is all latency).  The 2x2 mean is tile-local, so a block takes a 64x64 tile of level 0 down every
// level: a thread reads its 4x4 patch, keeps levels 1 and 2 in registers, levels 3.. go through LDS (16x16 -> 8x8 -> ..).
// Needs 3 <= n_levels <= 7 and level-0 sizes divisible by 2^(n_levels-1) (whole cells on every level) and by 4; same integers
// as k_halve.  Rows of pitch[l] elements.
constexpr int kPyrMaxLevels = 7;
template <typename T>
struct PyramidArgs {
  const T* src;                 // level 0, slot 0
  T* dst[kPyrMaxLevels];        // dst[l]: level l, slot 0 (dst[0] unused)
  size_t stride[kPyrMaxLevels]; // elements per frame at level l
  int pitch[kPyrMaxLevels];     // elements per row at level l
  int w, h;                     // level 0
  int n_levels;
  const int* slots;
  int first_slot;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void k_pyramid_all(const PyramidArgs<T> a) {
  __shared__ uint32_t lv[2][16][16];
  const size_t frame = a.slots ? a.slots[blockIdx.y] : a.first_slot + (int)blockIdx.y;
  const int tiles_x = (a.w + 63) / 64;
  const int tyb = blockIdx.x / tiles_x, txb = blockIdx.x - tyb * tiles_x;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x = txb * 64 + 4 * tx, y = tyb * 64 + 4 * ty;   // level-0 corner of this thread's 4x4 patch
  uint32_t p2 = 0;
  if (x < a.w && y < a.h) {
    uint32_t px[4][4];
    const T* s = a.src + frame * a.stride[0] + (size_t)y * a.pitch[0] + x;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      T row[4];
      if constexpr (sizeof(T) == 1) *reinterpret_cast<uint32_t*>(row) = *reinterpret_cast<const uint32_t*>(s + (size_t)r * a.pitch[0]);
      else *reinterpret_cast<uint2*>(row) = *reinterpret_cast<const uint2*>(s + (size_t)r * a.pitch[0]);
#pragma unroll
      for (int c = 0; c < 4; c++) px[r][c] = row[c];
    }
    uint32_t p1[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 2; c++) p1[r][c] = (px[2 * r][2 * c] + px[2 * r][2 * c + 1] + px[2 * r + 1][2 * c] + px[2 * r + 1][2 * c + 1] + 2u) >> 2;
    const int w1 = a.pitch[1];
    T* d1 = a.dst[1] + frame * a.stride[1] + (size_t)(y >> 1) * w1 + (x >> 1);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      T o[2] = {(T)p1[r][0], (T)p1[r][1]};
      if constexpr (sizeof(T) == 1) *reinterpret_cast<uint16_t*>(d1 + (size_t)r * w1) = *reinterpret_cast<uint16_t*>(o);
      else *reinterpret_cast<uint32_t*>(d1 + (size_t)r * w1) = *reinterpret_cast<uint32_t*>(o);
    }
    p2 = (p1[0][0] + p1[0][1] + p1[1][0] + p1[1][1] + 2u) >> 2;
    a.dst[2][frame * a.stride[2] + (size_t)(y >> 2) * a.pitch[2] + (x >> 2)] = (T)p2;
  }
  if (a.n_levels <= 3) return;   // block-uniform
  lv[0][ty][tx] = p2;
  // level l (3..): 2^(6-l) x 2^(6-l) elements per tile, element (ex, ey) from the four level l-1 elements in LDS
#pragma unroll
  for (int l = 3; l < kPyrMaxLevels; l++) {
    if (l >= a.n_levels) break;   // block-uniform
    __syncthreads();
    const int side = 64 >> l;     // elements per tile side at this level: 8, 4, 2, 1
    const int b = (l - 3) & 1;
    if (tx < side && ty < side) {
      const uint32_t v = (lv[b][2 * ty][2 * tx] + lv[b][2 * ty][2 * tx + 1] + lv[b][2 * ty + 1][2 * tx] + lv[b][2 * ty + 1][2 * tx + 1] + 2u) >> 2;
      lv[b ^ 1][ty][tx] = v;
      const int ex = txb * side + tx, ey = tyb * side + ty, wl = a.w >> l, hl = a.h >> l;
      if (ex < wl && ey < hl) a.dst[l][frame * a.stride[l] + (size_t)ey * a.pitch[l] + ex] = (T)v;
    }
  }
}

// Levels 1..3 of a BATCH of frames in one pass over level 0 (round 3): the per-level k_halve chain reads level 0, level 1
// and level 2 again (1.64 bytes moved per level-0 byte); here a block takes a 128 x 64 tile of level 0 down three levels,
// reads it once with 16-byte loads and writes the three results (1.33).  Thread (tx = t mod 8, ty = t div 8) loads 16
// pixels of rows 2 ty, 2 ty + 1 and makes 8 pixels of level-1 row ty; the level-1 row below belongs to the lane 8 on, the
// level-2 row below to the lane 16 on (same wave: a wave holds ty = 8 k .. 8 k + 7), so levels 2 and 3 come out of two
// lane exchanges — no LDS, no barrier.  Same integers as k_halve level by level ((a + b + c + d + 2) >> 2 of the rounded
// values of the level above).  Needs the level-0 width divisible by 16 and the height by 8; frames by slot range or list.
template <typename T>
struct PyramidBatchArgs {
  const T* src;     // level 0, slot 0
  T* dst[3];        // levels 1, 2, 3, slot 0
  size_t stride[4]; // elements per frame at levels 0..3
  int pitch[4];     // elements per row at levels 0..3 (pitch[0] == w: the level-0 width is a multiple of 16)
  int w, h;         // level 0
  const int* slots;
  int first_slot;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void k_pyramid_batch(const PyramidBatchArgs<T> a) {
  const size_t frame = a.slots ? a.slots[blockIdx.y] : a.first_slot + (int)blockIdx.y;
  const int tiles_x = (a.w + 127) / 128;
  const int tyb = blockIdx.x / tiles_x, txb = blockIdx.x - tyb * tiles_x;
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int x = txb * 128 + 16 * tx, y = tyb * 64 + 2 * ty;   // level-0 corner of this thread's 16 x 2 patch
  const bool in = x < a.w && y < a.h;
  uint32_t p1[8];   // level-1 pixels (x / 2 .. x / 2 + 7, y / 2)
  {
    T r0[16], r1[16];
    const uint32_t off = in ? (uint32_t)y * (uint32_t)a.w + (uint32_t)x : 0u;   // (lanes outside the image read the frame's first pixels)
    const unsigned char* s = reinterpret_cast<const unsigned char*>(a.src + frame * a.stride[0]);
    if constexpr (sizeof(T) == 1) {
      *reinterpret_cast<uint4*>(r0) = *reinterpret_cast<const uint4*>(s + off);
      *reinterpret_cast<uint4*>(r1) = *reinterpret_cast<const uint4*>(s + off + (uint32_t)a.w);
    } else {
      const uint32_t o2 = off * 2u, w2 = (uint32_t)a.w * 2u;
      *reinterpret_cast<uint4*>(r0) = *reinterpret_cast<const uint4*>(s + o2);
      *reinterpret_cast<uint4*>(r0 + 8) = *reinterpret_cast<const uint4*>(s + o2 + 16u);
      *reinterpret_cast<uint4*>(r1) = *reinterpret_cast<const uint4*>(s + o2 + w2);
      *reinterpret_cast<uint4*>(r1 + 8) = *reinterpret_cast<const uint4*>(s + o2 + w2 + 16u);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) p1[i] = ((uint32_t)r0[2 * i] + (uint32_t)r0[2 * i + 1] + (uint32_t)r1[2 * i] + (uint32_t)r1[2 * i + 1] + 2u) >> 2;
  }
  const int w1 = a.pitch[1], w2 = a.pitch[2], w3 = a.pitch[3];
  if (in) {
    T o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = (T)p1[i];
    T* d = a.dst[0] + frame * a.stride[1] + (size_t)(y >> 1) * w1 + (x >> 1);
    if constexpr (sizeof(T) == 1) *reinterpret_cast<uint2*>(d) = *reinterpret_cast<uint2*>(o);
    else *reinterpret_cast<uint4*>(d) = *reinterpret_cast<uint4*>(o);
  }
  // level 2: level-1 rows ty (this lane, ty even) and ty + 1 (the lane 8 on)
  uint32_t p2[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t below0 = (uint32_t)__shfl_xor((int)p1[2 * i], 8), below1 = (uint32_t)__shfl_xor((int)p1[2 * i + 1], 8);
    p2[i] = (p1[2 * i] + p1[2 * i + 1] + below0 + below1 + 2u) >> 2;
  }
  if (in && (ty & 1) == 0) {
    T o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (T)p2[i];
    T* d = a.dst[1] + frame * a.stride[2] + (size_t)(y >> 2) * w2 + (x >> 2);
    if constexpr (sizeof(T) == 1) *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<uint32_t*>(o);
    else *reinterpret_cast<uint2*>(d) = *reinterpret_cast<uint2*>(o);
  }
  // level 3: level-2 rows ty / 2 (this lane, ty a multiple of 4) and ty / 2 + 1 (the lane 16 on)
  uint32_t p3[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const uint32_t below0 = (uint32_t)__shfl_xor((int)p2[2 * i], 16), below1 = (uint32_t)__shfl_xor((int)p2[2 * i + 1], 16);
    p3[i] = (p2[2 * i] + p2[2 * i + 1] + below0 + below1 + 2u) >> 2;
  }
  if (in && (ty & 3) == 0) {
    T o[2] = {(T)p3[0], (T)p3[1]};
    T* d = a.dst[2] + frame * a.stride[3] + (size_t)(y >> 3) * w3 + (x >> 3);
    if constexpr (sizeof(T) == 1) *reinterpret_cast<uint16_t*>(d) = *reinterpret_cast<uint16_t*>(o);
    else *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<uint32_t*>(o);
  }
}

// ------------------------------------------------------------------------------------------------------------
// gradients: 3 x Scharr, reflect-101 border, exact in int (src/Tracker.cpp:1133-1134).
// ------------------------------------------------------------------------------------------------------------

__device__ inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// Vector variant (any width: rows are pitched to whole words): a 128 x (8·RPT) output tile per block, the source patch (tile
// + 1-pixel ring) staged in LDS with 4-byte loads; a thread owns 4 adjacent columns and RPT consecutive rows, slides a
// three-row window of LDS words down them and stores two 8-byte vectors per row.  HBM-bound: 1 B read + 4 B written per
// pixel.  RPT = 4 on levels tall enough: fewer, longer blocks with 4-5 loads in flight per thread before the barrier.
// `slots` (optional) lists the frame slots to process.
constexpr int kGradVW = 128, kGradVRows = 8;  // columns per tile; thread rows per tile (x RPT output rows each)

// one 128 x (8·RPT) output tile; lds: scharr_v4_lds_rows(RPT) x (kGradVW / 4 + 2) words (the patch's 8·RPT + 2 rows rounded up
// to whole staging rounds of 8)
constexpr int scharr_v4_lds_rows(int rpt) { return (kGradVRows * rpt + 2 + kGradVRows - 1) / kGradVRows * kGradVRows; }
template <int RPT>
__device__ __forceinline__ void scharr_tile_v4(unsigned char* __restrict__ lds, const uint8_t* __restrict__ src,
                                               int16_t* __restrict__ gx, int16_t* __restrict__ gy, int w, int h, int pitch,
                                               size_t frame_stride, int slot, int tile_id) {
  constexpr int TH = kGradVRows * RPT, WPR = kGradVW / 4;
  uint32_t(*tile)[WPR + 2] = reinterpret_cast<uint32_t(*)[WPR + 2]>(lds);  // word 0: left halo in its top byte; word 33: right halo in its low byte
  const int tiles_x = (w + kGradVW - 1) / kGradVW;
  const int ty = tile_id / tiles_x, tx = tile_id - ty * tiles_x;
  const int x0 = tx * kGradVW, y0 = ty * TH;
  // the frame's planes: uniform bases, 32-bit offsets inside a frame (global_load / global_store with a scalar base)
  const unsigned char* img = src + (size_t)slot * frame_stride;
  unsigned char* gxf = reinterpret_cast<unsigned char*>(gx + (size_t)slot * frame_stride);
  unsigned char* gyf = reinterpret_cast<unsigned char*>(gy + (size_t)slot * frame_stride);
  const int tw = min(kGradVW, w - x0);       // image columns of this tile
  const int twp = min(kGradVW, pitch - x0);  // ... rounded up to whole words: the pitched row's pad columns ride along (never read back)
  const int rows = min(TH, h - y0) + 2;      // patch rows this tile needs
  const int ly = threadIdx.x / WPR, c = threadIdx.x - ly * WPR;
  // An image whose width is not a multiple of four ends INSIDE a word: the byte behind its last column (column w, the right
  // neighbour of column w - 1) must read as column w - 2 (reflect-101), not as the pad byte memory holds there.  The thread that
  // stages that word fetches the byte for each of its rows and puts it in place before the word goes to LDS.
  const bool patch = (tw & 3) != 0 && c == (tw >> 2);
  const int patch_shift = 8 * (tw & 3);
  // reflect-101 of a row index in [-1, h]: |y|, then folded at the bottom edge
  auto reflect_row = [&](int y) { const int ay = y < 0 ? -y : y; return max(min(ay, 2 * h - 2 - ay), 0); };   // (h = 1: row 0)
  // a thread stages its own column of words: rows ly, ly + 8, ...  All of its loads are issued before the first is waited
  // for: rows / columns past the tile's patch read a clamped (valid) address and are not written to LDS.
  constexpr int kStage = (TH + 2 + kGradVRows - 1) / kGradVRows;
  const uint32_t colb = (uint32_t)(x0 + min(4 * c, twp - 4));
  uint32_t staged[kStage], mirrored[kStage];
#pragma unroll
  for (int i = 0; i < kStage; i++) {
    const int r = min(ly + kGradVRows * i, rows - 1);
    const uint32_t row = __umul24((unsigned)reflect_row(y0 + r - 1), (unsigned)pitch);
    staged[i] = *reinterpret_cast<const uint32_t*>(img + (row + colb));
    mirrored[i] = patch ? img[row + (uint32_t)max(w - 2, 0)] : 0u;
  }
  if (patch) {
#pragma unroll
    for (int i = 0; i < kStage; i++) staged[i] = (staged[i] & ~(0xffu << patch_shift)) | (mirrored[i] << patch_shift);
  }
  // the two halo bytes of a patch row (threads 0 .. 2·rows-1 keep theirs), requested behind the words
  const int hr = min((int)threadIdx.x >> 1, rows - 1), side = threadIdx.x & 1;
  const uint32_t halo = img[__umul24((unsigned)reflect_row(y0 + hr - 1), (unsigned)pitch) + (unsigned)reflect101(side ? x0 + tw : x0 - 1, w)];
  if (4 * c < twp) {   // (rows past the patch hold a clamped row's words: never read)
#pragma unroll
    for (int i = 0; i < kStage; i++) {
      const int r = ly + kGradVRows * i;
      tile[r][1 + c] = staged[i];   // (the tile has kStage·8 rows)
    }
  }
  if ((int)threadIdx.x < 2 * rows) {
    if (!side) tile[hr][0] = halo << 24;
    else if ((tw & 3) == 0) tile[hr][1 + tw / 4] = halo;   // (a tile that ends inside a word: the staging thread patched the byte)
  }
  __syncthreads();
  const int x = x0 + 4 * c, yb = y0 + ly * RPT;
  if (4 * c >= twp || yb >= h) return;
  // Separable form over packed 16-bit pairs (round 3; the scalar form spent ~30 integer instructions per pixel and was bound
  // by them, not by its 5 bytes per pixel).  Per source row and pixel j: the horizontal difference d = p[j+2] - p[j] and the
  // horizontal smoothing s = 3 p[j] + 10 p[j+1] + 3 p[j+2]; then gx = 3 (3 d_top + 10 d_mid + 3 d_bot) = 9 (d_top + d_bot) +
  // 30 d_mid and gy = 3 (s_bot - s_top) — the same integers (|gx|, |gy| <= 12240, s <= 4080: everything fits 16 bits).  A
  // row's six pixels x-1 .. x+4 are spread into five overlapping pairs by v_perm_b32, and every sum runs on two pixels at
  // once (v_pk_add/sub/mul/mad on i16); the results are the stored i16 pairs as they stand.
  typedef short s2v __attribute__((ext_vector_type(2)));
  struct RowTerms { s2v d01, d23, s01, s23; };
  auto row_terms = [&](int r) {
    const uint32_t wl = tile[r][c], wc = tile[r][c + 1], wr = tile[r][c + 2];
    auto pair = [](uint32_t hi_src, uint32_t lo_src, uint32_t sel) {   // two bytes of {hi_src, lo_src} zero-extended to u16 x 2
      const uint32_t v = __builtin_amdgcn_perm(hi_src, lo_src, sel);
      return __builtin_bit_cast(s2v, v);
    };
    const s2v p01 = pair(wc, wl, 0x0c040c03u);   // (x-1, x)
    const s2v p12 = pair(0u, wc, 0x0c010c00u);   // (x, x+1)
    const s2v p23 = pair(0u, wc, 0x0c020c01u);
    const s2v p34 = pair(0u, wc, 0x0c030c02u);
    const s2v p45 = pair(wr, wc, 0x0c040c03u);   // (x+3, x+4)
    RowTerms t;
    t.d01 = p23 - p01;
    t.d23 = p45 - p23;
    t.s01 = (p01 + p23) * (short)3 + p12 * (short)10;
    t.s23 = (p23 + p45) * (short)3 + p34 * (short)10;
    return t;
  };
  auto out_rows = [&](auto whole) {   // whole: every row of the tile is inside the image (no per-row test)
    RowTerms r0 = row_terms(ly * RPT), r1 = row_terms(ly * RPT + 1);
    uint32_t o = 2u * (__umul24((unsigned)yb, (unsigned)pitch) + (unsigned)x);   // byte offset inside the frame's plane
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      if (decltype(whole)::value || yb + k < h) {
        const RowTerms r2 = row_terms(ly * RPT + k + 2);
        const s2v gx01 = (r0.d01 + r2.d01) * (short)9 + r1.d01 * (short)30;
        const s2v gx23 = (r0.d23 + r2.d23) * (short)9 + r1.d23 * (short)30;
        const s2v gy01 = (r2.s01 - r0.s01) * (short)3;
        const s2v gy23 = (r2.s23 - r0.s23) * (short)3;
        *reinterpret_cast<uint2*>(gxf + o) = make_uint2(__builtin_bit_cast(uint32_t, gx01), __builtin_bit_cast(uint32_t, gx23));
        *reinterpret_cast<uint2*>(gyf + o) = make_uint2(__builtin_bit_cast(uint32_t, gy01), __builtin_bit_cast(uint32_t, gy23));
        o += 2u * (unsigned)pitch;
        r0 = r1;
        r1 = r2;
      }
    }
  };
  if (y0 + TH <= h) out_rows(std::true_type{});
  else out_rows(std::false_type{});
}

template <int RPT>
__global__ __launch_bounds__(kBlock) void k_scharr3_v4(const uint8_t* __restrict__ src, int16_t* __restrict__ gx,
                                                       int16_t* __restrict__ gy, int w, int h, int pitch, size_t frame_stride,
                                                       const int* __restrict__ slots, int first_slot) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[scharr_v4_lds_rows(RPT) * (kGradVW / 4 + 2) * 4];
  // XCD-aware order of the tiles: blocks are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its own
  // L2), and a tile's halo bytes — one 128-byte line to the left and one to the right of every patch row — are its
  // neighbours' data: with the tiles in launch order every XCD fetched them for itself (2.75 x the plane per launch in
  // FETCH_SIZE).  Block b takes tile start(b mod 8) + b div 8, so that consecutive tiles (neighbours along x, then the next
  // tile row, then the next frame) run on ONE XCD one after the other and find those lines in its L2.
  const unsigned nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned xcd = b & 7u, q = nb >> 3, r = nb & 7u;
  const unsigned t = xcd * q + min(xcd, r) + (b >> 3);   // XCD x owns q + (x < r) consecutive tiles
  const unsigned fy = t / gridDim.x, tile = t - fy * gridDim.x;
  const int slot = slots ? slots[fy] : first_slot + (int)fy;
  scharr_tile_v4<RPT>(lds, src, gx, gy, w, h, pitch, frame_stride, slot, (int)tile);
}

// The gradients of every level of a frame (or a few) in one launch: block -> (level, tile) through the levels' tile
// counts, the 128x8 vector tile.
constexpr int kGradMaxLevels = 8;
struct GradLevelsArgs {
  const uint8_t* src[kGradMaxLevels];
  int16_t* gx[kGradMaxLevels];
  int16_t* gy[kGradMaxLevels];
  int w[kGradMaxLevels], h[kGradMaxLevels], pitch[kGradMaxLevels];   // the level's image size and its row pitch
  size_t stride[kGradMaxLevels];
  int tile_end[kGradMaxLevels];   // running sum of the levels' tile counts
  int n_levels;
  const int* slots;
  int first_slot;
};

static __global__ __launch_bounds__(kBlock) void k_scharr3_levels(const GradLevelsArgs a) {
  constexpr int kV4Bytes = scharr_v4_lds_rows(1) * (kGradVW / 4 + 2) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kV4Bytes];
  const int slot = a.slots ? a.slots[blockIdx.y] : a.first_slot + (int)blockIdx.y;
  const int b = (int)blockIdx.x;
  // the level's parameters by selects over the (by-value) table: an indexed read would go through scratch memory
  const uint8_t* src = a.src[0];
  int16_t *gx = a.gx[0], *gy = a.gy[0];
  int w = a.w[0], h = a.h[0], pitch = a.pitch[0], tile0 = 0;
  size_t stride = a.stride[0];
#pragma unroll
  for (int l = 1; l < kGradMaxLevels; l++)
    if (l < a.n_levels && b >= a.tile_end[l - 1]) {
      src = a.src[l]; gx = a.gx[l]; gy = a.gy[l]; w = a.w[l]; h = a.h[l]; pitch = a.pitch[l]; stride = a.stride[l]; tile0 = a.tile_end[l - 1];
    }
  scharr_tile_v4<1>(lds, src, gx, gy, w, h, pitch, stride, slot, b - tile0);
}

// ------------------------------------------------------------------------------------------------------------
// per-pixel terms: WarpFunction (src/Tracker.cpp:1439-1467) + validity / Jw / residual / Jl·Jw (:447-479).
// Float op order is the contract (mirrors oracle S1): small products are k-sequential FMA chains, everything
// else separate mul/add, divisions correctly rounded.
// ------------------------------------------------------------------------------------------------------------
struct WarpK {
  float T[12];
  double Td[12];   // the same entries widened (kArithOpenCV: the rigid product runs in double); block-uniform
};

// the rigid matrix of `pose` as block-uniform values (scalar registers); AR == kArithOpenCV also fills Td
template <int AR>
__device__ __forceinline__ void warp_setup(const Pose& pose, WarpK& K) {
  pose_to_T12(pose, K.T);
#pragma unroll
  for (int i = 0; i < 12; i++) {
    K.T[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(K.T[i])));
    if constexpr (AR == kArithOpenCV) {
      const unsigned long long b = (unsigned long long)__double_as_longlong((double)K.T[i]);
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
      const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
      K.Td[i] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
  }
}

// Packed f32.  On gfx950 a plain f32 multiply / add / fma issues in 2 cycles per wave only while every operand is a VGPR,
// an inline constant or a literal; with a scalar-register operand (the rigid matrix, the intrinsics) it takes 4, like
// every compare, select, conversion and integer op (tools/ubench/valu_classes*.hip).  v_pk_{mul,add,fma}_f32 take 4
// cycles for TWO pixels with or without a scalar operand, so the per-pixel float sequence is written over pairs of
// adjacent pixels: the block-uniform operands stay in SGPRs (no VGPR cost, 4 waves/SIMD kept) and cost nothing extra.
// Each component sees exactly the scalar IEEE operation (same rounding, no contraction), so results are bit-identical
// to the one-pixel form (F = float), which the per-stage kernels (general_pixel, the point tables, k_warp_table) use.  (The
// templates below still take VEC = 1 — one pixel per step — but since round 6 nothing instantiates it: every level is walked in
// groups of four, RAGGED where its grid rows are not whole groups.)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));   // (the type __builtin_nontemporal_load takes for an 8-byte load)

template <typename F> struct lanes { static constexpr int n = 1; };
template <> struct lanes<v2f> { static constexpr int n = 2; };
template <typename F> __device__ __forceinline__ F bc(float s) { return (F)(s); }   // broadcast a (uniform) scalar
__device__ __forceinline__ float get(float v, int) { return v; }
__device__ __forceinline__ float get(v2f v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ void put(float& v, int, float s) { v = s; }
__device__ __forceinline__ void put(v2f& v, int i, float s) { if (i) v.y = s; else v.x = s; }
template <typename F> __device__ __forceinline__ F fma_(F a, F b, F c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float rcp_(float d) { return __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ v2f rcp_(v2f d) { return (v2f){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)}; }

// Correctly rounded f32 quotients sharing one refined reciprocal of the common denominator: the same
// rcp / Newton / two-residual-correction sequence hipcc emits for an IEEE `a / b`, minus the div_scale / div_fixup
// range handling.  Bit-identical to `a / b` for normal-range operands and quotients; a zero, infinite or NaN
// denominator yields NaN here (IEEE: NaN or infinity), and such a pixel fails the bounds test either way.
// refined_rcp(d) itself equals the IEEE 1.0f / d for every one of the 2^23 mantissas (checked exhaustively at nine
// exponents, both signs, by tools/ubench/rcp_check.hip on gfx950), so a reciprocal needs no further correction; a general
// numerator does (Markstein: n * r can be 1.5 ulp off), hence the two residual corrections in div_by.
template <typename F>
__device__ __forceinline__ F refined_rcp(F d) {
  F r = rcp_(d);
  const F e = fma_(-d, r, bc<F>(1.0f));
  return fma_(e, r, r);
}
template <typename F>
__device__ __forceinline__ F div_by(F n, F d, F r) {
  F q = n * r;
  F e = fma_(-d, q, n);
  q = fma_(e, r, q);
  e = fma_(-d, q, n);
  return fma_(e, r, q);
}

// One row of rigid * points.t() (src/Tracker.cpp:1450) as cv::gemm computes it on CV_32F (GEMMSingleMul<float,double>,
// GEMM_2_T, len 4: s0 = a0*b0, s1 = a1*b1, s2 = a2*b2, s3 = a3*b3 in double — exact — then "s0 += s1 + s2 + s3;", whose
// right-hand side C++ evaluates first: T((s0 + ((s1 + s2) + s3)) * alpha)).  A product of two widened floats is exact in
// double, so fma(a, b, s) = RN(s + a*b) is the separate add.
__device__ __forceinline__ float rigid_row_f64(const double* Td, double Xd, double Yd, double zd) {   // w = 1
  double t = Td[1] * Yd;
  t = __builtin_fma(Td[2], zd, t);
  t = t + Td[3];
  return (float)__builtin_fma(Td[0], Xd, t);
}
__device__ __forceinline__ float rigid_row_f64(const double* Td, double Xd, double Yd, double zd, double wd) {
  double t = Td[1] * Yd;
  t = __builtin_fma(Td[2], zd, t);
  t = __builtin_fma(Td[3], wd, t);
  return (float)__builtin_fma(Td[0], Xd, t);
}

template <int AR, typename F>
__device__ __forceinline__ void warp_point(const LevelK& L, const WarpK& K, F xf, F yf, F z, F& u, F& v, F& zp, F& iz) {
  F X, Y, xp, yp;
  if constexpr (AR == kArithLegacy) {
    X = (xf - bc<F>(L.cx)) * bc<F>(L.invfx);
    Y = (yf - bc<F>(L.cy)) * bc<F>(L.invfy);
  } else {   // convertTo(alpha = invfx, beta = -cx * invfx) -> cvtScale32f: x * alpha + beta, two f32 operations
    X = xf * bc<F>(L.invfx);
    X = X + bc<F>(L.bx);
    Y = yf * bc<F>(L.invfy);
    Y = Y + bc<F>(L.by);
  }
  X = X * z;
  Y = Y * z;
  if constexpr (AR == kArithLegacy) {
    xp = bc<F>(K.T[0]) * X;
    xp = fma_(bc<F>(K.T[1]), Y, xp);
    xp = fma_(bc<F>(K.T[2]), z, xp);
    xp = xp + bc<F>(K.T[3]);  // fma(T03, w = 1, xp)
    yp = bc<F>(K.T[4]) * X;
    yp = fma_(bc<F>(K.T[5]), Y, yp);
    yp = fma_(bc<F>(K.T[6]), z, yp);
    yp = yp + bc<F>(K.T[7]);
    zp = bc<F>(K.T[8]) * X;
    zp = fma_(bc<F>(K.T[9]), Y, zp);
    zp = fma_(bc<F>(K.T[10]), z, zp);
    zp = zp + bc<F>(K.T[11]);
  } else {
#pragma unroll
    for (int c = 0; c < lanes<F>::n; c++) {
      const double Xd = (double)get(X, c), Yd = (double)get(Y, c), zd = (double)get(z, c);
      put(xp, c, rigid_row_f64(K.Td, Xd, Yd, zd));
      put(yp, c, rigid_row_f64(K.Td + 4, Xd, Yd, zd));
      put(zp, c, rigid_row_f64(K.Td + 8, Xd, Yd, zd));
    }
  }
  const F r = refined_rcp(zp);
  u = xp * bc<F>(L.fx);
  u = div_by(u, zp, r);
  u = u + bc<F>(L.cx);
  v = yp * bc<F>(L.fy);
  v = div_by(v, zp, r);
  v = v + bc<F>(L.cy);
  iz = r;  // inv_z2 = 1 / z2 (src/Tracker.cpp:447): the refined reciprocal IS the correctly rounded quotient (see refined_rcp)
}

// C round() (half away from zero) for the x >= 0 this path produces: v_cvt_rpi_i32_f32 = floor(x + 0.5) evaluated
// without an intermediate rounding — checked against roundf on every tie up to 4200 and 2M random values by
// tools/ubench/rpi_check.hip (0.49999997 -> 0, n + 0.5 -> n + 1).  One quarter-rate op instead of five.
__device__ __forceinline__ int round_pos(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// v ? x : 0 under a 64-lane mask held in an SGPR pair.  Written as the instruction so that the compiler keeps the
// sanitising selects straight-line: left to itself it turns them into an exec-masked region per pixel and pays four
// v_mov of zero, a saveexec and a branch for each.
// (In place: a select that sits in a conditional region then needs no copy where the paths meet.)
__device__ __forceinline__ float keep_f(float x, unsigned long long mask) {
  asm("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(x) : "s"(mask));
  return x;
}
__device__ __forceinline__ int keep_i(int x, unsigned long long mask) {
  asm("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(x) : "s"(mask));
  return x;
}

// Phase 1 of a pixel (F = float) or of two adjacent pixels (F = v2f): warp, validity (src/Tracker.cpp:450-453) and the
// gather index of the nearest-neighbour sample (:472).  Branch-free: an invalid pixel is sanitised (x2 = y2 = iz = 0) so
// that every later term is finite and its Jacobian row comes out as exact zeros, which leave the accumulators unchanged.
// Validity is built as a 64-lane mask in scalar registers: every comparison writes an SGPR pair, the conjunction is
// s_and_b64, nothing of it occupies the VALU beyond the compares themselves.  (LLVM FCmp / ICmp predicate codes.)
// The reference's fifth condition, z2 != 0 (:451), needs no compare of its own: a zero z2 makes the reciprocal infinite,
// refined_rcp turns that into NaN (0 * inf), x2 and y2 come out NaN and fail x2 > 0 — exactly as the IEEE quotient by
// zero (infinite or NaN) fails one of the reference's four bounds tests before z2 != 0 is looked at.
constexpr int kFcmpOGT = 2, kFcmpOLT = 4, kFcmpUGE = 11, kFcmpUNE = 14, kIcmpULT = 36, kIcmpSGT = 38, kIcmpSLT = 40;
__device__ __forceinline__ bool lane_bit(unsigned long long mask) { return (mask >> (threadIdx.x & 63u)) & 1ull; }

// 1a: warp and validity masks; x2, y2, iz come back raw (possibly NaN / out of range where the mask is clear)
template <int AR, typename F>
__device__ __forceinline__ void pixel_warp_raw(const LevelK& L, const WarpK& K, F xf, F yf, F z,
                                               const unsigned long long* okin_mask, F& x2, F& y2, F& iz,
                                               unsigned long long* okm) {
  F z2;
  warp_point<AR, F>(L, K, xf, yf, z, x2, y2, z2, iz);
#pragma unroll
  for (int c = 0; c < lanes<F>::n; c++) {
    const float uc = get(x2, c), vc = get(y2, c);
    okm[c] = okin_mask[c] & __builtin_amdgcn_fcmpf(vc, 0.f, kFcmpOGT) & __builtin_amdgcn_fcmpf(vc, (float)L.ih, kFcmpOLT) &
             __builtin_amdgcn_fcmpf(uc, 0.f, kFcmpOGT) & __builtin_amdgcn_fcmpf(uc, (float)L.iw, kFcmpOLT);
  }
}
// 1b: sanitise invalid pixels, clamp a negative reciprocal: "if (inv_z2 < 0) inv_z2 = 0" (:452-453)
template <typename F>
__device__ __forceinline__ void pixel_sanitize(F& x2, F& y2, F& iz, const unsigned long long* okm) {
#pragma unroll
  for (int c = 0; c < lanes<F>::n; c++) {
    const float rc = get(iz, c);
    put(x2, c, keep_f(get(x2, c), okm[c]));
    put(y2, c, keep_f(get(y2, c), okm[c]));
    put(iz, c, keep_f(rc, okm[c] & __builtin_amdgcn_fcmpf(rc, 0.f, kFcmpUGE)));
  }
}
// 1c: index of the nearest-neighbour sample of (sanitised, or valid) x2, y2
template <typename F>
__device__ __forceinline__ void pixel_gather_index(const LevelK& L, F x2, F y2, uint32_t* gidx) {
#pragma unroll
  for (int c = 0; c < lanes<F>::n; c++) {
    int ix2 = round_pos(get(x2, c)), iy2 = round_pos(get(y2, c));
    ix2 = min(ix2, L.iw - 1);  // the reference reads one past the edge here (:450, :472); clamp
    iy2 = min(iy2, L.ih - 1);
    gidx[c] = __umul24((unsigned)iy2, (unsigned)L.pitch) + (unsigned)ix2;  // 24-bit multiply-add: one op (dims < 2^24)
  }
}
template <int AR, typename F>
__device__ __forceinline__ void pixel_warp(const LevelK& L, const WarpK& K, F xf, F yf, F z,
                                           const unsigned long long* okin_mask, F& x2, F& y2, F& iz,
                                           unsigned long long* okm, uint32_t* gidx) {
  pixel_warp_raw<AR, F>(L, K, xf, yf, z, okin_mask, x2, y2, iz, okm);
  pixel_sanitize(x2, y2, iz, okm);
  pixel_gather_index(L, x2, y2, gidx);
}

// Phase 2: Jw (src/Tracker.cpp:455-467) and Jacobian_row = Jl * Jw (:476-479), reference operation order.
// SQUARE (fx == fy bitwise, decided at launch): (fx*x2) = (fy*x2), (fx*y2) = (fy*y2), fx*iz = fy*iz are the same
// rounded values, so the reference's products that coincide up to sign are computed once — bit-identical results.
// SIGNED_ZEROS (the per-stage dump entry points): keep the reference's fma with the structural zero of Jw, which decides
// the sign of a zero J[0] / J[1] (g0 * a0 = -0, g1 * 0 = +0 -> +0).  The sums never see the difference (x + -0 = x,
// +0 + -0 = +0), so the solver path drops the two operations.
// AR == kArithOpenCV: Jl * Jw is a cv::gemm (flags 0, len 2, 1x6 result: GEMMSingleMul<float,double>, no inline special case —
// that needs len == d_size.width or height): J[k] = (float)((0 + (double)g0 * Jw0k) + (double)g1 * Jw1k).  The products are
// exact in double (a gradient has 14 bits), so the second step is one fma; columns 0 and 1 have a structural zero in Jw
// and reduce to the f32 product (the exact product rounded once).  A zero row entry is +0 there (the sum starts from +0);
// SIGNED_ZEROS reproduces that for the per-stage dumps, the sums cannot see it.
// the non-zero entries of Jw (src/Tracker.cpp:455-467): row 0 = (a0, 0, a2, a3, a4, a5), row 1 = (0, b1, b2, b3, b4, b5)
template <bool UNIT_FACTORS, bool SQUARE, typename F>
__device__ __forceinline__ void jw_terms(const LevelK& L, float zf_, float af_, F x2, F y2, F iz, F& a0, F& b1, F a[4], F b[4]) {
  const F fx = bc<F>(L.fx), fy = bc<F>(L.fy), one = bc<F>(1.0f);
  F a2, a3, a5, b2, b4, b5;
  if constexpr (SQUARE) {
    const F fx2 = fx * x2, fy2 = fx * y2;
    a0 = fx * iz;
    b1 = a0;
    b5 = fx2 * iz;                      // (fy*x2)*iz
    a2 = -(b5 * iz);                    // -(((fx*x2)*iz)*iz)
    b4 = ((fx2 * y2) * iz) * iz;        // (((fy*x2)*y2)*iz)*iz
    a3 = -b4;                           // -((((fx*x2)*y2)*iz)*iz)
    const F t = fy2 * iz;               // (fy*y2)*iz
    a5 = -t;                            // ((-fx)*y2)*iz
    b2 = -(t * iz);                     // -(((fy*y2)*iz)*iz)
  } else {
    a0 = fx * iz;
    a2 = -(((fx * x2) * iz) * iz);
    a3 = -((((fx * x2) * y2) * iz) * iz);
    a5 = ((-fx) * y2) * iz;
    b1 = fy * iz;
    b2 = -(((fy * y2) * iz) * iz);
    b4 = (((fy * x2) * y2) * iz) * iz;
    b5 = (fy * x2) * iz;
  }
  F a4 = fx * (one + ((x2 * x2) * iz) * iz);
  F b3 = -(fy * (one + ((y2 * y2) * iz) * iz));
  if constexpr (!UNIT_FACTORS) {
    const F zf = bc<F>(zf_), af = bc<F>(af_);
    a2 = a2 * zf; a3 = a3 * af; a4 = a4 * af; a5 = a5 * af;
    b2 = b2 * zf; b3 = b3 * af; b4 = b4 * af; b5 = b5 * af;
  }
  a[0] = a2; a[1] = a3; a[2] = a4; a[3] = a5;
  b[0] = b2; b[1] = b3; b[2] = b4; b[3] = b5;
}

template <int AR, bool UNIT_FACTORS, bool SQUARE = false, bool SIGNED_ZEROS = false, typename F = float>
__device__ __forceinline__ void pixel_jacobian(const LevelK& L, float zf_, float af_, F x2, F y2, F iz, F g0, F g1, F J[6]) {
  const F zero = bc<F>(0.0f);
  F a0, b1, av[4], bv[4];
  jw_terms<UNIT_FACTORS, SQUARE, F>(L, zf_, af_, x2, y2, iz, a0, b1, av, bv);
  const F a2 = av[0], a3 = av[1], a4 = av[2], a5 = av[3], b2 = bv[0], b3 = bv[1], b4 = bv[2], b5 = bv[3];
  if constexpr (AR == kArithOpenCV) {
    if constexpr (SIGNED_ZEROS) {   // (0 + p0) + p1 in double: a zero result is +0
      J[0] = (g0 * a0) + zero;
      J[1] = (g1 * b1) + zero;
    } else {
      J[0] = g0 * a0;
      J[1] = g1 * b1;
    }
#pragma unroll
    for (int c = 0; c < lanes<F>::n; c++) {
      const double g0d = (double)get(g0, c), g1d = (double)get(g1, c);
      const F* as[4] = {&a2, &a3, &a4, &a5};
      const F* bs[4] = {&b2, &b3, &b4, &b5};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        double p = g0d * (double)get(*as[k], c);
        if constexpr (SIGNED_ZEROS) p = 0.0 + p;
        p = __builtin_fma(g1d, (double)get(*bs[k], c), p);
        put(J[2 + k], c, (float)p);
      }
    }
  } else {
    if constexpr (SIGNED_ZEROS) {
      J[0] = fma_(g1, zero, g0 * a0);
      J[1] = fma_(g1, b1, g0 * zero);
    } else {
      J[0] = g0 * a0;
      J[1] = g1 * b1;
    }
    J[2] = fma_(g1, b2, g0 * a2);
    J[3] = fma_(g1, b3, g0 * a3);
    J[4] = fma_(g1, b4, g0 * a4);
    J[5] = fma_(g1, b5, g0 * a5);
  }
}

// The kArithOpenCV row of one pixel pair handed to the f64 sums as doubles (the identity path of residual_core): column k >= 2
// is (double)(float)(g0 * Jw0k + g1 * Jw1k) with the sum formed in double, columns 0 and 1 the widened f32 products.
// (j0 = g0 * a0, j1 = g1 * b1: the packed f32 products; av, bv: jw_terms' columns 2..5; c: the pixel of the pair)
// SQUARE: jw_terms made a3 = -b4 (av[1] = -bv[2]), so its widening is the other's with the sign flipped: one conversion less.
template <bool SQUARE, typename F>
__device__ __forceinline__ void jacobian_row_f64(F g0, F g1, F j0, F j1, const F av[4], const F bv[4], int c, double Jd[6]) {
  const double g0d = (double)get(g0, c), g1d = (double)get(g1, c);
  Jd[0] = (double)get(j0, c);
  Jd[1] = (double)get(j1, c);
  double b4d = 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    constexpr int order[4] = {0, 2, 1, 3};   // column 4 (k = 2) ahead of column 3 (k = 1), which reuses its widened b4
    const int k = order[i];
    const double bd = (double)get(bv[k], c);
    if (k == 2) b4d = bd;
    const double ad = (SQUARE && k == 1) ? -b4d : (double)get(av[k], c);
    double p = g0d * ad;
    p = __builtin_fma(g1d, bd, p);
    Jd[2 + k] = (double)(float)p;
  }
}

// The same row as the six floats the reference holds (pixel_jacobian's kArithOpenCV arithmetic for pixel c of a unit): what a
// caller that still has to weight the row needs (w * J[k] is an f32 product of the stored float).
template <bool SQUARE, typename F>
__device__ __forceinline__ void jacobian_row_f32(F g0, F g1, F j0, F j1, const F av[4], const F bv[4], int c, float J[6]) {
  const double g0d = (double)get(g0, c), g1d = (double)get(g1, c);
  J[0] = get(j0, c);
  J[1] = get(j1, c);
  double b4d = 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    constexpr int order[4] = {0, 2, 1, 3};   // as jacobian_row_f64
    const int k = order[i];
    const double bd = (double)get(bv[k], c);
    if (k == 2) b4d = bd;
    const double ad = (SQUARE && k == 1) ? -b4d : (double)get(av[k], c);
    double p = g0d * ad;
    p = __builtin_fma(g1d, bd, p);
    J[2 + k] = (float)p;
  }
}

// Accumulator type: double reproduces the reference's double-accumulating gemm (src/Tracker.cpp:560-561) to the
// last bit of the f32 result in practice (products of two f32 are exact in f64); float is the cheaper variant.
__device__ __forceinline__ void accumulate(float acc[kAccFloats], const float J[6], int ri) {
  const float r = (float)ri;
  int s = 0;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++, s++) acc[s] = __builtin_fmaf(J[i], J[j], acc[s]);
#pragma unroll
  for (int i = 0; i < 6; i++) acc[21 + i] = __builtin_fmaf(J[i], r, acc[21 + i]);
}

__device__ __forceinline__ void accumulate(double acc[kAccFloats], const float J[6], int ri) {
  double Jd[6];
#pragma unroll
  for (int i = 0; i < 6; i++) Jd[i] = (double)J[i];
  const double rd = (double)ri;  // the residual is an integer: one conversion instead of two

  int s = 0;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++, s++) acc[s] = __builtin_fma(Jd[i], Jd[j], acc[s]);
#pragma unroll
  for (int i = 0; i < 6; i++) acc[21 + i] = __builtin_fma(Jd[i], rd, acc[21 + i]);
}

// General path: J <- w·J, r <- gain·r, A += (wJ)(wJ)ᵀ, jtr += (wJ)·((gain r)·w) (src/Tracker.cpp:554-561) and the error
// numerator Σ r·(r·w) (:499-502).  An invalid (sanitised) pixel has J = 0 and r = 0, so it adds exact zeros.
template <typename AccT>
__device__ __forceinline__ void accumulate_weighted(AccT acc[kAccFloats], AccT& err, const float J[6], float rf, float w,
                                                    float gain) {
  const float rw1 = rf * w;
  err = (AccT)__builtin_fma((double)rf, (double)rw1, (double)err);
  const float rw = (rf * gain) * w;
  double Jd[6];
#pragma unroll
  for (int k = 0; k < 6; k++) Jd[k] = (double)(w * J[k]);
  int s = 0;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++, s++) acc[s] = (AccT)__builtin_fma(Jd[i], Jd[j], (double)acc[s]);
#pragma unroll
  for (int i = 0; i < 6; i++) acc[21 + i] = (AccT)__builtin_fma(Jd[i], (double)rw, (double)acc[21 + i]);
}

// the same sums from a row that has been weighted already: wJ[k] = w * J[k] (f32), rwd = (double)((r * gain) * w)
__device__ __forceinline__ void accumulate_preweighted(double acc[kAccFloats], const float wJ[6], double rwd) {
  double Jd[6];
#pragma unroll
  for (int k = 0; k < 6; k++) Jd[k] = (double)wJ[k];
  int s = 0;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++, s++) acc[s] = __builtin_fma(Jd[i], Jd[j], acc[s]);
#pragma unroll
  for (int i = 0; i < 6; i++) acc[21 + i] = __builtin_fma(Jd[i], rwd, acc[21 + i]);
}
__device__ __forceinline__ void accumulate_preweighted(float acc[kAccFloats], const float wJ[6], double rwd) {
  const float rw = (float)rwd;   // exact: rwd was converted from this float
  int s = 0;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++, s++) acc[s] = (float)__builtin_fma((double)wJ[i], (double)wJ[j], (double)acc[s]);
#pragma unroll
  for (int i = 0; i < 6; i++) acc[21 + i] = (float)__builtin_fma((double)wJ[i], (double)rw, (double)acc[21 + i]);
}

// Masked accumulation (round 3).  The sums of one pixel are added under an EXEC mask that holds the lanes whose pixel is
// valid — two scalar instructions (s_and_saveexec_b64, s_mov_b64 exec) around the 28 f64 operations — instead of zeroing
// every invalid pixel's gradients and residual with vector selects so that its row adds exact zeros: an invalid pixel's
// lane simply does not take part.  Its x2, y2, 1 / z2 therefore need no sanitising either (a NaN there never reaches a
// sum): 6 selects and a compare per pixel less, 376 -> 347 vector instructions per 4 pixels on the identity path.  The
// sums are the same bit for bit: a masked-out lane keeps its accumulators, exactly what adding +-0 did.
// One asm statement cannot take all 35 operands (the limit is 30), hence two statements of 14 sums each.
__device__ __forceinline__ void masked_sums_lo(double acc[kAccFloats], const double Jd[6], unsigned long long mask) {
  unsigned long long saved;
  asm volatile("s_and_saveexec_b64 %14, %21\n\t"
               "v_fmac_f64 %0, %15, %15\n\t"
               "v_fmac_f64 %1, %15, %16\n\t"
               "v_fmac_f64 %2, %15, %17\n\t"
               "v_fmac_f64 %3, %15, %18\n\t"
               "v_fmac_f64 %4, %15, %19\n\t"
               "v_fmac_f64 %5, %15, %20\n\t"
               "v_fmac_f64 %6, %16, %16\n\t"
               "v_fmac_f64 %7, %16, %17\n\t"
               "v_fmac_f64 %8, %16, %18\n\t"
               "v_fmac_f64 %9, %16, %19\n\t"
               "v_fmac_f64 %10, %16, %20\n\t"
               "v_fmac_f64 %11, %17, %17\n\t"
               "v_fmac_f64 %12, %17, %18\n\t"
               "v_fmac_f64 %13, %17, %19\n\t"
               "s_mov_b64 exec, %14"
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                 "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "=&s"(saved)
               : "v"(Jd[0]), "v"(Jd[1]), "v"(Jd[2]), "v"(Jd[3]), "v"(Jd[4]), "v"(Jd[5]), "s"(mask)
               : "scc");
}
// sums 14..26 and a 28th: TAIL 0: x += rd * rd (the sum of r^2, integers below 2^53: exact); TAIL 1: x += e (the error term of
// the weighted path, see WeightEntry)
template <int TAIL>
__device__ __forceinline__ void masked_sums_hi(double acc[kAccFloats], double& x, const double Jd[6], double rd, double e,
                                               unsigned long long mask) {
  unsigned long long saved;
  if constexpr (TAIL == 0) {
    asm volatile("s_and_saveexec_b64 %14, %22\n\t"
                 "v_fmac_f64 %0, %17, %20\n\t"
                 "v_fmac_f64 %1, %18, %18\n\t"
                 "v_fmac_f64 %2, %18, %19\n\t"
                 "v_fmac_f64 %3, %18, %20\n\t"
                 "v_fmac_f64 %4, %19, %19\n\t"
                 "v_fmac_f64 %5, %19, %20\n\t"
                 "v_fmac_f64 %6, %20, %20\n\t"
                 "v_fmac_f64 %7, %15, %21\n\t"
                 "v_fmac_f64 %8, %16, %21\n\t"
                 "v_fmac_f64 %9, %17, %21\n\t"
                 "v_fmac_f64 %10, %18, %21\n\t"
                 "v_fmac_f64 %11, %19, %21\n\t"
                 "v_fmac_f64 %12, %20, %21\n\t"
                 "v_fmac_f64 %13, %21, %21\n\t"
                 "s_mov_b64 exec, %14"
                 : "+v"(acc[14]), "+v"(acc[15]), "+v"(acc[16]), "+v"(acc[17]), "+v"(acc[18]), "+v"(acc[19]), "+v"(acc[20]),
                   "+v"(acc[21]), "+v"(acc[22]), "+v"(acc[23]), "+v"(acc[24]), "+v"(acc[25]), "+v"(acc[26]), "+v"(x), "=&s"(saved)
                 : "v"(Jd[0]), "v"(Jd[1]), "v"(Jd[2]), "v"(Jd[3]), "v"(Jd[4]), "v"(Jd[5]), "v"(rd), "s"(mask)
                 : "scc");
  } else {
    asm volatile("s_and_saveexec_b64 %14, %22\n\t"
                 "v_fmac_f64 %0, %17, %20\n\t"
                 "v_fmac_f64 %1, %18, %18\n\t"
                 "v_fmac_f64 %2, %18, %19\n\t"
                 "v_fmac_f64 %3, %18, %20\n\t"
                 "v_fmac_f64 %4, %19, %19\n\t"
                 "v_fmac_f64 %5, %19, %20\n\t"
                 "v_fmac_f64 %6, %20, %20\n\t"
                 "v_fmac_f64 %7, %15, %21\n\t"
                 "v_fmac_f64 %8, %16, %21\n\t"
                 "v_fmac_f64 %9, %17, %21\n\t"
                 "v_fmac_f64 %10, %18, %21\n\t"
                 "v_fmac_f64 %11, %19, %21\n\t"
                 "v_fmac_f64 %12, %20, %21\n\t"
                 "v_add_f64 %13, %13, %23\n\t"
                 "s_mov_b64 exec, %14"
                 : "+v"(acc[14]), "+v"(acc[15]), "+v"(acc[16]), "+v"(acc[17]), "+v"(acc[18]), "+v"(acc[19]), "+v"(acc[20]),
                   "+v"(acc[21]), "+v"(acc[22]), "+v"(acc[23]), "+v"(acc[24]), "+v"(acc[25]), "+v"(acc[26]), "+v"(x), "=&s"(saved)
                 : "v"(Jd[0]), "v"(Jd[1]), "v"(Jd[2]), "v"(Jd[3]), "v"(Jd[4]), "v"(Jd[5]), "v"(rd), "s"(mask), "v"(e)
                 : "scc");
  }
}

// Deterministic block reduction of the per-thread accumulators through LDS (fixed order, no atomics).
// Threads accumulate a handful of pixels in f32; from here on every sum is f64 so that the totals are, to ~1e-9,
// the exact sums the reference's double-accumulating gemm produces (src/Tracker.cpp:560-561).
// Stage 1 transposes 29 rows x 256 threads into LDS; stage 2: 232 threads each fold 32 columns; stage 3: 29
// threads fold the 8 segment sums and write the 256-B record.
// The LDS image of the reduction is carved out of a raw buffer so that a kernel can reuse the same bytes for another phase
// (sized for AccT = double).  PASS = rows per
// LDS pass: 14 (two passes, 34 KB: four blocks per CU) for the batch kernels; 27 (one pass, 60 KB) where a block has its
// CU to itself and the reduction's latency is on the critical path (k_iterate).
constexpr int reduce_lds_bytes(int pass) { return pass * kBlock * 8 + 2 * kBlock * 8 + (kAccFloats + 3) * 8 * 8; }
constexpr int kReduceLdsBytes = reduce_lds_bytes(14);   // 34688
constexpr int kIteratePass = kAccFloats;                // k_iterate, up to 3 pairs (blocks alone on their CUs): one pass, 61312 B

template <typename AccT, bool HAS_EXTRA = false, int PASS = 14, typename R2T = uint32_t>
__device__ __forceinline__ void block_reduce_store_at(unsigned char* __restrict__ lds, const AccT acc[kAccFloats],
                                                      R2T sum_r2, uint32_t n_valid, uint32_t* __restrict__ rec,
                                                      AccT extra = (AccT)0, bool coherent = false) {
  constexpr int kPass = PASS;  // accumulators per LDS pass
  constexpr int kRows = kAccFloats + (HAS_EXTRA ? 1 : 0);   // the extra sum rides in the last pass
  constexpr int kPasses = (kRows + kPass - 1) / kPass;
  static_assert(kPass + 2 <= 32, "the count rows are folded by threads v = kPass, kPass + 1 of pass 0");
  // red[kPass][256] AccT | cnt[2][256] f64 | seg[30][8] f64.  The two counts (valid pixels, Σ r² — integers below 2^53)
  // travel as doubles, which add exactly: with AccT = double they are simply rows kPass and kPass + 1 of the image and every
  // wave of the fold runs one code path.  seg rows are record slots: 0..26 sums, 27 valid count, 28 Σ r², 29 the extra sum.
  AccT(*red)[kBlock] = reinterpret_cast<AccT(*)[kBlock]>(lds);
  double(*cnt2)[kBlock] = reinterpret_cast<double(*)[kBlock]>(lds + kPass * kBlock * 8);
  double(*seg_f)[8] = reinterpret_cast<double(*)[8]>(lds + kPass * kBlock * 8 + 2 * kBlock * 8);
  const int tid = (int)thread_here();   // (opaque: see thread_here)
  const int v = tid >> 3, seg = tid & 7;
  cnt2[0][tid] = (double)n_valid;
  cnt2[1][tid] = (double)sum_r2;
#pragma unroll
  for (int pass = 0; pass < kPasses; pass++) {
    const int base = pass * kPass;
    const int cnt = kRows - base < kPass ? kRows - base : kPass;
    if (pass) __syncthreads();
#pragma unroll
    for (int i = 0; i < kPass; i++)
      if (i < cnt) red[i][tid] = (base + i < kAccFloats) ? acc[base + i < kAccFloats ? base + i : 0] : extra;
    __syncthreads();
    const bool count_row = pass == 0 && v >= kPass && v < kPass + 2;
    if constexpr (std::is_same<AccT, double>::value) {
      if (v < cnt || count_row) {
        const double* row = v < cnt ? red[v] : cnt2[v - kPass];
        double s = 0.0;
#pragma unroll 8
        for (int j = 0; j < 32; j++) s += row[seg * 32 + ((j + tid) & 31)];
        const int slot = v < cnt ? (base + v < kAccFloats ? base + v : 29) : 27 + v - kPass;
        seg_f[slot][seg] = s;
      }
    } else {
      if (v < cnt) {
        double s = 0.0;
#pragma unroll 8
        for (int j = 0; j < 32; j++) s += (double)red[v][seg * 32 + ((j + tid) & 31)];
        seg_f[base + v < kAccFloats ? base + v : 29][seg] = s;
      } else if (count_row) {
        double s = 0.0;
#pragma unroll 8
        for (int j = 0; j < 32; j++) s += cnt2[v - kPass][seg * 32 + ((j + tid) & 31)];
        seg_f[27 + v - kPass][seg] = s;
      }
    }
  }
  __syncthreads();
  if (tid < (HAS_EXTRA ? 30 : 29)) {
    double s = seg_f[tid][0];
#pragma unroll
    for (int k = 1; k < 8; k++) s += seg_f[tid][k];
    if (coherent) {
      // The record goes to the device's coherence point and this wave learns that it has (tail_update_wave: the pair's last
      // block reads it in this same launch, from another XCD as a rule): exchanges that RETURN — a value that has come back
      // was exchanged where atomics execute — instead of stores followed by a release fence (= a write-back of the XCD's
      // whole L2: what took the scale pass from 41 to 265 us, see k_resid_hist_v).
      const unsigned long long bits = tid == 27 ? (unsigned long long)(uint32_t)(unsigned long long)s
                                    : tid == 28 ? (unsigned long long)s : (unsigned long long)__double_as_longlong(s);
      const unsigned long long old = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(rec) + tid, bits, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(old) : "memory");   // waited for here
    } else if (tid == 27) rec[54] = (uint32_t)(unsigned long long)s;                              // valid pixels
    else if (tid == 28) reinterpret_cast<unsigned long long*>(rec)[28] = (unsigned long long)s;   // Σ r² (integer residuals)
    else reinterpret_cast<double*>(rec)[tid] = s;   // 29: Σ r·(r·w), the error numerator when residuals are not integers / weighted
  }
}

constexpr int kBatchPass = 14;   // rows per LDS pass of the block reduction in the batch kernels (two passes, 34 KB: four blocks per CU)
template <typename AccT, bool HAS_EXTRA = false, typename R2T = uint32_t>
__device__ __forceinline__ void block_reduce_store(const AccT acc[kAccFloats], R2T sum_r2, uint32_t n_valid,
                                                   uint32_t* __restrict__ rec, AccT extra = (AccT)0, bool coherent = false) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[reduce_lds_bytes(kBatchPass)];
  block_reduce_store_at<AccT, HAS_EXTRA, kBatchPass, R2T>(lds, acc, sum_r2, n_valid, rec, extra, coherent);
}

// robust weights / bilinear sampler shared by the dense kernels and the general (dump-capable) kernel
enum { kWeightsIdentity = 0, kWeightsTukeyRef = 1, kWeightsHuber = 2 };
constexpr int kHistBins = 512;  // signed residual bins q + 255 (0..510) / deviation bins 0..510

struct PairScale {
  float med0;     // median of the (saturated / signed) residuals
  float inv_mad;  // 1 / (1.4826 * median deviation), 1 / 1 when that is 0
  int n_valid;
  int pad;
};

// EXTENSION: bilinear sample; same f32 operation order as the oracle's uwo_bilinear_u8.  The four sample addresses are
// 32-bit offsets from the level's base (no 64-bit address arithmetic per sample) and clamped both ways, so that a position
// that has not been sanitised (an invalid pixel of the masked path: anything, NaN included) still reads inside the level.
__device__ __forceinline__ float sample_bilinear(const uint8_t* __restrict__ I2, const LevelK& L, float x, float y) {
  const float x0 = floorf(x), y0 = floorf(y);
  const float ax = x - x0, ay = y - y0;
  int ix0, iy0;   // (the instruction saturates and maps NaN to 0; a C++ conversion of such a value would be undefined)
  asm("v_cvt_i32_f32 %0, %1" : "=v"(ix0) : "v"(x0));
  asm("v_cvt_i32_f32 %0, %1" : "=v"(iy0) : "v"(y0));
  asm("v_med3_i32 %0, %0, 0, %1" : "+v"(ix0) : "s"(L.iw - 1));
  asm("v_med3_i32 %0, %0, 0, %1" : "+v"(iy0) : "s"(L.ih - 1));
  const int ix1 = min(ix0 + 1, L.iw - 1), iy1 = min(iy0 + 1, L.ih - 1);
  const uint32_t r0 = __umul24((unsigned)iy0, (unsigned)L.pitch), r1 = __umul24((unsigned)iy1, (unsigned)L.pitch);
  const float a = (float)I2[r0 + (unsigned)ix0], b = (float)I2[r0 + (unsigned)ix1];
  const float c = (float)I2[r1 + (unsigned)ix0], d = (float)I2[r1 + (unsigned)ix1];
  const float top = __builtin_fmaf(ax, b - a, a);
  const float bot = __builtin_fmaf(ax, d - c, c);
  return __builtin_fmaf(ay, bot - top, top);
}

// FINITE: rf is a finite residual of the alignment loop (|rf| <= 255): the Huber quotient may take the short division
template <bool FINITE = false>
__device__ __forceinline__ float robust_weight(int mode, float rf, float inv_mad) {
  if (mode == kWeightsTukeyRef) {  // Tracker::TukeyFunctionWeights, src/Tracker.cpp:1626-1654
    const float b = 4.6851f;
    const float inv_b2 = (float)(1.0 / (double)(b * b));
    const float x = rf * inv_mad;
    if (fabsf(x) <= b) {
      const float t = (float)(1.0 - (double)((x * x) * inv_b2));
      return t * t;
    }
    return 0.f;
  }
  if (mode == kWeightsHuber) {  // EXTENSION
    const float k = 1.345f;
    const float ax = fabsf(rf * inv_mad);
    // k / ax as the refined reciprocal + two residual corrections (div_by: bit-identical to the IEEE quotient for normal-range
    // operands — ax is in (1.345, 255 / MAD] where the quotient is used — at half the instructions of the compiler's sequence)
    if constexpr (FINITE) return ax <= k ? 1.0f : div_by<float>(k, ax, refined_rcp<float>(ax));
    return ax <= k ? 1.0f : k / ax;
  }
  return 1.0f;
}

// With the nearest-neighbour sampler a residual is an integer in [-255, 255], so everything the weighted accumulation
// derives from (r, scale) alone takes at most 511 values per pair and evaluation: the weight w(r), the gained and weighted
// residual (r * gain) * w that multiplies wJ (src/Tracker.cpp:559-561) and the error term r * (r * w) (:499-502).  A block
// evaluates them once per value into LDS (the same float sequence as the per-pixel form: robust_weight on (float)r, then
// the two products) and every pixel reads its entry — one 16-byte and one 8-byte LDS read in place of a float division
// (Huber) or two f64 operations (Tukey), two multiplies and three conversions.  The table occupies the bytes of the block
// reduction's image, which is not live before the loop ends.
struct WeightEntry {
  double rwd;    // (double)((r * gain) * w)
  double e;      // (double)r * (double)(r * w): the pixel's term of the error numerator (exact product)
  float w;       // robust weight of this residual value
  float pad[3];
};
static_assert(sizeof(WeightEntry) == 32, "one ds_read_b32 (w) and, later, one ds_read_b128 (rwd, e) per pixel");
constexpr int kWeightTabBytes = 511 * (int)sizeof(WeightEntry);

__device__ __forceinline__ void fill_weight_table(WeightEntry* __restrict__ tab, int mode, float inv_mad, float gain) {
  for (int i = threadIdx.x; i < 511; i += kBlock) {
    const float rf = (float)(i - 255);
    const float w = robust_weight(mode, rf, inv_mad);
    const float rw1 = rf * w;
    WeightEntry t;
    t.w = w; t.pad[0] = t.pad[1] = t.pad[2] = 0.f;
    t.rwd = (double)((rf * gain) * w);
    t.e = (double)rf * (double)rw1;
    tab[i] = t;
  }
}

// Tracker::MedianMat / MedianAbsoluteDeviation / IdentityWeights / TukeyFunctionWeights on an explicit N x 1 residual
// vector (src/Tracker.cpp:1571-1654).  One block: a 256-bin LDS histogram of the saturated, rounded values gives the
// median by the reference's rule (first bin whose cumulative count exceeds float(n / 2)), a second histogram of the
// deviations gives the MAD, then the weights.  out_stats: [median, 1.4826 * median deviation].
static __global__ __launch_bounds__(1024) void k_robust_weights(const float* __restrict__ r, int n, int kind, float* __restrict__ w,
                                                         float* __restrict__ out_stats) {
  __shared__ unsigned int hist[256];
  __shared__ float s_med, s_mad;
  auto median_of_hist = [&]() {   // thread 0
    const float m = (float)(n / 2);
    unsigned int bin = 0;
    float med = -1.0f;
    for (int i = 0; i < 256 && med < 0.0f; ++i) {
      bin += hist[i];
      if ((float)bin > m) med = (float)i;
    }
    return med;
  };
  for (int pass = 0; pass < 2; pass++) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float v = pass == 0 ? r[i] : fabsf(r[i] - s_med);
      int q = (int)rintf(v);   // cvRound: round half to even, then saturate_cast<uchar>
      q = q < 0 ? 0 : (q > 255 ? 255 : q);
      atomicAdd(&hist[q], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float med = median_of_hist();
      if (pass == 0) s_med = med; else s_mad = 1.4826f * med;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out_stats[0] = s_med; out_stats[1] = s_mad; }
  if (!w) return;
  float mad = s_mad;
  if (mad == 0.0f) mad = 1.0f;
  const float inv_mad = (float)(1.0 / (double)mad);
  for (int i = threadIdx.x; i < n; i += blockDim.x) w[i] = robust_weight(kind, r[i], inv_mad);
}

// ------------------------------------------------------------------------------------------------------------
// k_residual: one launch = one Gauss-Newton residual evaluation of one pyramid level for a whole batch.
// grid = (slices, pairs); each block walks `groups_per_block` groups of VEC consecutive pixels of its pair's
// reference level (implicit dense point table), gathers the target level, accumulates in registers and writes
// one partial record.  Reference planes are read with one VEC-wide load per plane per group.
// ------------------------------------------------------------------------------------------------------------
// The update of an evaluation in the tail of the evaluation's own launch (tail_update_wave) instead of a k_gn_update launch.
struct TailUpdate {
  unsigned int* tickets;    // one word per pair (indexed like `state`), zero between launches
  PairState* state;         // the pairs' states, writable
  int* active;              // see UpdateArgs
  int on;                   // 0: the launch leaves its records to k_gn_update
  int k, max_iters, early_exit, general, legacy_solve;
  float epsilon, gain;
};

struct ResidualArgs {
  const uint8_t* img;       // level plane of all frame slots: [slot][n]
  const int16_t* gx;
  const int16_t* gy;
  const uint16_t* depth;    // nullptr when !DEPTH
  const int* ref_slots;
  const int* tgt_slots;
  const PairState* state;   // nullptr: use `pose` (per-stage entry point)
  Pose pose;
  LevelK L;
  float zf, af;
  int groups_per_block;
  int slices;
  int pair_base;            // first pair of this launch (0 for a whole batch)
  uint32_t* partials;       // [pair][slice][kRecWords]
  const PairScale* scale;   // robust-weight scale per pair (WEIGHTS != 0)
  float gain;               // residual gain, applied inside the kernel on the weighted / bilinear path (src/Tracker.cpp:559)
  float* dumpJ;             // optional per-pixel dumps (DUMP only)
  float* dumpR;
  uint8_t* dumpV;
  float* dumpW;             // per-pixel robust weights (general path only)
  int typed_loads;          // 1: the launch takes the TYPED instantiation where one exists (uwt_tuning::typed_loads; load_group_typed)
  int stream_planes;        // 1: the launch takes the STREAM instantiation (load_group) where one exists: the batch's planes of this level exceed the caches
  int probe;                // 1: thread 0 of every block leaves its shader-clock / 100 MHz real-time deltas in words 60, 61
                            // of the block's record (uwt_profile_clock: the clock the chip holds under this kernel)
  TailUpdate tail;
};

// What a caller that evaluates a level inside its own launch (k_coarse) changes of a level's ResidualArgs — handed over beside
// the arguments instead of in a modified copy: the copy (200 bytes, no longer backed by the kernel-argument segment) would have
// to live in registers for the whole launch.
// The caller evaluates the whole level as ONE slice of its pair: `rec` is the record itself (the caller's LDS), `scale` (robust
// weights) the pair's scale itself — neither is indexed by the pair.
struct CoreOverride {
  int groups_per_block;
  uint32_t* rec;
  const PairScale* scale = nullptr;   // k_coarse_weighted: in LDS
};

typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
// reference planes of one group of VEC pixels, as loaded (one vector load per plane)
template <int VEC>
struct RefGroup {
  uint8_t i1[VEC];
  int16_t gx[VEC], gy[VEC];
  uint16_t dp[VEC];
  f4v gx4, gy4, dp4;    // TYPED (VEC = 4): the same values as floats, converted by the texture path (load_group_typed)
};

// STREAM (the loads of the loop; a template parameter of everything down to here): bit 0 = non-temporal plane loads (below), bit 1 =
// TYPED plane loads (load_group_typed).
constexpr int kLoadsPlain = 0, kLoadsStream = 1, kLoadsTyped = 2;
template <int VEC, bool DEPTH, bool COMPUTE_ONLY = false, int STREAM = 0>
__device__ __forceinline__ void load_group(RefGroup<VEC>& r, const uint8_t* __restrict__ I1, const int16_t* __restrict__ GX,
                                           const int16_t* __restrict__ GY, const uint16_t* __restrict__ DP, uint32_t idx) {
  constexpr bool FAKE_PLANES = COMPUTE_ONLY;
  if constexpr (FAKE_PLANES) {  // diagnostic instantiation: plane values made up from the index, no memory operation
#pragma unroll
    for (int j = 0; j < VEC; j++) { r.i1[j] = (uint8_t)(idx + j); r.gx[j] = (int16_t)(idx * 3 + j); r.gy[j] = (int16_t)(idx * 5 - j); r.dp[j] = (uint16_t)(4000 + (idx & 255)); }
    return;
  }
  if constexpr (VEC == 4) {
    // byte offsets in 32 bits (a level plane of one frame is < 2^31 bytes): uniform base + 32-bit lane offset addressing,
    // no 64-bit address arithmetic per load
    const uint32_t o2 = idx * 2u;
    // STREAM: a batch whose planes of this level exceed the caches reads them exactly once per evaluation — streamed (nt), so
    // that they do not displace the target level's lines, which the gathers of neighbouring lanes and steps do re-use, from the
    // 32 KB L1 (a step of a CU's 16 waves streams 28 KB of planes).  Same-box A/B (round 4): default batch +0.9 %, Huber at 256
    // pairs +3 %.  A batch that fits the 256 MB memory-side cache keeps the plain loads (its planes come back from there at the
    // next evaluation; streamed: 64 pairs -3.6 %).  A template parameter, not a flag: the compiler merges the two sides of a
    // run-time choice into plain loads.
    if constexpr ((STREAM & kLoadsStream) != 0) {
      *reinterpret_cast<uint32_t*>(r.i1) = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(I1 + idx));
      *reinterpret_cast<u2v*>(r.gx) = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(reinterpret_cast<const uint8_t*>(GX) + o2));
      *reinterpret_cast<u2v*>(r.gy) = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(reinterpret_cast<const uint8_t*>(GY) + o2));
      if constexpr (DEPTH) *reinterpret_cast<u2v*>(r.dp) = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(reinterpret_cast<const uint8_t*>(DP) + o2));
    } else {
      *reinterpret_cast<uint32_t*>(r.i1) = *reinterpret_cast<const uint32_t*>(I1 + idx);
      *reinterpret_cast<uint2*>(r.gx) = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(GX) + o2);
      *reinterpret_cast<uint2*>(r.gy) = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(GY) + o2);
      if constexpr (DEPTH) *reinterpret_cast<uint2*>(r.dp) = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(DP) + o2);
    }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      r.i1[j] = I1[idx + j]; r.gx[j] = GX[idx + j]; r.gy[j] = GY[idx + j];
      if constexpr (DEPTH) r.dp[j] = DP[idx + j];
    }
  }
}

// TYPED plane loads (round 5).  The gradients and the depth are 16-bit integers in memory and floats in the arithmetic: three
// conversions per pixel on the vector ALU, which is what bounds the kernel.  A typed buffer load (tbuffer_load_format_xyzw,
// 16_16_16_16 SSCALED: the texture path's format conversion) delivers the four values of a group as floats — the same floats
// (every int16 is exact in f32; tools/ubench/tbuffer_check.hip) — for no vector instruction; the price is registers (16 bytes
// per plane and group in flight instead of 8).  The compiler has no builtin for typed loads, and loads hidden in asm
// statements are invisible to its s_waitcnt bookkeeping, so in this form EVERY vector-memory operation of the loop is an asm
// statement and the waits are written out: vmcnt retires in order, the gathers are issued ahead of the next group's planes,
// so "the gathers have landed" is vmcnt(<plane loads issued behind them>) and "the planes have landed" is vmcnt(0) at the head
// of the next step.  Each wait names the registers it releases as in-out operands: nothing that reads them can be scheduled
// above it.
// raw buffer resource of `bytes` bytes at p (uniform): stride 0, destination select xyzw; the format comes from the instruction
__device__ __forceinline__ i4v make_rsrc(const void* p, uint32_t bytes) {
  const unsigned long long b = (unsigned long long)p;
  i4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = __builtin_amdgcn_readfirstlane(0x00027FAC);
  return r;
}
struct TypedPlanes { const uint8_t* I1; i4v gx, gy, dp; };
// The request is IN PLACE: the asm's operands are in-out, so the registers of the group that is being consumed are the ones the
// next group lands in (whatever still needs an old value has copied it out before — a copy of landed data), and between this
// request and the wait that releases them no instruction touches them.
template <bool DEPTH, bool NT>
__device__ __forceinline__ void load_group_typed(RefGroup<4>& r, const TypedPlanes& P, uint32_t idx) {
  const uint32_t o2 = idx * 2u;
  uint32_t& w = *reinterpret_cast<uint32_t*>(r.i1);
  if constexpr (NT) {
    asm volatile("global_load_dword %0, %1, %2 nt" : "+v"(w) : "v"(idx), "s"(P.I1));
    asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_SSCALED] offen nt" : "+v"(r.gx4) : "v"(o2), "s"(P.gx));
    asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_SSCALED] offen nt" : "+v"(r.gy4) : "v"(o2), "s"(P.gy));
    if constexpr (DEPTH) asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_SSCALED] offen nt" : "+v"(r.dp4) : "v"(o2), "s"(P.dp));
  } else {
    asm volatile("global_load_dword %0, %1, %2" : "+v"(w) : "v"(idx), "s"(P.I1));
    asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_SSCALED] offen" : "+v"(r.gx4) : "v"(o2), "s"(P.gx));
    asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_SSCALED] offen" : "+v"(r.gy4) : "v"(o2), "s"(P.gy));
    if constexpr (DEPTH) asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_SSCALED] offen" : "+v"(r.dp4) : "v"(o2), "s"(P.dp));
  }
}
// "the planes of rg have landed": everything issued so far has (the head of a step, and behind the loop)
template <bool DEPTH>
__device__ __forceinline__ void wait_planes_typed(RefGroup<4>& r) {
  uint32_t& w = *reinterpret_cast<uint32_t*>(r.i1);
  if constexpr (DEPTH) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w), "+v"(r.gx4), "+v"(r.gy4), "+v"(r.dp4));
  else asm volatile("s_waitcnt vmcnt(0)" : "+v"(w), "+v"(r.gx4), "+v"(r.gy4));
}

// COMPUTE_ONLY (diagnostic, uwt_profile_enable(ctx, 2)): the same instruction stream with every load of the loop replaced
// by register arithmetic — results are meaningless, its duration is the kernel's own instruction-issue floor.
// residual_core evaluates one slice of one pair at `pose`; `lds` (optional) is the caller's buffer for the block reduction.
template <int AR, int VEC, bool DEPTH, bool UNIT_FACTORS, bool DUMP, typename AccT, bool SQUARE = false, int SAMPLER = 0, int WEIGHTS = 0,
          bool COMPUTE_ONLY = false, int EXT_LDS = 0, int STREAM = 0, bool RAGGED = false>   // EXT_LDS: 0 = own LDS, else rows per reduction pass in the caller's; STREAM: load_group; RAGGED: the level's grid rows are not whole groups of four (colm)
__device__ __forceinline__ void residual_core(const ResidualArgs& a, const int pair, const int slice, const Pose& pose,
                                              unsigned char* lds, const RefGroup<VEC>* first = nullptr, int ref_slot = -1,
                                              int tgt_slot = -1, const CoreOverride* ov = nullptr);

// the reference planes of the first group of (pair, slice) for this thread: what residual_core loads before anything else
template <int VEC, bool DEPTH, bool COMPUTE_ONLY>
__device__ __forceinline__ void load_first_group(RefGroup<VEC>& rg, const ResidualArgs& a, int ref_slot, int slice) {
  const size_t ref_off = (size_t)ref_slot * a.L.n;
  const int n_groups = a.L.ng / VEC;
  const int g = slice * a.groups_per_block + (int)threadIdx.x;
  load_group<VEC, DEPTH, COMPUTE_ONLY>(rg, a.img + ref_off, a.gx + ref_off, a.gy + ref_off, DEPTH ? a.depth + ref_off : nullptr,
                                       (uint32_t)min(g, n_groups - 1) * VEC);
}

template <int AR, int VEC, bool DEPTH, bool UNIT_FACTORS, bool DUMP, typename AccT, bool SQUARE = false, int SAMPLER = 0, int WEIGHTS = 0,
          bool COMPUTE_ONLY = false, int STREAM = 0, bool RAGGED = false>
__device__ __forceinline__ bool residual_block(const ResidualArgs& a, const int pair, const int slice) {   // false: the pair is not iterating
  Pose pose;
  if constexpr (COMPUTE_ONLY) {
    pose_identity(pose);   // never skips a pair, never follows the (meaningless) updates: every launch does the full work
  } else if (a.state) {
    const PairState st = a.state[pair];
    if (st.level_done || st.status) return false;
    pose = st.pose;
  } else {
    pose = a.pose;
  }
  residual_core<AR, VEC, DEPTH, UNIT_FACTORS, DUMP, AccT, SQUARE, SAMPLER, WEIGHTS, COMPUTE_ONLY, 0, STREAM, RAGGED>(a, pair, slice, pose, nullptr);
  return true;
}

template <int AR, int VEC, bool DEPTH, bool UNIT_FACTORS, bool DUMP, typename AccT, bool SQUARE, int SAMPLER, int WEIGHTS, bool COMPUTE_ONLY,
          int EXT_LDS, int STREAM, bool RAGGED>
__device__ __forceinline__ void residual_core(const ResidualArgs& a, const int pair, const int slice, const Pose& pose,
                                              unsigned char* lds, const RefGroup<VEC>* first, int ref_slot, int tgt_slot,
                                              const CoreOverride* ov) {
  const int a_groups_per_block = ov ? ov->groups_per_block : a.groups_per_block;
  const bool a_probe = ov ? false : a.probe != 0;
  // MASKED: f64 sums added under an EXEC mask of the valid lanes (masked_sums_*), nothing of an invalid pixel sanitised
  constexpr bool MASKED = std::is_same<AccT, double>::value && !DUMP;
  WarpK K;
  // kArithOpenCV: the rigid matrix as doubles lives in LDS and is read back at the head of every step of the loop (TD_LDS).  As
  // scalar registers the 12 doubles do not fit beside the kernel's ~95 (the compiler then keeps them in 24 vector registers
  // for the whole loop: 151 registers, three waves per SIMD); read per step they are live through the warp phase only,
  // where the pressure is lowest.  6 ds_read_b128 per 4 pixels, no vector-ALU instruction.
  constexpr bool TD_LDS = AR == kArithOpenCV && !DUMP;
  __shared__ __attribute__((aligned(16))) double s_td[TD_LDS ? 12 : 2];
  if constexpr (TD_LDS) {
    warp_setup<kArithLegacy>(pose, K);   // T in scalar registers (the hand-over of the block's pose); Td goes through LDS
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < 12; i++) s_td[i] = (double)K.T[i];
    }
    __syncthreads();
  } else {
    warp_setup<AR>(pose, K);   // the rigid matrix is block-uniform: scalar registers
  }
  const LevelK L = a.L;
  if (ref_slot < 0) {   // the pair list in memory (batches); k_iterate hands the slots over
    ref_slot = a.ref_slots[pair];
    tgt_slot = a.tgt_slots[pair];
  }
  const size_t ref_off = (size_t)ref_slot * L.n, tgt_off = (size_t)tgt_slot * L.n;
  const uint8_t* __restrict__ I1 = a.img + ref_off;
  const uint8_t* __restrict__ I2 = a.img + tgt_off;
  const int16_t* __restrict__ GX = a.gx + ref_off;
  const int16_t* __restrict__ GY = a.gy + ref_off;
  const uint16_t* __restrict__ DP = DEPTH ? a.depth + ref_off : nullptr;

  // the probe's start stamps wait in LDS, not in four scalar registers through the loop (the kernels sit near the scalar-register
  // limit, and what does not fit there is parked in a vector register)
  __shared__ unsigned long long s_probe[2];
  if (a_probe && threadIdx.x == 0) {
    s_probe[0] = __builtin_amdgcn_s_memtime();
    s_probe[1] = __builtin_amdgcn_s_memrealtime();
  }
  AccT acc[kAccFloats];
#pragma unroll
  for (int i = 0; i < kAccFloats; i++) acc[i] = (AccT)0;
  uint32_t sum_r2 = 0, n_valid_wave = 0;  // the valid count is kept per wave in a scalar register
  double r2d = 0.0;                       // MASKED: the sum of r^2 as a 28th f64 sum (masked_sums_hi)
  constexpr bool GENERAL = SAMPLER != 0 || WEIGHTS != 0;  // float residuals and/or robust weights
  AccT err = (AccT)0;                                      // Σ r·(r·w), the error numerator on the general path
  float inv_mad = 1.f;
  if constexpr (WEIGHTS != 0) inv_mad = ov ? ov->scale->inv_mad : a.scale[pair].inv_mad;
  // robust weights over integer residuals: the per-value table (see WeightEntry) in the bytes of the reduction's image
  constexpr bool TABLE = WEIGHTS != 0 && SAMPLER == 0;
  static_assert(!TABLE || EXT_LDS == 0, "the weighted path reduces in its own LDS");
  __shared__ __attribute__((aligned(16))) unsigned char tlds[TABLE ? kReduceLdsBytes : 16];
  static_assert(!TABLE || kWeightTabBytes <= kReduceLdsBytes, "table fits the reduction image");
  if constexpr (TABLE) {
    fill_weight_table(reinterpret_cast<WeightEntry*>(tlds), WEIGHTS, inv_mad, a.gain);
    __syncthreads();
  }

  const int n_groups = L.ng / VEC;
  const int g_begin = slice * a_groups_per_block;
  const int g_end = min(g_begin + a_groups_per_block, n_groups);
  const int iters = (g_end - g_begin + kBlock - 1) / kBlock;  // block-uniform trip count

  // Software pipeline over groups.  `rg` holds the reference planes of the group being processed and is re-requested IN
  // PLACE for the thread's next group as soon as its last value has been consumed — after this group's gathers have been
  // issued, not before: vmcnt retires in order, so a wait for the gathered bytes would otherwise also wait for the plane
  // loads requested ahead of them, which always come cold from HBM.  Only the reference intensities are copied out (they
  // are needed last, for the residuals).
  RefGroup<VEC> rg;
  int g = g_begin + (int)threadIdx.x;
  // TYPED: the planes' 16-bit values arrive as floats (load_group_typed): no conversion on the vector ALU
  constexpr bool TYPED = (STREAM & kLoadsTyped) != 0;
  constexpr bool NT = (STREAM & kLoadsStream) != 0;
  // (measured on the weighted path and on the general Jacobian form too: no gain there — profiles/r05/EXPERIMENTS.md)
  static_assert(!TYPED || (VEC == 4 && std::is_same<AccT, double>::value && !DUMP && !COMPUTE_ONLY && SAMPLER == 0 && WEIGHTS == 0),
                "typed plane loads: the identity path's production shape");
  TypedPlanes TP;
  if constexpr (TYPED) {
    TP.I1 = I1;
    TP.gx = make_rsrc(GX, (uint32_t)L.n * 2u);
    TP.gy = make_rsrc(GY, (uint32_t)L.n * 2u);
    TP.dp = make_rsrc(DEPTH ? (const void*)DP : (const void*)GX, (uint32_t)L.n * 2u);
    *reinterpret_cast<uint32_t*>(rg.i1) = 0u;
    rg.gx4 = rg.gy4 = rg.dp4 = (f4v)(0.f);
    load_group_typed<DEPTH, NT>(rg, TP, (uint32_t)min(g, n_groups - 1) * VEC);
  } else if (first) rg = *first;   // requested by the caller ahead of the pose (k_iterate: before the update)
  else load_group<VEC, DEPTH, COMPUTE_ONLY, STREAM>(rg, I1, GX, GY, DP, (uint32_t)min(g, n_groups - 1) * VEC);
  // Pixel coordinates of the thread's group, as floats (small integers: exact).  One division up front, then each step
  // of kBlock groups moves (x, y) by the level's fixed (step mod w, step / w) with at most one wrap.  Lanes past the end
  // of the level run on with coordinates outside the image; they are inactive and every term of theirs is discarded.
  const uint32_t idx0 = (uint32_t)g * VEC;
  const uint32_t y0 = __umulhi(idx0, L.magic);
  float yf = (float)y0, xf0 = (float)(idx0 - y0 * (uint32_t)L.pitch);
  const uint32_t step_y = __umulhi((uint32_t)(kBlock * VEC), L.magic);
  const float step_yf = (float)step_y, step_xf = (float)((uint32_t)(kBlock * VEC) - step_y * (uint32_t)L.pitch), wf = (float)L.pitch;
  // Pixels are processed in units of N adjacent ones (N = 2: the packed-f32 form, see v2f above), all VEC pixels of the
  // group in flight through four phases: warp + validity + gather index, gather, Jacobian, residual + accumulation.
  constexpr int N = (VEC % 2 == 0) ? 2 : 1;
  using F = typename std::conditional<N == 2, v2f, float>::type;
  constexpr int NU = VEC / N;
  // one step of the loop on the planes in `rg`, which are re-requested in place for the group `ahead` steps on
  // (`nxt`: where the next group's planes are requested — `rg` itself, or the other register set of the TYPED form)
  auto body = [&](RefGroup<VEC>& rg, const int ahead, RefGroup<VEC>& nxt) __attribute__((always_inline)) {
    const bool active = g < g_end;
    const unsigned long long active_mask = __builtin_amdgcn_sicmp(g, g_end, kIcmpSLT);
    constexpr bool TD_READ = TD_LDS;
    if constexpr (TD_READ) {
      unsigned off = 0;
      asm volatile("" : "+v"(off));   // an offset the compiler cannot see through: the reads stay inside the loop
      const double* tdp = reinterpret_cast<const double*>(reinterpret_cast<const unsigned char*>(s_td) + off);
#pragma unroll
      for (int i = 0; i < 12; i++) K.Td[i] = tdp[i];
    }
    const uint32_t idx = (uint32_t)min(g, n_groups - 1) * VEC;
    if constexpr (TYPED) wait_planes_typed<DEPTH>(rg);   // this group's planes, requested a step ago, have landed
    uint8_t i1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) i1[j] = rg.i1[j];
    // phase 1: warp, validity, gather index
    F x2[NU], y2[NU], iz[NU];
    unsigned long long okm[VEC];  // validity as a wave mask (SGPR pair)
    uint32_t gidx[VEC];
    // Which pixels of the group are points of the level's grid.  A level whose grid rows are whole groups of four (gw == pitch)
    // has no other position.  RAGGED — every level of a frame size that is not a multiple of 2^(levels-1) x 4: the ROI crops of
    // src/System.cpp:148-191 — : the last group of a row holds positions x >= gw of the pitched row, which carry no point
    // (src/Tracker.cpp:1267-1268 walks x < w_[lvl]).  Groups start at multiples of four, so pixel j of a group is a point iff
    // x0 < gw - (gw & 3) ("every pixel of the group is") or j < (gw & 3) and x0 < gw ("the row's last, partial group"): two compares
    // per group; the whole-level instantiations carry nothing of it.
    unsigned long long colm[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) colm[j] = active_mask;
    if constexpr (RAGGED) {
      const unsigned long long m_in = active_mask & __builtin_amdgcn_fcmpf(xf0, (float)L.gw, kFcmpOLT);
      const unsigned long long m_all = active_mask & __builtin_amdgcn_fcmpf(xf0, (float)(L.gw & ~3), kFcmpOLT);
      const int rem = L.gw & 3;
#pragma unroll
      for (int j = 0; j < VEC; j++) colm[j] = (VEC == 1 || j < rem) ? m_in : m_all;
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
      F z = bc<F>(1.0f), xf;
      unsigned long long okin[N];
      float dlow[N];   // MASKED with depth: the depth as a float (its "> 0" test joins the lower bounds below)
#pragma unroll
      for (int c = 0; c < N; c++) {
        const int j = u * N + c;
        okin[c] = colm[j];
        dlow[c] = 1.0f;
        if constexpr (DEPTH && TYPED) {
          const float d = rg.dp4[j];              // the same signed 16-bit value, converted by the load
          dlow[c] = d;
          put(z, c, d);
        } else if constexpr (DEPTH) {
          const int d = (int)(int16_t)rg.dp[j];   // depths_ is read through at<short> (src/Tracker.cpp:1272)
          dlow[c] = (float)d;
          if constexpr (!MASKED) okin[c] &= __builtin_amdgcn_sicmp(d, 0, kIcmpSGT);
          put(z, c, dlow[c]);
        }
        put(xf, c, (float)j);
      }
      if constexpr (DEPTH) z = z * bc<F>(L.zscale);
      xf = bc<F>(xf0) + xf;
      if constexpr (MASKED) {
        if constexpr (DEPTH) {
          // The three lower bounds — depth > 0 (:1268), y2 > 0, x2 > 0 (:450) — as ONE compare of their minimum.  A NaN position
          // (z2 = 0: both quotients are NaN, see pixel_warp_raw) drops out of the minimum, which the depth then decides — and
          // fails the ordered upper-bound compares below, as it fails every compare of the reference's test; -0 and +0 are not
          // greater than 0 either way.
          F z2;
          warp_point<AR, F>(L, K, xf, bc<F>(yf), z, x2[u], y2[u], z2, iz[u]);
#pragma unroll
          for (int c = 0; c < N; c++) {
            const float uc = get(x2[u], c), vc = get(y2[u], c);
            float lo;
            asm("v_min3_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(uc), "v"(vc), "v"(dlow[c]));
            okm[u * N + c] = okin[c] & __builtin_amdgcn_fcmpf(lo, 0.f, kFcmpOGT) & __builtin_amdgcn_fcmpf(vc, (float)L.ih, kFcmpOLT) &
                             __builtin_amdgcn_fcmpf(uc, (float)L.iw, kFcmpOLT);
          }
        } else {
          pixel_warp_raw<AR, F>(L, K, xf, bc<F>(yf), z, okin, x2[u], y2[u], iz[u], &okm[u * N]);
        }
#pragma unroll
        for (int c = 0; c < N; c++) {
          float r = get(iz[u], c);
          asm("v_max_f32 %0, 0, %0" : "+v"(r));   // "if (inv_z2 < 0) inv_z2 = 0" (:452-453); what an invalid pixel gets does not matter
          put(iz[u], c, r);
          int ix2 = round_pos(get(x2[u], c)), iy2 = round_pos(get(y2[u], c));
          // x2, y2 are not sanitised here: clamp both ways, in one instruction (the compiler keeps min and max apart)
          asm("v_med3_i32 %0, %0, 0, %1" : "+v"(ix2) : "s"(L.iw - 1));
          asm("v_med3_i32 %0, %0, 0, %1" : "+v"(iy2) : "s"(L.ih - 1));
          gidx[u * N + c] = __umul24((unsigned)iy2, (unsigned)L.pitch) + (unsigned)ix2;
        }
      } else {
        pixel_warp<AR, F>(L, K, xf, bc<F>(yf), z, okin, x2[u], y2[u], iz[u], &okm[u * N], &gidx[u * N]);
      }
    }
    // phase 2: the samples of the target level
    int i2[VEC];
    float s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      constexpr bool FAKE_GATHER = COMPUTE_ONLY;
      if constexpr (FAKE_GATHER) i2[j] = (int)i1[j] + (int)(gidx[j] & 1);
      else if constexpr (TYPED) asm volatile("global_load_ubyte %0, %1, %2" : "=v"(i2[j]) : "v"(gidx[j]), "s"(I2));   // (waited for by gathers_landed)
      else if constexpr (SAMPLER == 0) i2[j] = I2[gidx[j]];   // nearest-neighbour gather of the target level (:472)
      else s2[j] = sample_bilinear(I2, L, get(x2[j / N], j % N), get(y2[j / N], j % N));  // EXTENSION; x2 = y2 = 0 for sanitised invalid pixels
    }
    // the gradients leave rg here, so that it can be re-requested
    F g0[NU], g1[NU];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      if constexpr (MASKED) {
        float gxf, gyf;
        if constexpr (TYPED) { gxf = rg.gx4[j]; gyf = rg.gy4[j]; }
        else { gxf = (float)rg.gx[j]; gyf = (float)rg.gy[j]; }
        if constexpr (AR == kArithOpenCV && !TYPED) {
          // the gradient is needed as f32 (columns 0, 1) and as f64 (columns 2..5).  Opaque here, so that the double is widened
          // from the float (one conversion with the 16-bit extraction folded in — SDWA — and one widening) instead of being
          // converted from the integer a second time, which costs a separate sign extension per value.
          asm("" : "+v"(gxf));
          asm("" : "+v"(gyf));
        }
        put(g0[j / N], j % N, gxf);
        put(g1[j / N], j % N, gyf);
      } else {
        put(g0[j / N], j % N, keep_f((float)rg.gx[j], okm[j]));
        put(g1[j / N], j % N, keep_f((float)rg.gy[j], okm[j]));
      }
    }
    // The scheduling fences keep the requests where they are written: left alone the scheduler sinks them to the end of
    // the body (shorter live ranges), and the residual subtractions — the first use of the gathered bytes — rise to just
    // behind the gathers.  The request is unconditional (the index is clamped): inside a branch, the compiler's wait for
    // the gathers would have to assume the branch not taken and count the plane loads in.
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TYPED) load_group_typed<DEPTH, NT>(nxt, TP, (uint32_t)min(g + ahead * kBlock, n_groups - 1) * VEC);
    else load_group<VEC, DEPTH, COMPUTE_ONLY, STREAM>(nxt, I1, GX, GY, DP, (uint32_t)min(g + ahead * kBlock, n_groups - 1) * VEC);
    __builtin_amdgcn_sched_barrier(0);
    // TYPED: "the gathers have landed" = all but the plane loads issued behind them have (vmcnt retires in order)
    auto gathers_landed = [&]() __attribute__((always_inline)) {
      if constexpr (TYPED) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(i2[0]), "+v"(i2[1]), "+v"(i2[2]), "+v"(i2[3]) : "n"(DEPTH ? 4 : 3));
    };
    // kArithOpenCV, identity path: a unit's row is formed in double and goes straight into the sums, one unit at a time (no f32
    // rows of all four pixels held across the phase: the doubles of the small products take their registers)
    constexpr bool DIRECT = AR == kArithOpenCV && MASKED && !GENERAL;
    if constexpr (DIRECT) {
#pragma unroll
      for (int u = 0; u < NU; u++) {
        F a0, b1, av[4], bv[4];
        jw_terms<UNIT_FACTORS, SQUARE, F>(L, a.zf, a.af, x2[u], y2[u], iz[u], a0, b1, av, bv);
        const F j0 = g0[u] * a0, j1 = g1[u] * b1;
#pragma unroll
        for (int c = 0; c < N; c++) {
          const int j = u * N + c;
          double Jd[6];
          jacobian_row_f64<SQUARE, F>(g0[u], g1[u], j0, j1, av, bv, c, Jd);
          if (u == 0 && c == 0) gathers_landed();
          const int ri = i2[j] - (int)i1[j];
          masked_sums_lo(acc, Jd, okm[j]);
          masked_sums_hi<0>(acc, r2d, Jd, (double)ri, 0.0, okm[j]);
          n_valid_wave += (uint32_t)__builtin_popcountll(okm[j]);  // scalar
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if constexpr (MASKED && GENERAL && !TABLE) {
      // float residuals (bilinear sampler) and / or weights evaluated per pixel: the same one-unit-at-a-time form —
      // a pixel's row is formed, weighted and added before the next one's is begun, so that no f32 rows of all four pixels sit
      // beside the doubles of the small products (the instantiation with the most live values: bilinear + Huber)
#pragma unroll
      for (int u = 0; u < NU; u++) {
        if constexpr (SAMPLER != 0) {
          // every sample finished here, ahead of the Jacobian terms: four floats stay,
          // not the sixteen bytes and eight fractions of interpolations the scheduler would otherwise leave pending into the sums
          if (u == 0) {
#pragma unroll
            for (int j = 0; j < VEC; j++) asm volatile("" : "+v"(s2[j]));
          }
        }
        F a0, b1, av[4], bv[4], j0, j1, Ju[6];
        if constexpr (AR == kArithOpenCV) {
          jw_terms<UNIT_FACTORS, SQUARE, F>(L, a.zf, a.af, x2[u], y2[u], iz[u], a0, b1, av, bv);
          j0 = g0[u] * a0;
          j1 = g1[u] * b1;
        } else {
          pixel_jacobian<AR, UNIT_FACTORS, SQUARE, false, F>(L, a.zf, a.af, x2[u], y2[u], iz[u], g0[u], g1[u], Ju);
        }
#pragma unroll
        for (int c = 0; c < N; c++) {
          const int j = u * N + c;
          float Jp[6];
          if constexpr (AR == kArithOpenCV) jacobian_row_f32<SQUARE, F>(g0[u], g1[u], j0, j1, av, bv, c, Jp);
          else {
#pragma unroll
            for (int k = 0; k < 6; k++) Jp[k] = get(Ju[k], c);
          }
          float rf;
          if constexpr (SAMPLER == 0) rf = (float)(i2[j] - (int)i1[j]);
          else rf = s2[j] - (float)i1[j];
          const float w = robust_weight<true>(WEIGHTS, rf, inv_mad);
          double Jd[6];
#pragma unroll
          for (int k = 0; k < 6; k++) Jd[k] = (double)(w * Jp[k]);
          const double e = (double)rf * (double)(rf * w);
          masked_sums_lo(acc, Jd, okm[j]);
          masked_sums_hi<1>(acc, err, Jd, (double)((rf * a.gain) * w), e, okm[j]);
          n_valid_wave += (uint32_t)__builtin_popcountll(okm[j]);  // scalar
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    // phase 3: Jacobians (cover the gather latency)
    F J[NU][6];
#pragma unroll
    for (int u = 0; u < NU; u++)
      pixel_jacobian<AR, UNIT_FACTORS, SQUARE, DUMP, F>(L, a.zf, a.af, x2[u], y2[u], iz[u], g0[u], g1[u], J[u]);
    __builtin_amdgcn_sched_barrier(0);
    gathers_landed();
    // phase 4: residuals and accumulation
    if constexpr (TABLE) {
      // weights from the per-value table; J <- w * J over packed pairs (src/Tracker.cpp:554-557)
      // (one packed unit at a time, the table offset computed once per pixel: short live ranges keep the kernel at 4 waves / SIMD)
#pragma unroll
      for (int u = 0; u < NU; u++) {
        uint32_t off[N];
        F wv;
#pragma unroll
        for (int c = 0; c < N; c++) {
          const int j = u * N + c;
          const int ri = i2[j] - (int)i1[j];   // in [-255, 255] whatever the pixel's validity: a table index either way
          n_valid_wave += (uint32_t)__builtin_popcountll(okm[j]);
          off[c] = (uint32_t)(ri + 255) * (uint32_t)sizeof(WeightEntry);
          put(wv, c, *reinterpret_cast<const float*>(tlds + off[c] + 16));
        }
#pragma unroll
        for (int k = 0; k < 6; k++) J[u][k] = wv * J[u][k];
#pragma unroll
        for (int c = 0; c < N; c++) {
          const int j = u * N + c;
          const double2 re = *reinterpret_cast<const double2*>(tlds + off[c]);   // rwd, e
          if constexpr (MASKED) {
            double Jd[6];
#pragma unroll
            for (int k = 0; k < 6; k++) Jd[k] = (double)get(J[u][k], c);
            masked_sums_lo(acc, Jd, okm[j]);
            masked_sums_hi<1>(acc, err, Jd, re.x, re.y, okm[j]);
          } else {
            float Jp[6];
#pragma unroll
            for (int k = 0; k < 6; k++) Jp[k] = get(J[u][k], c);
            const unsigned long long m = okm[j];
            if (lane_bit(m)) {
              err += (AccT)re.y;
              accumulate_preweighted(acc, Jp, re.x);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      float Jp[6];
#pragma unroll
      for (int k = 0; k < 6; k++) Jp[k] = get(J[j / N][k], j % N);
      int ri = 0;
      if constexpr (!GENERAL && MASKED) {
        ri = i2[j] - (int)i1[j];
        double Jd[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Jd[k] = (double)Jp[k];
        masked_sums_lo(acc, Jd, okm[j]);
        masked_sums_hi<0>(acc, r2d, Jd, (double)ri, 0.0, okm[j]);
      } else if constexpr (!GENERAL) {
        ri = keep_i(i2[j] - (int)i1[j], okm[j]);
        accumulate(acc, Jp, ri);
      } else if constexpr (MASKED) {
        // float residuals (bilinear sampler) and / or weights evaluated per pixel: the same masked sums, fed with w * J,
        // (r * gain) * w and the error term r * (r * w) (accumulate_weighted's operations, src/Tracker.cpp:499-502, 554-561)
        float rf;
        if constexpr (SAMPLER == 0) rf = (float)(i2[j] - (int)i1[j]);
        else rf = s2[j] - (float)i1[j];
        const float w = robust_weight<true>(WEIGHTS, rf, inv_mad);
        double Jd[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Jd[k] = (double)(w * Jp[k]);
        const double e = (double)rf * (double)(rf * w);
        masked_sums_lo(acc, Jd, okm[j]);
        masked_sums_hi<1>(acc, err, Jd, (double)((rf * a.gain) * w), e, okm[j]);
      } else {
        float rf;
        if constexpr (SAMPLER == 0) rf = (float)keep_i(i2[j] - (int)i1[j], okm[j]);
        else rf = keep_f(s2[j] - (float)i1[j], okm[j]);
        const float w = robust_weight(WEIGHTS, rf, inv_mad);
        accumulate_weighted(acc, err, Jp, rf, w, a.gain);
        ri = (int)rintf(rf);
      }
      if constexpr (GENERAL || !MASKED) sum_r2 += (uint32_t)__mul24(ri, ri);  // |ri| <= 255 (masked path: r2d)
      n_valid_wave += (uint32_t)__builtin_popcountll(okm[j]);  // scalar
      if constexpr (DUMP) {
        if (active) {
          const size_t p = (size_t)pair * L.ng + idx + j;
          if (a.dumpV) a.dumpV[p] = lane_bit(okm[j]) ? 1 : 0;
          if (a.dumpR) a.dumpR[p] = (float)ri;
          if (a.dumpJ)
            for (int k = 0; k < 6; k++) a.dumpJ[p * 6 + k] = Jp[k];
        }
      }
    }
    }
    }   // !DIRECT
    xf0 += step_xf;
    yf += step_yf;
    const bool wrap = xf0 >= wf;
    xf0 -= wrap ? wf : 0.f;
    yf += wrap ? 1.f : 0.f;
  };
  if constexpr (TYPED) {
    // Two register sets take turns: while the floats of one are being consumed (the gradients live until the step's Jacobians)
    // the next group lands in the other, whose contents are dead by then — requested in place, no copy of anything in flight
    // and none of anything landed.  (One set re-requested in place would have to copy the eight gradient floats out first:
    // what the conversions used to do for nothing.)
    RefGroup<VEC> rgB;
    *reinterpret_cast<uint32_t*>(rgB.i1) = 0u;
    rgB.gx4 = rgB.gy4 = rgB.dp4 = (f4v)(0.f);
    for (int it = 0; it < iters; it += 2) {
      body(rg, 1, rgB);
      g += kBlock;
      if (it + 1 < iters) {   // block-uniform
        body(rgB, 1, rg);
        g += kBlock;
      }
    }
    wait_planes_typed<DEPTH>(rg);    // the last step's request (a clamped, unused group) must not land in registers that have moved on
    wait_planes_typed<DEPTH>(rgB);
  } else {
    for (int it = 0; it < iters; it++, g += kBlock) body(rg, 1, rg);
  }
  const uint32_t n_valid = (threadIdx.x & 63) == 0 ? n_valid_wave : 0u;
  uint32_t* out_rec = ov ? ov->rec : a.partials + ((size_t)pair * a.slices + slice) * kRecWords;
  constexpr bool R2D = MASKED && !GENERAL;   // the identity path's sum of r^2 is the f64 one
  const bool coherent = !ov && a.tail.on != 0;   // the record is read in this launch (tail_update_wave)
  if constexpr (EXT_LDS != 0 && R2D) block_reduce_store_at<AccT, false, EXT_LDS, double>(lds, acc, r2d, n_valid, out_rec, err);   // the caller's bytes: k_iterate
  else if constexpr (EXT_LDS != 0) block_reduce_store_at<AccT, GENERAL, EXT_LDS>(lds, acc, sum_r2, n_valid, out_rec, err);
  else if constexpr (TABLE) {
    __syncthreads();   // every wave has read its last table entry: the bytes become the reduction's image
    block_reduce_store_at<AccT, true>(tlds, acc, sum_r2, n_valid, out_rec, err, coherent);
  } else if constexpr (R2D) block_reduce_store<AccT, false, double>(acc, r2d, n_valid, out_rec, err, coherent);
  else block_reduce_store<AccT, GENERAL>(acc, sum_r2, n_valid, out_rec, err, coherent);
  if (a_probe && threadIdx.x == 0) {
    uint32_t* rec = a.partials + ((size_t)pair * a.slices + slice) * kRecWords;
    rec[60] = (uint32_t)(__builtin_amdgcn_s_memtime() - s_probe[0]);
    rec[61] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - s_probe[1]);
  }
}

__device__ __forceinline__ void tail_update_wave(const ResidualArgs& a, int pair);   // (behind update_solve_wave)

template <int AR, int VEC, bool DEPTH, bool UNIT_FACTORS, bool DUMP, typename AccT, bool SQUARE = false, int SAMPLER = 0, int WEIGHTS = 0,
          bool COMPUTE_ONLY = false, int STREAM = 0, bool RAGGED = false>
__global__ __launch_bounds__(kBlock) void k_residual(const ResidualArgs a) {
  const int pair = (int)blockIdx.y + a.pair_base;
  const bool live = residual_block<AR, VEC, DEPTH, UNIT_FACTORS, DUMP, AccT, SQUARE, SAMPLER, WEIGHTS, COMPUTE_ONLY, STREAM, RAGGED>(a, pair, (int)blockIdx.x);
  if constexpr (!COMPUTE_ONLY && !DUMP) {
    if (a.tail.on && live && threadIdx.x < 64) tail_update_wave(a, pair);   // wave 0 wrote the block's record
  }
}

// The same kernel held to four waves per SIMD (128 registers), for the one instantiation whose allocation lands just above
// (bilinear sampler + Huber: 129 — the 129th register holds scalar registers the compiler parks across the loop).
template <int AR, int VEC, bool DEPTH, bool UNIT_FACTORS, bool DUMP, typename AccT, bool SQUARE = false, int SAMPLER = 0, int WEIGHTS = 0,
          int STREAM = 0, bool RAGGED = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_residual_w4(const ResidualArgs a) {
  const int pair = (int)blockIdx.y + a.pair_base;
  const bool live = residual_block<AR, VEC, DEPTH, UNIT_FACTORS, DUMP, AccT, SQUARE, SAMPLER, WEIGHTS, false, STREAM, RAGGED>(a, pair, (int)blockIdx.x);
  if constexpr (!DUMP) {
    if (a.tail.on && live && threadIdx.x < 64) tail_update_wave(a, pair);
  }
}

// ------------------------------------------------------------------------------------------------------------
// General residual path: robust weights (src/Tracker.cpp:495-496, 1571-1654) and the bilinear sampler extension.
// Not the throughput path — one pixel per thread step, three passes per iteration when weights are on
// (residual histogram -> median, deviation histogram -> MAD, weighted accumulation).
// ------------------------------------------------------------------------------------------------------------
struct GeneralArgs {
  int sampler;            // 0 nearest, 1 bilinear
  int weights;            // kWeights*
  int stage;              // unused (kept for layout)
  float gain;
  unsigned int* hist;     // [pair][kHistBins]: signed bins q + 255 of the rounded residuals
  PairScale* scale;       // [pair]
};

// one pixel of the dense table: warp, validity, residual (either sampler)
template <int AR, bool DEPTH>
__device__ __forceinline__ bool general_pixel(const LevelK& L, const WarpK& K, int sampler,
                                              const uint8_t* I1, const uint8_t* I2, const uint16_t* DP, uint32_t idx,
                                              float& x2, float& y2, float& iz, float& rf) {
  const uint32_t y = __umulhi(idx, L.magic), x = idx - y * (uint32_t)L.pitch;
  float z = 1.0f;
  bool ok = x < (uint32_t)L.gw;   // positions of the pitched row beyond the point grid carry no point
  if constexpr (DEPTH) {
    const int d = (int)(int16_t)DP[idx];
    ok = ok && d > 0;
    z = (float)d * L.zscale;
  }
  uint32_t gidx;
  bool valid;
  unsigned long long okm;
  const unsigned long long okin = __builtin_amdgcn_ballot_w64(ok);
  pixel_warp<AR, float>(L, K, (float)x, (float)y, z, &okin, x2, y2, iz, &okm, &gidx);
  valid = lane_bit(okm);
  const int i1 = I1[idx];
  rf = sampler ? sample_bilinear(I2, L, x2, y2) - (float)i1 : (float)((int)I2[gidx] - i1);
  return valid;
}

template <int AR, bool DEPTH>
__global__ __launch_bounds__(kBlock) void k_resid_hist(const ResidualArgs a, const GeneralArgs ga) {
  const int pair = blockIdx.y + a.pair_base;
  const PairState st = a.state[pair];
  if (st.level_done || st.status) return;
  __shared__ unsigned int h[kHistBins];
  for (int i = threadIdx.x; i < kHistBins; i += kBlock) h[i] = 0;
  __syncthreads();
  WarpK K;
  warp_setup<AR>(st.pose, K);
  const LevelK L = a.L;
  const size_t ref_off = (size_t)a.ref_slots[pair] * L.n, tgt_off = (size_t)a.tgt_slots[pair] * L.n;
  const uint8_t* I1 = a.img + ref_off;
  const uint8_t* I2 = a.img + tgt_off;
  const uint16_t* DP = DEPTH ? a.depth + ref_off : nullptr;
  const int p_begin = blockIdx.x * a.groups_per_block, p_end = min(p_begin + a.groups_per_block, L.ng);
  for (int p = p_begin + (int)threadIdx.x; p < p_end; p += kBlock) {
    float x2, y2, iz, rf;
    if (!general_pixel<AR, DEPTH>(L, K, ga.sampler, I1, I2, DP, (uint32_t)p, x2, y2, iz, rf)) continue;
    const int q = (int)rintf(rf);   // saturate_cast<uchar> / lrint: round half to even; |q| <= 255
    atomicAdd(&h[q + 255], 1u);     // signed bins; integer atomics are order-independent
  }
  __syncthreads();
  unsigned int* gh = ga.hist + (size_t)pair * kHistBins;
  for (int i = threadIdx.x; i < kHistBins; i += kBlock)
    if (h[i]) atomicAdd(&gh[i], h[i]);
}

// Scale from the signed residual histogram alone (residuals are integers, or are binned by their rounded value):
//   median by the reference's rule (MedianMat, src/Tracker.cpp:1575-1591: first bin whose cumulative count exceeds
//   (float)(n / 2)); the reference Tukey saturates negatives to 0 first (:1572-1573), Huber keeps the sign;
//   the deviation histogram |q - med| follows from the same bins (clamped at 255 like the u8 conversion of
//   MedianAbsoluteDeviation's input, :1613, or at 510 for Huber); MAD = 1.4826 * its median (:1607-1619), 0 => 1 (:1634-1637).
// One wave per pair (four pairs per block).  Both medians are "the first bin whose cumulative count exceeds (float)(n / 2)"
// of a histogram that is a function of the pair's 511 signed bins, so each is one pass: every lane owns 8 consecutive bins,
// the lane totals are scanned across the wave, and the lane that holds the crossing reports the smallest index.
__device__ __forceinline__ int first_crossing(const unsigned int v[8], float m, int lane) {
  unsigned int tot = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) tot += v[k];
  unsigned int inc = tot;   // inclusive scan of the lane totals over the 64 lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int up = __shfl_up(inc, d, 64);
    if (lane >= d) inc += up;
  }
  unsigned int cum = inc - tot;
  int idx = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    cum += v[k];
    if (idx == 0x7fffffff && (float)cum > m) idx = lane * 8 + k;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) idx = min(idx, __shfl_xor(idx, d, 64));
  return idx;   // 0x7fffffff: no bin crosses
}

// one wave: the pair's scale from its 511 signed bins; lane l holds bins 8l .. 8l+7 in `mine`, `h` is a 512-word LDS
// scratch of this wave
__device__ __forceinline__ PairScale wave_scale(const unsigned int mine[8], unsigned int* __restrict__ h, bool tukey, int lane) {
  unsigned int n = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    h[lane * 8 + k] = mine[k];
    n += mine[k];
  }
  __builtin_amdgcn_wave_barrier();   // h is written and read by this wave alone (LDS operations of a wave stay in order)
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) n += __shfl_xor(n, d, 64);
  const float m = (float)(n / 2);
  // median of the residuals: bins q + 255; the reference Tukey saturates negatives first — every q <= 0 lands in bin 0
  // of its histogram (MedianMat, src/Tracker.cpp:1572-1591) — Huber keeps the sign
  int med;
  {
    unsigned int v[8];
    if (tukey) {
      unsigned int neg = 0;   // sum of bins 0..255
#pragma unroll
      for (int k = 0; k < 8; k++) neg += (lane * 8 + k <= 255) ? mine[k] : 0u;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) neg += __shfl_xor(neg, d, 64);
#pragma unroll
      for (int k = 0; k < 8; k++) {   // saturated histogram: value 0 = neg, value t (1..255) = bin t + 255
        const int t = lane * 8 + k;
        v[k] = t == 0 ? neg : (t <= 255 ? h[t + 255] : 0u);
      }
      const int idx = first_crossing(v, m, lane);
      med = idx == 0x7fffffff ? 255 : idx;
    } else {
      const int idx = first_crossing(mine, m, lane);
      med = idx == 0x7fffffff ? 255 : idx - 255;
    }
  }
  // median of |q - med| from the same bins: deviation d collects bins med - d and med + d; deviations >= dmax (the u8
  // saturation of MedianAbsoluteDeviation's input, :1613, or 510 for Huber) share the last value, which the cumulative count
  // always reaches
  const int dmax = tukey ? 255 : 510;
  int dmed;
  {
    unsigned int v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int d = lane * 8 + k;
      unsigned int c = 0;
      if (d < dmax) {
        const int lo = med - d, hi = med + d;
        if (lo >= -255 && lo <= 255) c += h[lo + 255];
        if (d > 0 && hi <= 255 && hi >= -255) c += h[hi + 255];
      }
      v[k] = c;
    }
    const int idx = first_crossing(v, m, lane);
    dmed = idx == 0x7fffffff ? dmax : idx;
  }
  PairScale sc;
  sc.med0 = (float)med;
  sc.n_valid = (int)n;
  sc.pad = 0;
  float mad = 1.4826f * (float)dmed;
  if (!n || mad == 0.f) mad = 1.f;
  sc.inv_mad = (float)(1.0 / (double)mad);
  return sc;
}

static __global__ __launch_bounds__(256) void k_scale_stage(const GeneralArgs ga, const PairState* state, int n_pairs, int pair_base) {
  __shared__ unsigned int sh[4][kHistBins];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n_pairs) return;
  const int pair = i + pair_base;
  const PairState st = state[pair];
  if (st.level_done || st.status) return;
  const unsigned int* gh = ga.hist + (size_t)pair * kHistBins;
  unsigned int mine[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int b = lane * 8 + k;
    mine[k] = b < 511 ? gh[b] : 0u;
  }
  const PairScale sc = wave_scale(mine, sh[wave], ga.weights == kWeightsTukeyRef, lane);
  if (lane == 0) ga.scale[pair] = sc;
}

// Vector form used by the alignment loop: same group walk as residual_block (VEC pixels per step), and kHistRep
// replicas of the LDS histogram interleaved by bin (word = bin * kHistRep + replica): residuals pile up around 0,
// same-address LDS atomics of one wave serialise, and the replicas of the crowded bins spread over the banks.
//
// The scale stage rides in the tail (round 3: no launch of its own, no per-evaluation clearing of the histograms): every
// block adds its counts to the pair's global bins, then takes a ticket from the pair's counter (word 511 of its
// histogram, which has 511 bins); the block that draws the last ticket reads-and-clears the bins (atomic exchange — the
// counts of the other blocks, made on other XCDs, are at the device's coherence point, not in this XCD's L2), derives the
// scale (wave_scale) and clears the counter.  The histograms are therefore all-zero between evaluations; the context
// clears them once per alignment call.
constexpr int kHistRep = 8;
constexpr int kHistTicketWord = kHistBins - 1;

// The scale pass over groups [g_begin, g_end) of one pair's level at `pose`: every valid pixel's rounded residual counted in the
// caller's LDS histogram (myh: bin 0 of this thread's replica).  Shared by k_resid_hist_v (one slice per block) and
// k_coarse_weighted (a whole level per block).  The masked ds_add_u32 are the asm's own: the caller waits (lgkmcnt(0)).
template <int AR, int VEC, bool DEPTH, int SAMPLER, bool RAGGED = false>
__device__ __forceinline__ void hist_groups(const LevelK& L, const WarpK& K, const uint8_t* __restrict__ I1, const uint8_t* __restrict__ I2,
                                            const uint16_t* __restrict__ DP, unsigned int* myh, const int g_begin, const int g_end,
                                            const int n_groups) {
  // the planes of a thread's next group are requested behind this group's gathers, as in residual_core
  uint8_t i1n[VEC];
  uint16_t dpn[VEC];
  auto load_planes = [&](int g) {
    const uint32_t idx = (uint32_t)min(g, n_groups - 1) * VEC;
    if constexpr (VEC == 4) {   // (plain loads: the weighted pass reads the same planes next, and finds them cached — streamed here: -1.2 %)
      *reinterpret_cast<uint32_t*>(i1n) = *reinterpret_cast<const uint32_t*>(I1 + idx);
      if constexpr (DEPTH) *reinterpret_cast<uint2*>(dpn) = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(DP) + idx * 2u);
    } else {
      i1n[0] = I1[idx];
      if constexpr (DEPTH) dpn[0] = DP[idx];
    }
  };
  load_planes(g_begin + (int)threadIdx.x);
  for (int g = g_begin + (int)threadIdx.x; g < g_end; g += kBlock) {
    const uint32_t idx = (uint32_t)g * VEC;
    const uint32_t y = __umulhi(idx, L.magic), x = idx - y * (uint32_t)L.pitch;
    uint8_t i1[VEC];
    uint16_t dp[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) { i1[j] = i1n[j]; if constexpr (DEPTH) dp[j] = dpn[j]; }
    // pairs of adjacent pixels through the packed float sequence (see v2f), as in the accumulation kernel
    constexpr int N = (VEC % 2 == 0) ? 2 : 1;
    using F = typename std::conditional<N == 2, v2f, float>::type;
    float x2[VEC], y2[VEC];
    unsigned long long okv[VEC];
    uint32_t gidx[VEC];
    unsigned long long colm[VEC];   // the grid points of the group (see residual_core): all of them where the level is whole
#pragma unroll
    for (int j = 0; j < VEC; j++) colm[j] = ~0ull;
    if constexpr (RAGGED) {
      const unsigned long long m_in = __builtin_amdgcn_uicmp(x, (uint32_t)L.gw, kIcmpULT);
      const unsigned long long m_all = __builtin_amdgcn_uicmp(x, (uint32_t)(L.gw & ~3), kIcmpULT);
      const int rem = L.gw & 3;
#pragma unroll
      for (int j = 0; j < VEC; j++) colm[j] = (VEC == 1 || j < rem) ? m_in : m_all;
    }
#pragma unroll
    for (int u = 0; u < VEC / N; u++) {
      F z = bc<F>(1.0f), xf, x2u, y2u, izu;
      unsigned long long okin_m[N], okm[N];
      float dlow[N];
#pragma unroll
      for (int c = 0; c < N; c++) {
        const int j = u * N + c;
        okin_m[c] = colm[j];
        dlow[c] = 1.0f;
        if constexpr (DEPTH) {
          const int d = (int)(int16_t)dp[j];
          dlow[c] = (float)d;
          put(z, c, dlow[c]);
        }
        put(xf, c, (float)x + (float)j);
      }
      if constexpr (DEPTH) z = z * bc<F>(L.zscale);
      // the reciprocal's clamp and its select are of no use here: only x2, y2 (the sample position) are read
      if constexpr (DEPTH) {   // depth > 0, y2 > 0, x2 > 0 as one compare of their minimum (see residual_core)
        F z2;
        warp_point<AR, F>(L, K, xf, bc<F>((float)y), z, x2u, y2u, z2, izu);
#pragma unroll
        for (int c = 0; c < N; c++) {
          const float uc = get(x2u, c), vc = get(y2u, c);
          float lo;
          asm("v_min3_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(uc), "v"(vc), "v"(dlow[c]));
          okm[c] = __builtin_amdgcn_fcmpf(lo, 0.f, kFcmpOGT) & __builtin_amdgcn_fcmpf(vc, (float)L.ih, kFcmpOLT) & __builtin_amdgcn_fcmpf(uc, (float)L.iw, kFcmpOLT);
          okm[c] &= okin_m[c];
        }
      } else {
        pixel_warp_raw<AR, F>(L, K, xf, bc<F>((float)y), z, okin_m, x2u, y2u, izu, okm);
      }
      if constexpr (SAMPLER == 0) {
        // nothing of an invalid pixel is sanitised: its sample index is clamped both ways (any in-range byte will do) and
        // its count is added under an EXEC mask of the valid lanes, like the sums of the accumulation kernel
#pragma unroll
        for (int c = 0; c < N; c++) {
          int ix2 = round_pos(get(x2u, c)), iy2 = round_pos(get(y2u, c));
          asm("v_med3_i32 %0, %0, 0, %1" : "+v"(ix2) : "s"(L.iw - 1));
          asm("v_med3_i32 %0, %0, 0, %1" : "+v"(iy2) : "s"(L.ih - 1));
          gidx[u * N + c] = __umul24((unsigned)iy2, (unsigned)L.pitch) + (unsigned)ix2;
        }
      }   // (bilinear: sample_bilinear clamps its four addresses both ways; nothing to prepare)
#pragma unroll
      for (int c = 0; c < N; c++) {
        x2[u * N + c] = get(x2u, c);
        y2[u * N + c] = get(y2u, c);
        okv[u * N + c] = okm[c];
      }
    }
    if constexpr (SAMPLER == 0) {
      int q[VEC];
#pragma unroll
      for (int j = 0; j < VEC; j++) q[j] = (int)I2[gidx[j]];
      __builtin_amdgcn_sched_barrier(0);
      load_planes(g + kBlock);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const int b = q[j] - (int)i1[j];   // in [-255, 255] for any two bytes: a bin of this thread's replica whatever the validity
        const unsigned addr = (unsigned)(uintptr_t)(myh + b * kHistRep);   // LDS byte address (the low 32 bits of the pointer)
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %1\n\t"
                     "ds_add_u32 %2, %3\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved) : "s"(okv[j]), "v"(addr), "v"(1u) : "scc", "memory");
      }
    } else {
      float rf[VEC];
#pragma unroll
      for (int j = 0; j < VEC; j++) rf[j] = sample_bilinear(I2, L, x2[j], y2[j]) - (float)i1[j];
      __builtin_amdgcn_sched_barrier(0);
      load_planes(g + kBlock);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        int b;   // |rf| <= 255 on a valid lane; an invalid lane's may be NaN (its position was): the instruction maps that to 0, the count is masked out
        asm("v_cvt_i32_f32 %0, %1" : "=v"(b) : "v"(rintf(rf[j])));
        const unsigned addr = (unsigned)(uintptr_t)(myh + b * kHistRep);
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %1\n\t"
                     "ds_add_u32 %2, %3\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved) : "s"(okv[j]), "v"(addr), "v"(1u) : "scc", "memory");
      }
    }
  }
}

// the scale pass of one block: slice blockIdx.x of `pair` at `pose` (the body of k_resid_hist_v and of k_hist_iterate)
template <int AR, int VEC, bool DEPTH, int SAMPLER, bool RAGGED>
__device__ __forceinline__ void hist_block(const ResidualArgs& a, const int pair, const Pose& pose, const int ref_slot, const int tgt_slot,
                                           unsigned int* __restrict__ hist, PairScale* __restrict__ scale_out, int weights) {
  __shared__ unsigned int h[kHistBins * kHistRep];
  __shared__ int s_last;
  for (int i = threadIdx.x; i < kHistRep * kHistBins; i += kBlock) h[i] = 0;
  __syncthreads();
  WarpK K;
  warp_setup<AR>(pose, K);
  const LevelK L = a.L;
  const size_t ref_off = (size_t)ref_slot * L.n, tgt_off = (size_t)tgt_slot * L.n;
  const uint8_t* __restrict__ I1 = a.img + ref_off;
  const uint8_t* __restrict__ I2 = a.img + tgt_off;
  const uint16_t* __restrict__ DP = DEPTH ? a.depth + ref_off : nullptr;
  // bin q of this thread's replica: myh[q * kHistRep]
  unsigned int* myh = h + 255 * kHistRep + (threadIdx.x & (kHistRep - 1));
  const int n_groups = L.ng / VEC;
  const int g_begin = blockIdx.x * a.groups_per_block, g_end = min(g_begin + a.groups_per_block, n_groups);
  hist_groups<AR, VEC, DEPTH, SAMPLER, RAGGED>(L, K, I1, I2, DP, myh, g_begin, g_end, n_groups);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the masked ds_add_u32 above are the asm's own: the compiler does not count them
  __syncthreads();
  unsigned int* gh = hist + (size_t)pair * kHistBins;
  // This block's counts must be performed (at the device's coherence point, where atomics execute) before its ticket is
  // drawn.  No fence: a release fence writes back the XCD's whole L2 (and 256 threads issuing one each took the scale
  // pass from 41 to 265 us per launch) — instead the adds RETURN their old values, which come from the coherence point, so
  // a thread that holds them has its adds performed; the barrier collects all threads.
  unsigned int seen = 0;
  for (int i = threadIdx.x; i < kHistBins - 1; i += kBlock) {
    const uint4 lo = *reinterpret_cast<const uint4*>(&h[i * kHistRep]), hi = *reinterpret_cast<const uint4*>(&h[i * kHistRep + 4]);
    const unsigned int t = lo.x + lo.y + lo.z + lo.w + hi.x + hi.y + hi.z + hi.w;
    if (t) seen |= atomicAdd(&gh[i], t);
  }
  if (seen == 0xffffffffu) s_last = 0;   // (never true: a use the compiler cannot drop, so the returns are waited for)
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&gh[kHistTicketWord], 1u) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!s_last || threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  unsigned int mine[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int b = lane * 8 + k;
    mine[k] = b < 511 ? atomicExch(&gh[b], 0u) : 0u;   // read at the coherence point, leave the bin cleared
  }
  if (lane == 0) atomicExch(&gh[kHistTicketWord], 0u);
  const PairScale sc = wave_scale(mine, h, weights == kWeightsTukeyRef, lane);   // (h: every wave of the block is past its flush)
  if (lane == 0) scale_out[pair] = sc;
}

template <int AR, int VEC, bool DEPTH, int SAMPLER, bool RAGGED = false>
__global__ __launch_bounds__(kBlock) void k_resid_hist_v(const ResidualArgs a, unsigned int* __restrict__ hist, PairScale* __restrict__ scale_out,
                                                         int weights) {
  const int pair = blockIdx.y + a.pair_base;
  const PairState st = a.state[pair];
  if (st.level_done || st.status) return;
  hist_block<AR, VEC, DEPTH, SAMPLER, RAGGED>(a, pair, st.pose, a.ref_slots[pair], a.tgt_slots[pair], hist, scale_out, weights);
}

// weighted / bilinear accumulation: J <- w·J, r <- gain·r, A = Σ(wJ)(wJ)ᵀ, jtr = Σ(wJ)·((gain r)·w) (src/Tracker.cpp:554-561),
// error numerator Σ r·(r·w) (:499-502).  With identity weights this is the plain sum with float residuals.
template <int AR, bool DEPTH, bool UNIT_FACTORS>
__global__ __launch_bounds__(kBlock) void k_residual_general(const ResidualArgs a, const GeneralArgs ga) {
  const int pair = blockIdx.y + a.pair_base;
  Pose pose;
  if (a.state) {
    const PairState st = a.state[pair];
    if (st.level_done || st.status) return;
    pose = st.pose;
  } else {
    pose = a.pose;
  }
  WarpK K;
  warp_setup<AR>(pose, K);
  const LevelK L = a.L;
  const size_t ref_off = (size_t)a.ref_slots[pair] * L.n, tgt_off = (size_t)a.tgt_slots[pair] * L.n;
  const uint8_t* I1 = a.img + ref_off;
  const uint8_t* I2 = a.img + tgt_off;
  const int16_t* GX = a.gx + ref_off;
  const int16_t* GY = a.gy + ref_off;
  const uint16_t* DP = DEPTH ? a.depth + ref_off : nullptr;
  const float inv_mad = ga.weights ? ga.scale[pair].inv_mad : 1.f;
  double acc[kAccFloats];
#pragma unroll
  for (int i = 0; i < kAccFloats; i++) acc[i] = 0.0;
  double err = 0.0;
  uint32_t sum_r2 = 0, n_valid = 0;
  const int p_begin = blockIdx.x * a.groups_per_block, p_end = min(p_begin + a.groups_per_block, L.ng);
  for (int p = p_begin + (int)threadIdx.x; p < p_end; p += kBlock) {
    float x2, y2, iz, rf;
    const bool ok = general_pixel<AR, DEPTH>(L, K, ga.sampler, I1, I2, DP, (uint32_t)p, x2, y2, iz, rf);
    float J[6], w = 1.f;
    if (ok) {
      pixel_jacobian<AR, UNIT_FACTORS, false, true>(L, a.zf, a.af, x2, y2, iz, (float)GX[p], (float)GY[p], J);
      w = robust_weight(ga.weights, rf, inv_mad);
      const float rw1 = rf * w;                 // Residuals.mul(W) for the error (:500)
      err += (double)rf * (double)rw1;
      float Jw[6];
#pragma unroll
      for (int k = 0; k < 6; k++) Jw[k] = w * J[k];   // :556
      const float rg = rf * ga.gain;            // :559
      const float rw = rg * w;                  // :561
      double Jd[6];
#pragma unroll
      for (int k = 0; k < 6; k++) Jd[k] = (double)Jw[k];
      int s = 0;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++, s++) acc[s] = __builtin_fma(Jd[i], Jd[j], acc[s]);
#pragma unroll
      for (int i = 0; i < 6; i++) acc[21 + i] = __builtin_fma(Jd[i], (double)rw, acc[21 + i]);
      const int q = (int)rintf(rf);
      sum_r2 += (uint32_t)(q * q);
      n_valid += 1;
    }
    if (a.dumpV) a.dumpV[(size_t)pair * L.ng + p] = ok ? 1 : 0;
    if (a.dumpR) a.dumpR[(size_t)pair * L.ng + p] = ok ? rf : 0.f;
    if (a.dumpJ)
      for (int k = 0; k < 6; k++) a.dumpJ[((size_t)pair * L.ng + p) * 6 + k] = ok ? J[k] : 0.f;
    if (a.dumpW) a.dumpW[(size_t)pair * L.ng + p] = ok ? w : 0.f;
  }
  block_reduce_store<double, true>(acc, sum_r2, n_valid, a.partials + ((size_t)pair * a.slices + blockIdx.x) * kRecWords, err);
}

// ------------------------------------------------------------------------------------------------------------
// The scalar tail of one Gauss-Newton iteration for one pair: fold the block partials of the evaluation in slice order
// (f64), error (src/Tracker.cpp:499-502), exit test (:508), A/b (:554-561), A.inv()*b (:564), pose <- pose * exp(delta)
// (:574).  update_compute is executed by a whole 256-thread block and hands every thread the pair's new state; it is
// the body of k_gn_update (one block per pair) and the first phase of k_iterate (every block of the pair's next
// evaluation recomputes it — a few microseconds of mostly serial work — instead of waiting for a launch of its own).
// ------------------------------------------------------------------------------------------------------------
struct UpdateArgs {
  const uint32_t* partials;
  PairState* state;
  int slices;
  int k;             // iteration index at this level
  int max_iters;
  int early_exit;
  float epsilon;
  float gain;
  int pair_base;
  int general;       // 1: records come from k_residual_general (gain already applied, error numerator in slot 29)
  int* active;       // optional: counts the pairs still iterating after this update (early-exit polling)
  int legacy_solve;  // uwt_params::arith == UWT_ARITH_LEGACY: A.inv() formed, then multiplied (else cv::solve's LU on b)
};

constexpr int kUpdateBlock = 256;   // threads of an updating block: all fold the records, wave 0 solves
constexpr int kFoldBatch = 16;      // record loads a thread keeps in flight (8 x 16 = 128 records per round trip; 20 — every fold of up to kMaxSlices records one
                                    // round trip — was measured in round 6: no gain at 150 records, the reference schedule 4 % slower: profiles/r06/EXPERIMENTS.md)
constexpr int kUpdateLdsBytes = 2 * 8 * 32 * 8 + 512;   // part sums (two readings) + sums, integer sums, the state to broadcast

// One wave, uniform values: from the evaluation's folded sums (sums[0..26]: JtJ upper triangle and Jtr, sums[27]: the
// weighted error numerator; isums: valid pixels, integer sum of r^2) to the pair's next state.
__device__ __forceinline__ void update_solve_wave(const UpdateArgs& a, PairState& st, const double* __restrict__ sums,
                                                  const long long* __restrict__ isums, bool count_active, int lane) {
  const int n = (int)isums[0];
  const long long sr2 = isums[1];
  st.iters += 1;
  st.n_valid = n;
  bool update = true;
  if (n == 0) {
    st.status = 2;  // UWT_ERR_NO_VALID_POINTS
    st.level_done = 1;
    update = false;
  } else {
    const float inv_n = (float)(1.0 / (double)n);                 // src/Tracker.cpp:499
    const float error = a.general ? (float)((double)inv_n * sums[kAccFloats])   // Σ r·(r·w): float / weighted residuals
                                  : (float)((double)inv_n * (double)sr2);       // :501, scaled-gemm form (exact integer Σr²)
    st.error = error;
    if (a.early_exit &&
        (error >= st.last_error || a.k == a.max_iters - 1 || fabsf(error - st.last_error) < a.epsilon)) {  // :508
      st.level_done = 1;
      update = false;
    } else {
      st.last_error = error;  // :529
    }
  }
  if (update) {
    float b[6], delta[6];
#pragma unroll
    for (int i = 0; i < 6; i++)
      b[i] = a.general ? (float)(-sums[21 + i]) : (float)(-((double)a.gain * sums[21 + i]));  // :559-561
    solve_delta_wave(sums, b, delta, a.legacy_solve != 0);                                             // :554-564, A = (float)sums[0..20]
    Pose d, np;
    se3_exp_wave(delta, d);                                                         // :574
    se3_mul(st.pose, d, np);
    st.pose = np;
    if (count_active && a.active && lane == 0) atomicAdd(a.active, 1);
  }
}

// The update in the tail of the evaluation's own launch (round 3): no k_gn_update launch behind every residual launch.
// Wave 0 of a block writes the block's record (block_reduce_store_at, `coherent`), then draws a ticket from the pair's
// counter; the wave that draws the last one (slices - 1) knows every record of the pair is at the coherence point —
// each was there before its block's ticket was drawn — and folds them (agent-scope loads: past this XCD's L2), solves
// and writes the pair's next state, which the next launch reads.  The fold adds in update_compute's order (parts 0..7 of
// records part, part + 8, ..., then the part sums in part order): the same sums bit for bit, by one wave — lane (slot,
// half) takes parts half, half + 2, half + 4, half + 6 of its slot.  The counter is zero again afterwards.
// Ordering without fences: see block_reduce_store_at; the ticket is drawn behind the returned exchanges in program order,
// the loads are issued behind the returned ticket.
constexpr int kTailRounds = 4;   // rounds of 8 records a lane has in flight per part (4 x 4 loads)
__device__ __forceinline__ void tail_update_wave(const ResidualArgs& a, int pair) {
  __shared__ __attribute__((aligned(16))) double t_sums[kAccFloats + 1];
  __shared__ long long t_isums[2];
  const int lane = (int)threadIdx.x;   // wave 0
  unsigned int ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(a.tail.tickets + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
  if (ticket != (unsigned int)a.slices - 1u) return;
  if (lane == 0) __hip_atomic_store(a.tail.tickets + pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int slot = lane & 31, half = lane >> 5;
  const unsigned long long* g8 = reinterpret_cast<const unsigned long long*>(a.partials + (size_t)pair * a.slices * kRecWords) + slot;
  double cs[4] = {0.0, 0.0, 0.0, 0.0};
  long long is[4] = {0, 0, 0, 0};
  for (int q0 = 0; q0 < a.slices; q0 += 8 * kTailRounds) {
    unsigned long long v[kTailRounds][4];
#pragma unroll
    for (int r = 0; r < kTailRounds; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = q0 + 8 * r + half + 2 * j;
        v[r][j] = q < a.slices ? __hip_atomic_load(g8 + (size_t)q * (kRecWords / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      }
#pragma unroll
    for (int r = 0; r < kTailRounds; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        cs[j] += __longlong_as_double((long long)v[r][j]);
        is[j] += (long long)(slot == 27 ? (v[r][j] & 0xffffffffull) : v[r][j]);   // slot 27: n_valid in the low word
      }
  }
  double fs = 0.0;
  long long ls = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {   // parts 2j (this half's when half == 0) and 2j + 1 (the other half's), in part order
    const double of = __shfl_xor(cs[j], 32);
    const long long oi = __shfl_xor(is[j], 32);
    if (j == 0) { fs = cs[0]; ls = is[0]; } else { fs += cs[j]; ls += is[j]; }
    fs += of;
    ls += oi;
  }
  if (lane < 32) {
    if (lane < kAccFloats) t_sums[lane] = fs;
    else if (lane == 27) t_isums[0] = (long long)(uint32_t)ls;
    else if (lane == 28) t_isums[1] = ls;
    else if (lane == 29) t_sums[kAccFloats] = a.tail.general ? fs : 0.0;
  }
  __builtin_amdgcn_wave_barrier();
  UpdateArgs u;
  u.k = a.tail.k;
  u.max_iters = a.tail.max_iters;
  u.early_exit = a.tail.early_exit;
  u.epsilon = a.tail.epsilon;
  u.gain = a.tail.gain;
  u.general = a.tail.general;
  u.legacy_solve = a.tail.legacy_solve;
  u.active = a.tail.active;
  PairState st = a.tail.state[pair];   // as every block of the pair read it at the start: nobody has written it since
  update_solve_wave(u, st, t_sums, t_isums, true, lane);
  if (lane == 0) a.tail.state[pair] = st;
}

__device__ __forceinline__ PairState update_compute(const UpdateArgs& a, const uint32_t* __restrict__ recs, const PairState* __restrict__ st_in,
                                                    unsigned char* __restrict__ lds, bool count_active) {
  const int tid = (int)thread_here(), lane = tid & 63;   // (opaque: see thread_here)
  double(*part_f)[32] = reinterpret_cast<double(*)[32]>(lds);                              // [8][32] part sums, f64 reading
  long long(*part_i)[32] = reinterpret_cast<long long(*)[32]>(lds + 8 * 32 * 8);            // [8][32] part sums, integer reading
  double* sums = reinterpret_cast<double*>(lds + 2 * 8 * 32 * 8);                           // [kAccFloats + 1]
  long long* isums = reinterpret_cast<long long*>(sums + kAccFloats + 1);                 // [2]
  PairState* s_state = reinterpret_cast<PairState*>(isums + 2);
  // Fold of the evaluation's records, all threads at once: a record is 32 eight-byte slots; thread (slot = tid & 31,
  // part = tid >> 5) pulls slot `slot` of records part, part + 8, part + 16, ... straight from memory — independent
  // loads, one round trip per kFoldBatch of them — and adds them in that order; the eight part sums of a slot are then
  // added in part order by one lane.  The order of the additions is fixed by (slices) alone.  The first batch is requested
  // before the pair's state is looked at, so that both arrive in one round trip.
  const int slot = tid & 31, part = tid >> 5;
  const unsigned long long* g8 = reinterpret_cast<const unsigned long long*>(recs) + slot;
  unsigned long long v[kFoldBatch];
#pragma unroll
  for (int u = 0; u < kFoldBatch; u++) {
    const int q = part + 8 * u;
    v[u] = q < a.slices ? g8[(size_t)q * (kRecWords / 2)] : 0ull;   // +0.0 / 0: neutral in both readings
  }
  PairState st = *st_in;
  const bool live = !(st.level_done || st.status);  // block-uniform
  if (live) {
    double cs = 0.0;
    long long is = 0;
    for (int q0 = part;;) {
#pragma unroll
      for (int u = 0; u < kFoldBatch; u++) {
        cs += __longlong_as_double((long long)v[u]);
        is += (long long)(slot == 27 ? (v[u] & 0xffffffffull) : v[u]);   // slot 27: n_valid in the low word
      }
      q0 += 8 * kFoldBatch;
      if (q0 >= a.slices) break;
#pragma unroll
      for (int u = 0; u < kFoldBatch; u++) {
        const int q = q0 + 8 * u;
        v[u] = q < a.slices ? g8[(size_t)q * (kRecWords / 2)] : 0ull;
      }
    }
    part_f[part][slot] = cs;
    part_i[part][slot] = is;
    __syncthreads();
    if (tid < 32) {
      double fs = part_f[0][tid];
      long long ls = part_i[0][tid];
#pragma unroll
      for (int q = 1; q < 8; q++) {
        fs += part_f[q][tid];
        ls += part_i[q][tid];
      }
      if (tid < kAccFloats) sums[tid] = fs;
      else if (tid == 27) isums[0] = (long long)(uint32_t)ls;
      else if (tid == 28) isums[1] = ls;
      else if (tid == 29) sums[kAccFloats] = a.general ? fs : 0.0;
    }
  }
  __builtin_amdgcn_wave_barrier();   // the sums were written by lanes of wave 0, which alone reads them
  if (tid < 64) {   // wave 0 runs the tail together on the same (uniform) values
    if (live) update_solve_wave(a, st, sums, isums, count_active, lane);
    if (lane == 0) *s_state = st;
  }
  __syncthreads();
  return *s_state;
}

static __global__ __launch_bounds__(kUpdateBlock) void k_gn_update(const UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kUpdateLdsBytes];
  const int pair = (int)blockIdx.x + a.pair_base;
  const PairState st = update_compute(a, a.partials + (size_t)pair * a.slices * kRecWords, &a.state[pair], lds, true);
  if (threadIdx.x == 0) a.state[pair] = st;
}

// ------------------------------------------------------------------------------------------------------------
// k_iterate: one launch = one Gauss-Newton iteration of one pyramid level for a whole batch, on the dense
// nearest-neighbour / identity-weights path.  Every block first brings its pair's state up to date — the update that
// belongs to the PREVIOUS evaluation (records of the other parity), and the level hand-off when this launch opens a new
// level — then evaluates its slice at the new pose.  Compared with k_residual + k_gn_update per iteration this halves
// the launches of an alignment (a pair on its own spends most of its time in kernel boundaries and dependent memory
// round trips, not in arithmetic) and takes the update's launch out of the batch's critical path.  States and records
// are double-buffered: blocks of one launch read the previous launch's and write this launch's.
// ------------------------------------------------------------------------------------------------------------
struct IterArgs {
  UpdateArgs u;               // the pending update: u.partials / u.slices / u.k describe the previous evaluation
  const PairState* state_in;
  PairState* state_out;
  int mode;                   // 0: first evaluation of the alignment (fresh state); 1: update, evaluate; 2: update, hand-off, evaluate;
                              // 3: evaluate from state_in as it is (the launch behind k_coarse)
  int prev_lvl;               // level the pending update / hand-off belongs to (mode 2)
  int scale_t;
  float initial_error;
  int* cut_short;             // optional (speculative launching, early-exit schedules): set when a level's launches ran out
                              // before the pair's exit test fired, i.e. the host stopped launching too early (may be host memory)
  int inline_pairs;           // 1: the (at most two) pairs' slots travel in the kernel arguments, no pair list in memory
  int pair_slots[4];          // ref 0, tgt 0, ref 1, tgt 1
};

__device__ __forceinline__ PairState iterate_state(const IterArgs& ia, int pair, unsigned char* lds, bool count_active) {
  PairState st;
  if (ia.mode == 0) {
    pose_identity(st.pose);  // src/Tracker.cpp:385
    st.last_error = ia.initial_error;
    st.error = 0.f;
    st.level_done = 0;
    st.status = 0;
    st.iters = 0;
    st.n_valid = 0;
    return st;
  }
  if (ia.mode == 3) return ia.state_in[pair];   // behind k_coarse: the state is current, nothing is pending
  st = update_compute(ia.u, ia.u.partials + (size_t)pair * ia.u.slices * kRecWords, &ia.state_in[pair], lds, count_active);
  if (ia.mode == 2) {   // end of a pyramid level: hand-off (src/Tracker.cpp:580-590) and re-arm for the next level (:392-393)
    // with early exit a level ends only through its exit test (which fires at the last iteration at the latest)
    if (ia.cut_short && ia.u.early_exit && !st.level_done && st.status == 0 && count_active && threadIdx.x == 0)
      __hip_atomic_store(ia.cut_short, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (st.status == 0 && ia.prev_lvl != 0) {
      if (!se3_handoff(st.pose, ia.scale_t != 0)) st.status = 1;  // SOPHUS_ENSURE would abort
    }
    st.level_done = 0;
    st.last_error = ia.initial_error;
  }
  return st;
}

constexpr int iterate_lds_bytes(int pass) { return kUpdateLdsBytes > reduce_lds_bytes(pass) ? kUpdateLdsBytes : reduce_lds_bytes(pass); }

// PASS: rows per LDS pass of the block reduction — kIteratePass (one pass, 60 KB) while the blocks have their CUs to
// themselves (up to 3 pairs: 0.40 against 0.42 ms for one), 14 (two passes, 34 KB, four blocks per CU) from 4 pairs on
// (6 pairs: 0.49 against 0.52 ms).
// PLAIN (the flow kernels' one shape switch): true = the reference's constants — square pixels (fx == fy bitwise) and unit
// z / angle factors —, i.e. residual_core<UNIT_FACTORS, SQUARE>; false = the general form (fx != fy and / or other factors:
// every product written out, the factors multiplied in), which computes the same bits where both apply (x * 1.0f == x).
// f64 sums only (a context with accumulate_f64 = 0 stays on the per-evaluation launches).
template <int AR, int VEC, bool DEPTH, bool PLAIN, bool COMPUTE_ONLY = false, int PASS = kIteratePass, bool RAGGED = false>
__global__ __launch_bounds__(kBlock) void k_iterate(const ResidualArgs a, const IterArgs ia) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[iterate_lds_bytes(PASS)];   // the update's staging, then the reduction's image
  const int pair = (int)blockIdx.y + a.pair_base, slice = (int)blockIdx.x;
  // the first group's reference planes do not depend on the pose: their (cold) loads travel while the update runs
  RefGroup<VEC> first;
  const int lp = (int)blockIdx.y;
  // (selects, not an indexed read: indexing the by-value argument would send it through scratch memory)
  const int ref_slot = ia.inline_pairs ? (lp == 0 ? ia.pair_slots[0] : ia.pair_slots[2]) : a.ref_slots[pair];
  const int tgt_slot = ia.inline_pairs ? (lp == 0 ? ia.pair_slots[1] : ia.pair_slots[3]) : a.tgt_slots[pair];
  load_first_group<VEC, DEPTH, COMPUTE_ONLY>(first, a, ref_slot, slice);
  PairState st = iterate_state(ia, pair, lds, slice == 0);
  if (slice == 0 && threadIdx.x == 0) ia.state_out[pair] = st;
  if constexpr (COMPUTE_ONLY) {
    st.level_done = 0;
    st.status = 0;
    pose_identity(st.pose);   // the twin does the full work of every launch, whatever its meaningless sums produced
  }
  if (st.level_done || st.status) return;
  __syncthreads();   // the staging bytes become the reduction's
  residual_core<AR, VEC, DEPTH, PLAIN, false, double, PLAIN, 0, 0, COMPUTE_ONLY, PASS, 0, RAGGED>(a, pair, slice, st.pose, lds, &first, ref_slot, tgt_slot);
}

// k_hist_iterate (round 6): what k_iterate is to the identity path, for robust weights — a few pairs per call.  An evaluation under
// robust weights is two launches (the scale needs every residual before any weight exists): the scale pass (k_resid_hist_v) and
// the weighted sums (k_residual<.., WEIGHTS>).  The update that used to follow as a third launch (k_gn_update, and k_level_end where a
// level ends) is folded into the head of the NEXT evaluation's scale pass: every block first brings its pair's state up to date from
// the previous evaluation's records (iterate_state: the same fold, solve and hand-off, bit for bit), slice 0 publishes it for the
// weighted launch behind, then the block bins its slice at the new pose.  States and records are double-buffered as in k_iterate.
template <int AR, bool DEPTH, int SAMPLER, bool RAGGED = false>
__global__ __launch_bounds__(kBlock) void k_hist_iterate(const ResidualArgs a, const IterArgs ia, unsigned int* __restrict__ hist,
                                                         PairScale* __restrict__ scale_out, int weights) {
  __shared__ __attribute__((aligned(16))) unsigned char ulds[kUpdateLdsBytes];
  const int pair = (int)blockIdx.y + a.pair_base, slice = (int)blockIdx.x;
  const int lp = (int)blockIdx.y;
  const int ref_slot = ia.inline_pairs ? (lp == 0 ? ia.pair_slots[0] : ia.pair_slots[2]) : a.ref_slots[pair];
  const int tgt_slot = ia.inline_pairs ? (lp == 0 ? ia.pair_slots[1] : ia.pair_slots[3]) : a.tgt_slots[pair];
  const PairState st = iterate_state(ia, pair, ulds, slice == 0);
  if (slice == 0 && threadIdx.x == 0) ia.state_out[pair] = st;
  if (st.level_done || st.status) return;   // (every block of the pair alike: nobody draws a ticket)
  hist_block<AR, 4, DEPTH, SAMPLER, RAGGED>(a, pair, st.pose, ref_slot, tgt_slot, hist, scale_out, weights);
}

// ------------------------------------------------------------------------------------------------------------
// k_coarse: the coarsest levels of a lone pair's alignment in ONE launch.  A level of a few thousand pixels fits a single
// block; evaluating it there — the record stays in LDS, the update reads it from there, the exit test is taken on the
// device — removes the kernel boundary, the cross-CU round trip of the records and the re-launch per iteration (and, in
// early-exit schedules, every speculative launch of these levels): about 5 us per evaluation instead of 10.  One block
// per pair runs levels lv[0..n_levels) to their end (hand-offs included) and leaves the state for the first k_iterate
// launch of the next level (mode 3).  Same device functions as k_iterate (residual_core with one slice, update_compute on
// one record); the f64 sums are grouped differently, as between any two slicings.
// ------------------------------------------------------------------------------------------------------------
constexpr int kCoarseMaxLevels = 3;
constexpr int kCoarseMaxPixels = 6144;   // a level up to here is evaluated by one block (24 pixels per thread)
struct CoarseArgs {
  ResidualArgs lv[kCoarseMaxLevels];   // coarsest first; partials / slices / groups_per_block are set by the kernel
  int level_id[kCoarseMaxLevels];      // pyramid level of lv[i] (the hand-off is skipped behind level 0)
  int n_levels;
  UpdateArgs u;                        // max_iters, early_exit, epsilon, gain
  PairState* state_out;
  int scale_t;
  float initial_error;
  int inline_pairs;                    // see IterArgs
  int pair_slots[4];
  int resume;                          // 1: continue from state_out[pair] instead of starting the alignment
};

// PASS: rows per LDS pass of the block reduction: kIteratePass (61 KB, the block alone on its CU: a lone pair) or 14 (35 KB,
// four blocks per CU: the batch form, one block per pair of a whole batch — round 3)
// NLEV: levels the launch can run (the loop over them is unrolled).  The batch form (one block per pair of a whole batch, one
// level per launch, PASS 14) stays at ~210 registers, two waves per SIMD: forced to 128 it spills and loses (measured), so it
// pays only on the smallest levels, where the per-evaluation launches run furthest below the level-0 rate.
// VECSEL: the grid rows of the levels the launch runs: 4 (every level's are whole groups of four), 1 (none's are: a lone level
// like 46 x 30 or 47 x 30, every level of an odd-sized frame: the RAGGED instantiation of residual_core), 0 (decided per level at
// run time: a chain like 46 -> 92 wide)
template <int AR, bool DEPTH, bool PLAIN, int PASS, int NLEV, int VECSEL>
__device__ __forceinline__ void coarse_body(const CoarseArgs& ca);

template <int AR, bool DEPTH, bool PLAIN, int PASS = kIteratePass, int NLEV = kCoarseMaxLevels, int VECSEL = 4>
__global__ __launch_bounds__(kBlock) void k_coarse(const CoarseArgs ca) {
  coarse_body<AR, DEPTH, PLAIN, PASS, NLEV, VECSEL>(ca);
}
// The batch form (one block per pair of a whole batch, PASS 14, one level per launch) held to four waves per SIMD: all blocks
// of a 1024-pair batch are resident at once, and while one block's wave 0 runs its update the other three blocks of the CU
// evaluate.  Two things made that possible (round 4; before, the kernel took ~210 registers and spilled inside the loop when
// capped): the library is built without machine-level loop-invariant code motion (the f64 sine / cosine polynomials of the
// update had their ~40 constant registers hoisted above the iteration loop, live through the residual loop), and the thread
// index behind the update's and the reduction's lane roles is opaque (thread_here).  What remains above 128 is parked in
// scratch outside the residual loop (40 bytes per lane).
template <int AR, bool DEPTH, bool PLAIN, int VECSEL = 4, int PASS = 14, int NLEV = 1>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_coarse_w4(const CoarseArgs ca) {
  coarse_body<AR, DEPTH, PLAIN, PASS, NLEV, VECSEL>(ca);
}
template <int AR, bool DEPTH, bool PLAIN, int PASS, int NLEV, int VECSEL>
__device__ __forceinline__ void coarse_body(const CoarseArgs& ca) {
  constexpr int kLds = iterate_lds_bytes(PASS);
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLds + kRecWords * 4 + 64];
  uint32_t* rec = reinterpret_cast<uint32_t*>(lds + kLds);                     // the evaluation's record
  PairState* cur = reinterpret_cast<PairState*>(lds + kLds + kRecWords * 4);   // the state update_compute reads
  const int lp = (int)blockIdx.x, pair = lp + ca.u.pair_base;
  const int ref_slot = ca.inline_pairs ? (lp == 0 ? ca.pair_slots[0] : ca.pair_slots[2]) : ca.lv[0].ref_slots[pair];
  const int tgt_slot = ca.inline_pairs ? (lp == 0 ? ca.pair_slots[1] : ca.pair_slots[3]) : ca.lv[0].tgt_slots[pair];
  PairState st;
  if (ca.resume) {
    st = ca.state_out[pair];   // a level of a batch behind the coarser one's launch: re-armed by that launch's tail
  } else {
    pose_identity(st.pose);  // src/Tracker.cpp:385
    st.last_error = ca.initial_error;
    st.error = 0.f;
    st.level_done = 0;
    st.status = 0;
    st.iters = 0;
    st.n_valid = 0;
  }
#pragma unroll
  for (int li = 0; li < NLEV; li++) {
    if (li >= ca.n_levels) break;   // block-uniform
    const ResidualArgs& a = ca.lv[li];   // read in place (the kernel-argument segment); what differs travels in `ov`
    const bool v4 = VECSEL == 4 || (VECSEL == 0 && a.L.gw == a.L.pitch);   // block-uniform
    CoreOverride ov;
    ov.groups_per_block = ((a.L.ng / 4 + kBlock - 1) / kBlock) * kBlock;   // the whole level
    ov.rec = rec;
    UpdateArgs u = ca.u;
    u.slices = 1;
    u.active = nullptr;
    if (st.status == 0) {
      for (int k = 0; k < u.max_iters; k++) {
        __syncthreads();   // every thread has taken the state out of the update's LDS bytes: they become the reduction's
        if (threadIdx.x == 0) *cur = st;
        if constexpr (VECSEL == 4) {
          residual_core<AR, 4, DEPTH, PLAIN, false, double, PLAIN, 0, 0, false, PASS>(a, pair, 0, st.pose, lds, nullptr, ref_slot, tgt_slot, &ov);
        } else if constexpr (VECSEL == 1) {
          residual_core<AR, 4, DEPTH, PLAIN, false, double, PLAIN, 0, 0, false, PASS, 0, true>(a, pair, 0, st.pose, lds, nullptr, ref_slot, tgt_slot, &ov);
        } else {
          if (v4) residual_core<AR, 4, DEPTH, PLAIN, false, double, PLAIN, 0, 0, false, PASS>(a, pair, 0, st.pose, lds, nullptr, ref_slot, tgt_slot, &ov);
          else residual_core<AR, 4, DEPTH, PLAIN, false, double, PLAIN, 0, 0, false, PASS, 0, true>(a, pair, 0, st.pose, lds, nullptr, ref_slot, tgt_slot, &ov);
        }
        __syncthreads();   // the record and the state are in LDS; the reduction's image is free
        u.k = k;
        st = update_compute(u, rec, cur, lds, false);   // ends with a barrier: every thread has the new state
        if (st.level_done || st.status) break;          // block-uniform (the exit test, src/Tracker.cpp:508)
      }
    }
    // end of a pyramid level: hand-off (src/Tracker.cpp:580-590) and re-arm for the next level (:392-393)
    if (st.status == 0 && ca.level_id[li] != 0) {
      if (!se3_handoff(st.pose, ca.scale_t != 0)) st.status = 1;  // SOPHUS_ENSURE would abort
    }
    st.level_done = 0;
    st.last_error = ca.initial_error;
  }
  if (threadIdx.x == 0) ca.state_out[pair] = st;
}

// ------------------------------------------------------------------------------------------------------------
// k_coarse_weighted: what k_coarse is to the identity path, for robust weights over the nearest-neighbour sampler — a level of up
// to kCoarseMaxPixels pixels of one pair evaluated by ONE block, every iteration of it in one launch: the residual histogram in
// LDS (hist_groups), median and MAD from it (wave_scale), the weighted sums through the per-value weight table (residual_core),
// the update on the record in LDS (update_compute), the exit test on the device.  Per evaluation that replaces the scale
// launch, the weighted launch and the update (three dependent launches of a few blocks each, ~20 us for a lone pair) by ~8 us
// of one resident block.  Same device functions as the launches it replaces: same bits.
// ------------------------------------------------------------------------------------------------------------
template <int AR, bool DEPTH, int WEIGHTS, bool PLAIN, int NLEV = kCoarseMaxLevels, bool RAGGED = false>
__global__ __launch_bounds__(kBlock) void k_coarse_weighted(const CoarseArgs ca) {
  constexpr int VEC = 4;
  __shared__ unsigned int h[kHistBins * kHistRep];                                  // the residual histogram, kHistRep replicas per bin
  __shared__ unsigned int h_scratch[kHistBins];                                     // wave_scale's working copy
  __shared__ __attribute__((aligned(16))) unsigned char ulds[kUpdateLdsBytes];      // update_compute's staging
  __shared__ __attribute__((aligned(16))) uint32_t rec[kRecWords];                  // the evaluation's record
  __shared__ PairState cur;                                                         // the state update_compute reads
  __shared__ PairScale s_scale;
  const int lp = (int)blockIdx.x, pair = lp + ca.u.pair_base;
  const int ref_slot = ca.inline_pairs ? (lp == 0 ? ca.pair_slots[0] : ca.pair_slots[2]) : ca.lv[0].ref_slots[pair];
  const int tgt_slot = ca.inline_pairs ? (lp == 0 ? ca.pair_slots[1] : ca.pair_slots[3]) : ca.lv[0].tgt_slots[pair];
  PairState st;
  if (ca.resume) {
    st = ca.state_out[pair];
  } else {
    pose_identity(st.pose);  // src/Tracker.cpp:385
    st.last_error = ca.initial_error;
    st.error = 0.f;
    st.level_done = 0;
    st.status = 0;
    st.iters = 0;
    st.n_valid = 0;
  }
#pragma unroll
  for (int li = 0; li < NLEV; li++) {
    if (li >= ca.n_levels) break;   // block-uniform
    const ResidualArgs& a = ca.lv[li];
    const LevelK L = a.L;
    const int n_groups = L.ng / VEC;
    CoreOverride ov;
    ov.groups_per_block = ((n_groups + kBlock - 1) / kBlock) * kBlock;   // the whole level
    ov.rec = rec;
    ov.scale = &s_scale;
    UpdateArgs u = ca.u;
    u.slices = 1;
    u.active = nullptr;
    u.general = 1;
    const size_t ref_off = (size_t)ref_slot * L.n, tgt_off = (size_t)tgt_slot * L.n;
    if (st.status == 0) {
      for (int k = 0; k < u.max_iters; k++) {
        // the scale pass (MedianMat / MedianAbsoluteDeviation, src/Tracker.cpp:1571-1619) at this evaluation's pose
        for (int i = threadIdx.x; i < kHistRep * kHistBins; i += kBlock) h[i] = 0;
        __syncthreads();   // (also: every thread has taken the last state out of the update's bytes)
        if (threadIdx.x == 0) cur = st;
        {
          WarpK K;
          warp_setup<AR>(st.pose, K);
          hist_groups<AR, VEC, DEPTH, 0, RAGGED>(L, K, a.img + ref_off, a.img + tgt_off, DEPTH ? a.depth + ref_off : nullptr,
                                       h + 255 * kHistRep + (threadIdx.x & (kHistRep - 1)), 0, n_groups, n_groups);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // hist_groups' masked ds_add_u32 are its asm's own
        __syncthreads();
        if (threadIdx.x < 64) {
          const int lane = (int)threadIdx.x;
          unsigned int mine[8];
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const int b = lane * 8 + q;
            unsigned int t = 0;
            if (b < 511) {
              const uint4 lo = *reinterpret_cast<const uint4*>(&h[b * kHistRep]), hi = *reinterpret_cast<const uint4*>(&h[b * kHistRep + 4]);
              t = lo.x + lo.y + lo.z + lo.w + hi.x + hi.y + hi.z + hi.w;
            }
            mine[q] = t;
          }
          const PairScale sc = wave_scale(mine, h_scratch, WEIGHTS == kWeightsTukeyRef, lane);
          if (lane == 0) s_scale = sc;
        }
        __syncthreads();
        // the weighted sums (src/Tracker.cpp:554-561) through the weight table; the record lands in LDS
        residual_core<AR, VEC, DEPTH, PLAIN, false, double, PLAIN, 0, WEIGHTS, false, 0, 0, RAGGED>(a, pair, 0, st.pose, nullptr, nullptr, ref_slot, tgt_slot, &ov);
        __syncthreads();
        u.k = k;
        st = update_compute(u, rec, &cur, ulds, false);   // ends with a barrier: every thread has the new state
        if (st.level_done || st.status) break;            // block-uniform (the exit test, src/Tracker.cpp:508)
      }
    }
    // end of a pyramid level: hand-off (src/Tracker.cpp:580-590) and re-arm for the next level (:392-393)
    if (st.status == 0 && ca.level_id[li] != 0) {
      if (!se3_handoff(st.pose, ca.scale_t != 0)) st.status = 1;  // SOPHUS_ENSURE would abort
    }
    st.level_done = 0;
    st.last_error = ca.initial_error;
  }
  if (threadIdx.x == 0) ca.state_out[pair] = st;
}

struct StatsOut { int status, iterations, n_valid; float error; };

// After the last evaluation: its update, the last level's hand-off, and the results (one block per pair).
static __global__ __launch_bounds__(kUpdateBlock) void k_finish(const IterArgs ia, float* __restrict__ poses, StatsOut* __restrict__ stats) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kUpdateLdsBytes];
  const int pair = (int)blockIdx.x + ia.u.pair_base;
  const PairState st = iterate_state(ia, pair, lds, true);
  if (threadIdx.x) return;
  ia.state_out[pair] = st;
  for (int k = 0; k < 4; k++) poses[7 * pair + k] = st.pose.q[k];
  for (int k = 0; k < 3; k++) poses[7 * pair + 4 + k] = st.pose.t[k];
  if (stats) {
    StatsOut o;
    o.status = st.status; o.iterations = st.iters; o.n_valid = st.n_valid; o.error = st.error;
    stats[pair] = o;
  }
}

static __global__ void k_set_pose(PairState* state, Pose pose, float initial_error) {
  if (threadIdx.x || blockIdx.x) return;
  PairState st;
  st.pose = pose;
  st.last_error = initial_error;
  st.error = 0.f;
  st.level_done = 0;
  st.status = 0;
  st.iters = 0;
  st.n_valid = 0;
  state[0] = st;
}

static __global__ void k_init_state(PairState* state, int n, float initial_error) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PairState st;
  pose_identity(st.pose);  // src/Tracker.cpp:385
  st.last_error = initial_error;
  st.error = 0.f;
  st.level_done = 0;
  st.status = 0;
  st.iters = 0;
  st.n_valid = 0;
  state[i] = st;
}

// end of a pyramid level: hand-off (src/Tracker.cpp:580-590) and re-arm for the next level (:392-393)
static __global__ void k_level_end(PairState* state, int n, int lvl, int scale_t, float initial_error) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // `state` points at the first pair of the launch
  if (i >= n) return;
  PairState st = state[i];
  if (st.status == 0 && lvl != 0) {
    if (!se3_handoff(st.pose, scale_t != 0)) st.status = 1;  // SOPHUS_ENSURE would abort
  }
  st.level_done = 0;
  st.last_error = initial_error;
  state[i] = st;
}

static __global__ void k_write_out(const PairState* state, int n, float* poses, StatsOut* stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const PairState st = state[i];
  for (int k = 0; k < 4; k++) poses[7 * i + k] = st.pose.q[k];
  for (int k = 0; k < 3; k++) poses[7 * i + 4 + k] = st.pose.t[k];
  if (stats) {
    StatsOut o;
    o.status = st.status; o.iterations = st.iters; o.n_valid = st.n_valid; o.error = st.error;
    stats[i] = o;
  }
}

// ------------------------------------------------------------------------------------------------------------
// per-stage helpers (parity entry points)
// ------------------------------------------------------------------------------------------------------------

// WarpFunction's unprojection and rigid product for one point of an explicit table, the table's own w (src/Tracker.cpp:
// 1439-1450): o[0..2] = the first three rows, wq = the fourth, (0 0 0 1) * P.  The two arithmetic sets as in warp_point.
template <int AR>
__device__ __forceinline__ void warp_table_point(const LevelK& L, const float* T, const float4 p, float o[3], float& wq) {
  float X, Y;
  if constexpr (AR == kArithLegacy) {
    X = (p.x - L.cx) * L.invfx;
    Y = (p.y - L.cy) * L.invfy;
  } else {
    X = p.x * L.invfx; X = X + L.bx;
    Y = p.y * L.invfy; Y = Y + L.by;
  }
  X = X * p.z;
  Y = Y * p.z;
  if constexpr (AR == kArithLegacy) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float s = T[4 * k] * X;
      s = __builtin_fmaf(T[4 * k + 1], Y, s);
      s = __builtin_fmaf(T[4 * k + 2], p.z, s);
      s = __builtin_fmaf(T[4 * k + 3], p.w, s);
      o[k] = s;
    }
    wq = 0.f * X;  // fourth row of the rigid matrix is (0 0 0 1)
    wq = __builtin_fmaf(0.f, Y, wq);
    wq = __builtin_fmaf(0.f, p.z, wq);
    wq = __builtin_fmaf(1.f, p.w, wq);
  } else {
    const double Xd = (double)X, Yd = (double)Y, zd = (double)p.z, wd = (double)p.w;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double Td[4] = {(double)T[4 * k], (double)T[4 * k + 1], (double)T[4 * k + 2], (double)T[4 * k + 3]};
      o[k] = rigid_row_f64(Td, Xd, Yd, zd, wd);
    }
    const double last[4] = {0.0, 0.0, 0.0, 1.0};
    wq = rigid_row_f64(last, Xd, Yd, zd, wd);
  }
}

// Tracker::WarpFunction on an explicit N x 4 point table (src/Tracker.cpp:1417-1471)
template <int AR>
__global__ void k_warp_table(const float4* __restrict__ pts, float4* __restrict__ out, int n, Pose pose, LevelK L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float T[12];
  pose_to_T12(pose, T);
  const float4 p = pts[i];
  float o[3], wq;
  warp_table_point<AR>(L, T, p, o, wq);
  float u = o[0] * L.fx; u = u / o[2]; u = u + L.cx;
  float v = o[1] * L.fy; v = v / o[2]; v = v + L.cy;
  u = u * wq;
  v = v * wq;
  out[i] = make_float4(u, v, o[2], wq);
}

// LS::update / LS::updateSSE over n rows (src/LeastSquares.cpp:204-209, 148-202) IN THE REFERENCE'S ORDER.  Each of LS's accumulators
// is a chain of f32 additions over the points in call order — a parallel sum cannot reproduce its roundings.  But the 28 chains (21
// upper-triangle entries of A, 6 of b, the error) are independent of one another, and updateSSE's are 28 x 4 lane chains over every
// fourth point: one thread per chain walks the points sequentially, the chains run side by side.  Bit for bit what the reference's
// loop computes (rounds 1-5 folded per-thread partial sums in f64: closer to the exact sum, not the reference's floats); LS is a
// mirror of code the reference never calls, a few hundred microseconds per 10^4 points are of no concern.
//   SSE_ORDER = false: thread s < 28 -> out[s]; products (J_i J_j) w, J_i (r w), (r r) w.
//   SSE_ORDER = true:  thread 4 s + l -> out[4 s + l], points l, l + 4, ...; products (J_i w) J_j, (r w) J_i, (r w) r (:151-199).
template <bool SSE_ORDER>
__global__ __launch_bounds__(128) void k_ls_sequential(const float* __restrict__ J, const float* __restrict__ r,
                                                       const float* __restrict__ w, int n, float* __restrict__ out) {
  const int t = (int)threadIdx.x;
  const int s = SSE_ORDER ? t >> 2 : t, lane = SSE_ORDER ? t & 3 : 0, step = SSE_ORDER ? 4 : 1;
  if (s >= 28) return;
  int i = 0, j = 0;   // accumulator s: the upper triangle row-major (i <= j), then b_0..b_5 (i = s - 21), then the error
  if (s < 21) {
    int q = s;
    while (q >= 6 - i) { q -= 6 - i; i++; }
    j = i + q;
  } else {
    i = s - 21;
  }
  float acc = 0.f;
  for (int p = lane; p < n; p += step) {
    const float wi = w ? w[p] : 1.0f, ri = r[p];
    float term;
    if (s < 21) {
      const float Ji = J[(size_t)p * 6 + i], Jj = J[(size_t)p * 6 + j];
      term = SSE_ORDER ? (Ji * wi) * Jj : (Ji * Jj) * wi;
    } else if (s < 27) {
      const float Ji = J[(size_t)p * 6 + i], rw = ri * wi;
      term = SSE_ORDER ? rw * Ji : Ji * rw;
    } else {
      term = SSE_ORDER ? (ri * wi) * ri : (ri * ri) * wi;
    }
    acc = acc + term;
  }
  out[t] = acc;
}

// ------------------------------------------------------------------------------------------------------------
// explicit point tables (Frame::candidatePoints_[lvl], N x 4 [x y z w]) — what EstimatePose / EstimatePoseFeatures
// iterate over when a sparse producer filled them (src/Tracker.cpp:401, 669).  Same per-point terms and the same
// deterministic reduction as the dense kernel; reference values are gathered at ((int)y1, (int)x1) (:471-477).
// ------------------------------------------------------------------------------------------------------------
struct PointsArgs {
  const float4* pts;        // table of this launch's single pair
  int n_pts;
  int pts_per_block;
};

template <int AR, bool UNIT_FACTORS, bool DUMP, typename AccT>
__global__ __launch_bounds__(kBlock) void k_residual_points(const ResidualArgs a, const PointsArgs pa) {
  const int pair = a.pair_base;
  Pose pose;
  if (a.state) {
    const PairState st = a.state[pair];
    if (st.level_done || st.status) return;
    pose = st.pose;
  } else {
    pose = a.pose;
  }
  WarpK K;
  pose_to_T12(pose, K.T);
  const LevelK L = a.L;
  const size_t ref_off = (size_t)a.ref_slots[pair] * L.n, tgt_off = (size_t)a.tgt_slots[pair] * L.n;
  const uint8_t* __restrict__ I1 = a.img + ref_off;
  const uint8_t* __restrict__ I2 = a.img + tgt_off;
  const int16_t* __restrict__ GX = a.gx + ref_off;
  const int16_t* __restrict__ GY = a.gy + ref_off;
  AccT acc[kAccFloats];
#pragma unroll
  for (int i = 0; i < kAccFloats; i++) acc[i] = (AccT)0;
  uint32_t sum_r2 = 0, n_valid = 0;
  const int p_begin = blockIdx.x * pa.pts_per_block;
  const int p_end = min(p_begin + pa.pts_per_block, pa.n_pts);
  for (int q = p_begin + (int)threadIdx.x; q < p_end; q += kBlock) {
    const float4 P = pa.pts[q];
    // WarpFunction with the table's own w (src/Tracker.cpp:1439-1467)
    float o[3], wq;
    warp_table_point<AR>(L, K.T, P, o, wq);
    float x2 = o[0] * L.fx; x2 = x2 / o[2]; x2 = x2 + L.cx; x2 = x2 * wq;
    float y2 = o[1] * L.fy; y2 = y2 / o[2]; y2 = y2 + L.cy; y2 = y2 * wq;
    const float z2 = o[2];
    float iz = 1.0f / z2;
    bool ok = (y2 > 0.f) && (y2 < (float)L.ih) && (x2 > 0.f) && (x2 < (float)L.iw) && (z2 != 0.f);
    const int ix1 = (int)P.x, iy1 = (int)P.y;
    ok = ok && ix1 >= 0 && ix1 < L.iw && iy1 >= 0 && iy1 < L.ih;  // the reference would read out of bounds
    float J[6];
    int ri = 0;
    if (ok) {
      if (iz < 0.f) iz = 0.f;
      const uint32_t i1x = (uint32_t)(iy1 * L.pitch + ix1);
      int ix2 = round_pos(x2), iy2 = round_pos(y2);
      ix2 = min(ix2, L.iw - 1);
      iy2 = min(iy2, L.ih - 1);
      ri = (int)I2[iy2 * L.pitch + ix2] - (int)I1[i1x];
      pixel_jacobian<AR, UNIT_FACTORS, false, DUMP>(L, a.zf, a.af, x2, y2, iz, (float)GX[i1x], (float)GY[i1x], J);
      accumulate(acc, J, ri);
      sum_r2 += (uint32_t)(ri * ri);
      n_valid += 1;
    }
    if constexpr (DUMP) {
      if (a.dumpV) a.dumpV[q] = ok ? 1 : 0;
      if (a.dumpR) a.dumpR[q] = ok ? (float)ri : 0.f;
      if (a.dumpJ)
        for (int k = 0; k < 6; k++) a.dumpJ[(size_t)q * 6 + k] = ok ? J[k] : 0.f;
    }
  }
  block_reduce_store<AccT>(acc, sum_r2, n_valid, a.partials + ((size_t)pair * a.slices + blockIdx.x) * kRecWords);
}

// The same tables on the general path (robust weights and / or the bilinear sampler: uwt_params::weights, ::sampler): the
// per-stage form of the dense path — k_points_hist, k_scale_stage, k_points_general per evaluation — over table rows.
// One row: WarpFunction with the table's own w, the validity tests of k_residual_points, the residual of either sampler.
template <int AR>
__device__ __forceinline__ bool general_point(const LevelK& L, const float* T, int sampler, const uint8_t* __restrict__ I1,
                                              const uint8_t* __restrict__ I2, const float4 P, float& x2, float& y2, float& iz, float& rf,
                                              uint32_t& i1x) {
  float o[3], wq;
  warp_table_point<AR>(L, T, P, o, wq);
  x2 = o[0] * L.fx; x2 = x2 / o[2]; x2 = x2 + L.cx; x2 = x2 * wq;
  y2 = o[1] * L.fy; y2 = y2 / o[2]; y2 = y2 + L.cy; y2 = y2 * wq;
  const float z2 = o[2];
  iz = 1.0f / z2;
  bool ok = (y2 > 0.f) && (y2 < (float)L.ih) && (x2 > 0.f) && (x2 < (float)L.iw) && (z2 != 0.f);
  const int ix1 = (int)P.x, iy1 = (int)P.y;
  ok = ok && ix1 >= 0 && ix1 < L.iw && iy1 >= 0 && iy1 < L.ih;  // the reference would read out of bounds
  rf = 0.f;
  i1x = 0;
  if (!ok) return false;
  if (iz < 0.f) iz = 0.f;
  i1x = (uint32_t)(iy1 * L.pitch + ix1);
  const int i1 = I1[i1x];
  if (sampler) {
    rf = sample_bilinear(I2, L, x2, y2) - (float)i1;
  } else {
    int ix2 = round_pos(x2), iy2 = round_pos(y2);
    ix2 = min(ix2, L.iw - 1);
    iy2 = min(iy2, L.ih - 1);
    rf = (float)((int)I2[iy2 * L.pitch + ix2] - i1);
  }
  return true;
}

// the scale pass: every valid row's rounded residual into the pair's signed bins (k_resid_hist over a table)
template <int AR>
__global__ __launch_bounds__(kBlock) void k_points_hist(const ResidualArgs a, const PointsArgs pa, const GeneralArgs ga) {
  const int pair = a.pair_base;
  const PairState st = a.state[pair];
  if (st.level_done || st.status) return;
  __shared__ unsigned int h[kHistBins];
  for (int i = threadIdx.x; i < kHistBins; i += kBlock) h[i] = 0;
  __syncthreads();
  float T[12];
  pose_to_T12(st.pose, T);
  const LevelK L = a.L;
  const uint8_t* __restrict__ I1 = a.img + (size_t)a.ref_slots[pair] * L.n;
  const uint8_t* __restrict__ I2 = a.img + (size_t)a.tgt_slots[pair] * L.n;
  const int p_begin = blockIdx.x * pa.pts_per_block, p_end = min(p_begin + pa.pts_per_block, pa.n_pts);
  for (int q = p_begin + (int)threadIdx.x; q < p_end; q += kBlock) {
    float x2, y2, iz, rf;
    uint32_t i1x;
    if (!general_point<AR>(L, T, ga.sampler, I1, I2, pa.pts[q], x2, y2, iz, rf, i1x)) continue;
    atomicAdd(&h[(int)rintf(rf) + 255], 1u);
  }
  __syncthreads();
  unsigned int* gh = ga.hist + (size_t)pair * kHistBins;
  for (int i = threadIdx.x; i < kHistBins; i += kBlock)
    if (h[i]) atomicAdd(&gh[i], h[i]);
}

// the weighted / bilinear accumulation of k_residual_general over a table (records of the `general` kind)
template <int AR, bool UNIT_FACTORS>
__global__ __launch_bounds__(kBlock) void k_points_general(const ResidualArgs a, const PointsArgs pa, const GeneralArgs ga) {
  const int pair = a.pair_base;
  const PairState st = a.state[pair];
  if (st.level_done || st.status) return;
  float T[12];
  pose_to_T12(st.pose, T);
  const LevelK L = a.L;
  const size_t ref_off = (size_t)a.ref_slots[pair] * L.n;
  const uint8_t* __restrict__ I1 = a.img + ref_off;
  const uint8_t* __restrict__ I2 = a.img + (size_t)a.tgt_slots[pair] * L.n;
  const int16_t* __restrict__ GX = a.gx + ref_off;
  const int16_t* __restrict__ GY = a.gy + ref_off;
  const float inv_mad = ga.weights ? ga.scale[pair].inv_mad : 1.f;
  double acc[kAccFloats];
#pragma unroll
  for (int i = 0; i < kAccFloats; i++) acc[i] = 0.0;
  double err = 0.0;
  uint32_t sum_r2 = 0, n_valid = 0;
  const int p_begin = blockIdx.x * pa.pts_per_block, p_end = min(p_begin + pa.pts_per_block, pa.n_pts);
  for (int q = p_begin + (int)threadIdx.x; q < p_end; q += kBlock) {
    float x2, y2, iz, rf;
    uint32_t i1x;
    if (!general_point<AR>(L, T, ga.sampler, I1, I2, pa.pts[q], x2, y2, iz, rf, i1x)) continue;
    float J[6];
    pixel_jacobian<AR, UNIT_FACTORS, false, true>(L, a.zf, a.af, x2, y2, iz, (float)GX[i1x], (float)GY[i1x], J);
    const float w = robust_weight(ga.weights, rf, inv_mad);
    err += (double)rf * (double)(rf * w);         // Residuals.mul(W) for the error (src/Tracker.cpp:500)
    const float rw = (rf * ga.gain) * w;           // :559, :561
    double Jd[6];
#pragma unroll
    for (int k = 0; k < 6; k++) Jd[k] = (double)(w * J[k]);   // :556
    int s = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = i; j < 6; j++, s++) acc[s] = __builtin_fma(Jd[i], Jd[j], acc[s]);
#pragma unroll
    for (int i = 0; i < 6; i++) acc[21 + i] = __builtin_fma(Jd[i], (double)rw, acc[21 + i]);
    const int qr = (int)rintf(rf);
    sum_r2 += (uint32_t)(qr * qr);
    n_valid += 1;
  }
  block_reduce_store<double, true>(acc, sum_r2, n_valid, a.partials + ((size_t)pair * a.slices + blockIdx.x) * kRecWords, err);
}

// ------------------------------------------------------------------------------------------------------------
// sparse point producers
// ------------------------------------------------------------------------------------------------------------

// gradient_ = addWeighted(convertScaleAbs(gx), 0.5, convertScaleAbs(gy), 0.5) (src/Tracker.cpp:1139-1142), u8,
// plus its integer sum for cuda::meanStdDev (:1325).  (a + b)/2 with cvRound's round-half-to-even.
// n = pitch * ih plane elements, laid out like the gradient planes; the sum runs over the image's iw columns alone (the pad
// columns of a pitched row hold nothing of the image).
static __global__ __launch_bounds__(kBlock) void k_grad_mag(const int16_t* __restrict__ gx, const int16_t* __restrict__ gy, int n,
                                                     int pitch, int iw, uint8_t* __restrict__ mag,
                                                     unsigned long long* __restrict__ sum) {
  unsigned int local = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const int ax = min(abs((int)gx[i]), 255), ay = min(abs((int)gy[i]), 255);
    const int s = ax + ay;
    int m = s >> 1;
    if (s & 1) m += (m & 1);
    mag[i] = (uint8_t)m;
    if (i % pitch < iw) local += (unsigned int)m;
  }
  __shared__ unsigned int red[kBlock];
  red[threadIdx.x] = local;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(sum, (unsigned long long)red[0]);  // integer: order-independent
}

// Tracker::ObtainCandidatePoints (src/Tracker.cpp:1314-1362) for a batch of frames, many blocks per frame, three passes:
// gradient_ and its per-frame sum: grid (blocks, frames)
static __global__ __launch_bounds__(kBlock) void k_grad_mag_batch(const int16_t* __restrict__ gx, const int16_t* __restrict__ gy, int n,
                                                           int pitch, int iw, int first_slot, uint8_t* __restrict__ mag,
                                                           unsigned long long* __restrict__ sums) {
  const int f = blockIdx.y;
  const size_t src = (size_t)(first_slot + f) * n, dst = (size_t)f * n;
  unsigned int local = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const int ax = min(abs((int)gx[src + i]), 255), ay = min(abs((int)gy[src + i]), 255);
    const int s = ax + ay;
    int m = s >> 1;
    if (s & 1) m += (m & 1);
    mag[dst + i] = (uint8_t)m;
    if (i % pitch < iw) local += (unsigned int)m;
  }
  __shared__ unsigned int red[kBlock];
  red[threadIdx.x] = local;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&sums[f], (unsigned long long)red[0]);  // integer: order-independent
}

// One thread owns one column x of one row band of one frame and walks its rows top to bottom (a row of the block's
// columns is one coalesced read); a cell is kept iff gradient_ > mean + threshold and, with a depth plane, the byte the
// reference reads there is not zero (src/Tracker.cpp:1336-1347).  WRITE = false: counts[f][x * bands + band];
// WRITE = true: the points, at the offsets an exclusive scan of the counts in that (x, band) order gives — which is the
// reference's order, x outer, y inner (src/Tracker.cpp:1334-1335).
template <bool WRITE>
__global__ __launch_bounds__(kBlock) void k_candidates_batch(const uint8_t* __restrict__ mag, const uint16_t* __restrict__ depth,
                                                             int first_slot, int pitch, int iw, int ih, int w, int h, int bands,
                                                             const unsigned long long* __restrict__ sums, double threshold,
                                                             int* __restrict__ counts, const int* __restrict__ offsets,
                                                             float4* __restrict__ out, int cap) {
  // w x h: the level's point grid (w_[lvl] x h_[lvl], the loops of :1334-1335); iw x ih: its image (the mean of :1324 runs over
  // the whole gradient_ Mat); rows of `pitch` elements
  const int f = blockIdx.z, band = blockIdx.y, x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= w) return;
  const size_t n = (size_t)pitch * ih;
  const double thres = (double)sums[f] / (double)((size_t)iw * ih) + threshold;  // cuda::meanStdDev mean + GRADIENT_THRESHOLD (:1325-1327)
  const uint8_t* m = mag + (size_t)f * n;
  const uint16_t* dp = depth ? depth + (size_t)(first_slot + f) * n : nullptr;
  const int rows = (h + bands - 1) / bands, y0 = band * rows, y1 = min(y0 + rows, h);
  int k = WRITE ? offsets[(size_t)f * w * bands + (size_t)x * bands + band] : 0;
  float4* o = WRITE ? out + (size_t)f * cap : nullptr;
  for (int y = y0; y < y1; y++) {
    if (!((double)m[(size_t)y * pitch + x] > thres)) continue;
    float z = 1.0f;
    if (dp) {  // the reference indexes the 16-bit plane through at<uchar> (:1339, :1344): byte x of row y
      const uint8_t b = reinterpret_cast<const uint8_t*>(dp + (size_t)y * pitch)[x];
      if (b == 0) continue;
      z = (float)b * 0.0002f;
    }
    if (WRITE) {
      if (k < cap) o[k] = make_float4((float)x, (float)y, z, 1.0f);
    }
    k++;
  }
  if (!WRITE) counts[(size_t)f * w * bands + (size_t)x * bands + band] = k;
}

// exclusive scan of the m = w * bands counts of one frame (one block per frame), total to totals[f]
static __global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ counts, int m, int* __restrict__ offsets,
                                                      int* __restrict__ totals) {
  __shared__ int part[1024];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int* c = counts + (size_t)f * m;
  int* o = offsets + (size_t)f * m;
  const int per = (m + 1023) / 1024, i0 = min(tid * per, m), i1 = min(i0 + per, m);
  int s = 0;
  for (int i = i0; i < i1; i++) s += c[i];
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {   // Hillis-Steele inclusive scan of the 1024 partial sums
    const int v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = tid ? part[tid - 1] : 0;
  for (int i = i0; i < i1; i++) { o[i] = run; run += c[i]; }
  if (tid == 1023) totals[f] = part[1023];
}

// Tracker::ObtainPatchesPoints (src/Tracker.cpp:1178-1257): level 0, <= 200 key points, 11x11 patches
// ("patch_size_ - 1 / 2" = 5), x-major inside a patch, key points in order.  One thread per key point counts, a
// serial prefix orders, the thread then writes its patch.
static __global__ __launch_bounds__(256) void k_patch_points(const float2* __restrict__ kp, int n_kp, const uint16_t* __restrict__ depth0,
                                                      int pitch, int w, int h, float4* __restrict__ out, int cap, int* __restrict__ count) {
  __shared__ int cnt[256];
  const int q = threadIdx.x;
  const int start_point = 5;
  float x = 0.f, y = 0.f, z = 1.0f;
  bool live = q < n_kp && q < 200;
  if (live) {
    x = kp[q].x; y = kp[q].y;
    if (depth0) {
      const int d = (int)(int16_t)depth0[(size_t)(int)y * pitch + (int)x];  // at<short>(y, x) != 0 (:1202)
      if (d == 0) live = false;
      z = (float)d * 0.0002f * 1.0f;                                      // * factor_depth * factor_lvl (:1204)
    }
  }
  int k = 0;
  if (live)
    for (int i = (int)(x - (float)start_point); (float)i <= x + (float)start_point; i++)
      for (int j = (int)(y - (float)start_point); (float)j <= y + (float)start_point; j++)
        if (i > 0 && i < w && j > 0 && j < h) k++;
  cnt[q] = k;
  __syncthreads();
  if (q == 0) {
    int run = 0;
    for (int i = 0; i < 256; i++) { const int v = cnt[i]; cnt[i] = run; run += v; }
    *count = run;
  }
  __syncthreads();
  int o = cnt[q];
  if (live)
    for (int i = (int)(x - (float)start_point); (float)i <= x + (float)start_point; i++)
      for (int j = (int)(y - (float)start_point); (float)j <= y + (float)start_point; j++)
        if (i > 0 && i < w && j > 0 && j < h) {
          if (o < cap) out[o] = make_float4((float)i, (float)j, z, 1.0f);
          o++;
        }
}


// Tracker::AddPatchPointsFeatures (src/Tracker.cpp:599-629): the table itself, then for every point, in table order, the
// cells of the (2 * start + 1)^2 patch around its rounded position (x outer, y inner) that lie strictly inside the level
// (i > 0, j > 0) and are not the centre, each carrying the point's z and w = 1.  One block: chunks of 256 points, the
// cells of a chunk counted per point and scanned in point order, so that the output order is the reference's push_back order.
static __global__ __launch_bounds__(256) void k_add_patch_points(const float4* __restrict__ pts, int n, int w, int h, int start,
                                                          float4* __restrict__ out, int cap, int* __restrict__ count) {
  __shared__ int cnt[256];
  __shared__ int base;
  const int q = threadIdx.x;
  for (int i = q; i < n; i += 256)
    if (i < cap) out[i] = pts[i];   // candidatePoints.clone() (:601)
  if (q == 0) base = n;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 256) {
    const int idx = c0 + q;
    float x = 0.f, y = 0.f, z = 0.f;
    int k = 0;
    if (idx < n) {
      const float4 p = pts[idx];
      x = roundf(p.x); y = roundf(p.y); z = p.z;   // :607-609
      for (int i = (int)(x - (float)start); (float)i <= x + (float)start; i++)
        for (int j = (int)(y - (float)start); (float)j <= y + (float)start; j++)
          if (i > 0 && i < w && j > 0 && j < h && !((float)i == x && (float)j == y)) k++;
    }
    cnt[q] = k;
    __syncthreads();
    if (q == 0) {
      int run = base;
      for (int i = 0; i < 256; i++) { const int v = cnt[i]; cnt[i] = run; run += v; }
      base = run;
    }
    __syncthreads();
    int o = cnt[q];
    if (idx < n)
      for (int i = (int)(x - (float)start); (float)i <= x + (float)start; i++)
        for (int j = (int)(y - (float)start); (float)j <= y + (float)start; j++)
          if (i > 0 && i < w && j > 0 && j < h && !((float)i == x && (float)j == y)) {
            if (o < cap) out[o] = make_float4((float)i, (float)j, z, 1.0f);   // Mat::ones(1, 4): w = 1 (:614-617)
            o++;
          }
    __syncthreads();
  }
  if (q == 0) *count = base;
}

// ------------------------------------------------------------------------------------------------------------
// frame ingest: cv::remap(raw, undistorted, map1, map2, INTER_LINEAR) + ROI crop of System::AddFrame
// (src/System.cpp:231-235) fused: each output pixel of the crop window gathers its 2x2 source patch through the
// fixed-point maps (CV_16SC2 + 5-bit fractions) and blends with the 15-bit weights, border constant 0.
// ------------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kBlock) void k_remap_crop(const uint8_t* __restrict__ src, int sw, int sh, size_t s_stride,
                                                       const short2* __restrict__ map1, const uint16_t* __restrict__ map2,
                                                       int mw, int x0, int y0, uint8_t* __restrict__ dst, int cw, int ch, int dst_pitch) {
  const int q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= cw * ch) return;
  const int oy = q / cw, ox = q - oy * cw;
  const size_t m = (size_t)(y0 + oy) * mw + (x0 + ox);
  const short2 s = map1[m];
  const int sx = s.x, sy = s.y;
  const int fxy = map2[m] & 1023, fx = fxy & 31, fy = fxy >> 5;
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  const bool xin0 = sx >= 0 && sx < sw, xin1 = sx + 1 >= 0 && sx + 1 < sw;
  const bool yin0 = sy >= 0 && sy < sh, yin1 = sy + 1 >= 0 && sy + 1 < sh;
  const int p00 = (xin0 && yin0) ? src[(size_t)sy * s_stride + sx] : 0;
  const int p01 = (xin1 && yin0) ? src[(size_t)sy * s_stride + sx + 1] : 0;
  const int p10 = (xin0 && yin1) ? src[(size_t)(sy + 1) * s_stride + sx] : 0;
  const int p11 = (xin1 && yin1) ? src[(size_t)(sy + 1) * s_stride + sx + 1] : 0;
  const int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
  dst[(size_t)oy * dst_pitch + ox] = (uint8_t)min(v, 255);
}

// Visualizer::UpdateMessages pose accumulation (src/Visualizer.cpp:304-325): final_i = final_{i-1} * SE3(q_i, s·t_i).
// Strictly sequential (float SE(3) products are not associative to the last bit), one lane; n is a trajectory
// length, not a pixel count.
static __global__ void k_trajectory(const float* __restrict__ poses, int n, Pose prev, float t_scale, int reference_axes,
                             float* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  for (int i = 0; i < n; i++) {
    const float* p = poses + 7 * (size_t)i;
    Pose cur, fin;
    const float len = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);  // SE3(q, t) normalises q
    cur.q[0] = p[0] / len; cur.q[1] = p[1] / len; cur.q[2] = p[2] / len; cur.q[3] = p[3] / len;
    cur.t[0] = t_scale * p[4]; cur.t[1] = t_scale * p[5]; cur.t[2] = t_scale * p[6];
    se3_mul(prev, cur, fin);
    prev = fin;
    float* o = out + 7 * (size_t)i;
    o[0] = fin.q[0]; o[1] = fin.q[1]; o[2] = fin.q[2]; o[3] = fin.q[3];
    if (reference_axes) { o[4] = -fin.t[2]; o[5] = -fin.t[0]; o[6] = -fin.t[1]; }
    else { o[4] = fin.t[0]; o[5] = fin.t[1]; o[6] = fin.t[2]; }
  }
}

// The same accumulation as a prefix product (SE(3) composition is associative; in floats the grouping shows in the last
// bits, so this is the default "clean" mode, not the reference-visualiser one): one block, every thread multiplies its run
// of poses, a Hillis-Steele scan over the 1024 run products, then every thread replays its run behind its prefix.
static __global__ __launch_bounds__(1024) void k_trajectory_scan(const float* __restrict__ poses, int n, Pose start, float t_scale,
                                                          int reference_axes, float* __restrict__ out) {
  __shared__ Pose part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024, i0 = min(tid * per, n), i1 = min(i0 + per, n);
  auto load = [&](int i, Pose& cur) {
    const float* p = poses + 7 * (size_t)i;
    const float len = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);  // SE3(q, t) normalises q
    cur.q[0] = p[0] / len; cur.q[1] = p[1] / len; cur.q[2] = p[2] / len; cur.q[3] = p[3] / len;
    cur.t[0] = t_scale * p[4]; cur.t[1] = t_scale * p[5]; cur.t[2] = t_scale * p[6];
  };
  Pose acc;
  pose_identity(acc);
  for (int i = i0; i < i1; i++) {
    Pose cur, nxt;
    load(i, cur);
    se3_mul(acc, cur, nxt);
    acc = nxt;
  }
  part[tid] = acc;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    Pose left, mine = part[tid], prod;
    const bool has = tid >= d;
    if (has) left = part[tid - d];
    __syncthreads();
    if (has) { se3_mul(left, mine, prod); part[tid] = prod; }
    __syncthreads();
  }
  Pose prev;
  if (tid) { Pose pre = part[tid - 1]; se3_mul(start, pre, prev); } else prev = start;
  for (int i = i0; i < i1; i++) {
    Pose cur, fin;
    load(i, cur);
    se3_mul(prev, cur, fin);
    prev = fin;
    float* o = out + 7 * (size_t)i;
    o[0] = fin.q[0]; o[1] = fin.q[1]; o[2] = fin.q[2]; o[3] = fin.q[3];
    if (reference_axes) { o[4] = -fin.t[2]; o[5] = -fin.t[0]; o[6] = -fin.t[1]; }
    else { o[4] = fin.t[0]; o[5] = fin.t[1]; o[6] = fin.t[2]; }
  }
}

static __global__ void k_se3_ops(int op, const float* in_a, const float* in_b, float* out, int* flag) {
  if (threadIdx.x || blockIdx.x) return;
  Pose a, b, o;
  if (op == 0) {  // exp
    se3_exp(in_a, o);
    for (int k = 0; k < 4; k++) out[k] = o.q[k];
    for (int k = 0; k < 3; k++) out[4 + k] = o.t[k];
  } else if (op == 1) {  // mul
    for (int k = 0; k < 4; k++) { a.q[k] = in_a[k]; b.q[k] = in_b[k]; }
    for (int k = 0; k < 3; k++) { a.t[k] = in_a[4 + k]; b.t[k] = in_b[4 + k]; }
    se3_mul(a, b, o);
    for (int k = 0; k < 4; k++) out[k] = o.q[k];
    for (int k = 0; k < 3; k++) out[4 + k] = o.t[k];
  } else if (op == 2) {  // matrix
    for (int k = 0; k < 4; k++) a.q[k] = in_a[k];
    for (int k = 0; k < 3; k++) a.t[k] = in_a[4 + k];
    float T[12];
    pose_to_T12(a, T);
    for (int k = 0; k < 12; k++) out[k] = T[k];
    out[12] = 0.f; out[13] = 0.f; out[14] = 0.f; out[15] = 1.f;
  } else if (op == 3 || op == 4) {  // handoff (4: also scale t)
    for (int k = 0; k < 4; k++) a.q[k] = in_a[k];
    for (int k = 0; k < 3; k++) a.t[k] = in_a[4 + k];
    *flag = se3_handoff(a, op == 4) ? 1 : 0;
    for (int k = 0; k < 4; k++) out[k] = a.q[k];
    for (int k = 0; k < 3; k++) out[4 + k] = a.t[k];
  } else if (op == 5 || op == 6) {  // "A.inv() * b": in_a = A(36), in_b = b(6); out = delta(6) + Ainv(36); 6: the legacy set
    float d[6], Ai[36];
    *flag = solve_delta(in_a, in_b, d, Ai, op == 6) ? 1 : 0;
    for (int k = 0; k < 6; k++) out[k] = d[k];
    for (int k = 0; k < 36; k++) out[6 + k] = Ai[k];
  }
}

}  // namespace uwt
