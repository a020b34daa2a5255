// uwt_capi.hip — host side of libuwt_hip.so: the C ABI declared in include/uwt.h over the gfx950 kernels.
// HIP only; there is no CPU path in this library.
#include "../../include/uwt.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "uwt_launch.h"

using namespace uwt;

static_assert(sizeof(StatsOut) == sizeof(uwt_stats), "uwt_stats layout");

struct uwt_ctx {
  uwt_params p;
  uwt_level info[UWT_MAX_LEVELS];
  LevelK lv[UWT_MAX_LEVELS];
  int vecl[UWT_MAX_LEVELS];             // pixels per vector group at each level: 4 (rows are pitched to whole groups of four)
  bool whole = true;                    // the level-0 size is divisible by 2^(n_levels-1): every level's image is its grid, every cell of
                                        // the resize chain whole (the one-launch pyramid forms apply)
  int slices[UWT_MAX_LEVELS];
  int groups_per_block[UWT_MAX_LEVELS];
  hipStream_t stream = nullptr;
  // Side stream of uwt_track_batch_async: the gradients of the finer levels (HBM-bound) run beside the first, coarse
  // iterations of the alignment (VALU-bound), which only read the coarsest iterated level (uwt_tuning::overlap_gradients).
  hipStream_t side = nullptr;
  static constexpr int kMaxParts = 4;
  int dep_first = 0, dep_n = 0;         // slot range the running tracker call depends on (track_batch_enqueue)
  hipStream_t part_stream[kMaxParts] = {};   // compute streams of parts 1.. of a split batch (part 0: `stream`)
  hipEvent_t ev_fork = nullptr, ev_join[kMaxParts] = {};
  // launch-shape switches (uwt_tuning; defaults: default_tuning()).  split: parts a fixed-schedule batch is cut into (1 = one
  // stream); split_min: pairs per part at least; stream_bytes: a level whose planes of the whole batch exceed this is read
  // non-temporally; split_min_px: level-0 pixels of the batch at least — below, a launch is too short for a second stream to pay
  // (the host enqueues twice as many)
  uwt_tuning tn;
  hipEvent_t ev_pyramids = nullptr, ev_side_done = nullptr, ev_level[UWT_MAX_LEVELS] = {};
  uint8_t* img[UWT_MAX_LEVELS] = {};
  uint16_t* depth[UWT_MAX_LEVELS] = {};
  int16_t* gx[UWT_MAX_LEVELS] = {};
  int16_t* gy[UWT_MAX_LEVELS] = {};
  PairState* state = nullptr;
  int* d_ref = nullptr;
  int* d_tgt = nullptr;
  // pinned staging of the pair lists, a ring of kPairStages [ref(max_pairs) | tgt(max_pairs)] blocks: a new list is
  // written to the next block while the asynchronous copy of the previous one may still be reading its own
  static constexpr int kPairStages = 4;
  int* h_pairs = nullptr;
  hipEvent_t ev_pairs[kPairStages] = {};
  int pair_stage = 0;                   // block holding the lists that are on the device
  int n_pairs_cached = 0;
  // Copy stream + slot-range dependencies (uwt_upload_frames_async): `busy` = compute work enqueued on `stream` that
  // reads or writes a slot range, `fresh` = uploads enqueued on `copy` into a slot range.  An upload waits for the busy
  // entries it overlaps, a compute call for the fresh ones; both rings are in stream order, so once an entry has been
  // dropped the oldest survivor stands for everything before it.
  struct SlotDep { int first = 0, n = 0; hipEvent_t ev = nullptr; bool used = false; };
  static constexpr int kDeps = 8;
  hipStream_t copy = nullptr;
  SlotDep busy[kDeps], fresh[kDeps];
  int busy_next = 0, fresh_next = 0;
  long long ticket_seq = 0;              // compute calls noted so far; busy_seq[i] = the call ring entry i stands for
  long long busy_seq[kDeps] = {};
  bool busy_dropped = false, fresh_dropped = false;
  uint32_t* partials = nullptr;
  uint32_t* partials2 = nullptr;        // the other parity of the chained (k_iterate) flow
  PairState* state2 = nullptr;
  size_t partial_records = 0;
  float* d_poses = nullptr;
  StatsOut* d_stats = nullptr;
  unsigned int* hist = nullptr;         // general path: [pair][2][kHistBins]
  PairScale* scale = nullptr;           // general path: [pair]
  int* d_active = nullptr;              // early-exit polling counters
  unsigned int* d_tickets = nullptr;    // tail update: one counter per pair, zero between launches
  // tn.tail_update: the update in the tail of the residual launch instead of a k_gn_update launch: 1 = where a batch runs as
  // parts on streams of their own (the tail's ~10 us of dependent round trips and the solve run under the other part's
  // launches: +1.2 % at 1024 pairs, +3 % with Huber weights at 256; on one stream the tail is exposed at the end of every
  // launch and loses ~3 us per evaluation to the update launch), 0 = never, 2 = always.  tn.target_blocks: blocks per residual
  // launch the batch-dependent slicing aims at; 0: 1024 for a batch that runs as two halves (one block per slot of the chip),
  // else 4096
  int* h_active = nullptr;              // pinned
  void* scratch = nullptr;              // per-stage entry points
  size_t scratch_bytes = 0;
  void* stage[2] = {nullptr, nullptr};  // uploads of frames whose rows are pitched on the device: [0] context stream, [1] copy stream
  size_t stage_bytes[2] = {0, 0};
  bool profiling = false;
  int spec_budget = 0;                  // speculative launching: evaluations a level gets (0: first_poll + 1); doubled when an alignment
                                        // was cut short, halved again after kSpecCalm calls in a row that were not
  int spec_calm = 0;
  static constexpr int kSpecCalm = 64;
  bool speculate = false;               // set by the synchronous entry for one or two pairs: launch a level's usual number of
                                        // iterations without reading back, check once at the end, redo conservatively if cut short
  // the synchronous small-batch call: results and the cut-short flag are written by the last kernel straight into this
  // page-locked block (no device-to-host copies), and one or two pairs travel in the kernel arguments (no pair list upload)
  static constexpr int kSmallBatch = 8;
  struct SmallResults { float poses[kSmallBatch * 7]; StatsOut stats[kSmallBatch]; int cut; };
  SmallResults* h_small = nullptr;      // pinned, device-visible
  SmallResults* d_small = nullptr;      // its device address
  bool inline_pairs = false;
  bool deferred = false;                // uwt_set_deferred: stage calls return once enqueued
  int pair_slots[4] = {0, 0, 0, 0};
  unsigned poll_seq = 0;                // batch path: read-backs alternate between two counters / events (taken one evaluation late)
  hipEvent_t ev_poll[2] = {};
  const uint32_t* prof_records = nullptr;
  bool compute_only = false;            // uwt_profile_enable(ctx, 2): residual launches run their no-memory diagnostic twin
  std::vector<hipEvent_t> ev_pool;      // start/stop pairs
  size_t ev_used = 0;
  double prof_ms = 0.0;
  long long prof_launches = 0, prof_pixels = 0;
  double prof_level_ms[UWT_MAX_LEVELS] = {};        // the same durations by pyramid level (uwt_profile_read_levels)
  long long prof_level_launches[UWT_MAX_LEVELS] = {};
  std::vector<int> prof_ev_level, prof_ev_evals;                      // level of the launch each event pair brackets
  int prof_slices = 0, prof_pairs = 0;   // slicing of the last profiled residual launch (uwt_profile_clock)
  std::string last_error;
};

namespace {

// x VEC pixels per thread at the finest slicing, the one a single pair runs with: short blocks, at most kMaxSlices of them per
// level (the records the next launch's fold pulls in).  A lone pair's evaluation is a chain of latencies, and every further group a
// thread walks adds a dependent gather round trip to it: ONE group per thread (round 6; 2 until then) wherever that stays under
// kMaxSlices — 640 x 480: level 1 in 75 blocks, level 2 in 19; level 0 would need 300 and keeps two groups (150 blocks: 300 records
// to fold cost more than the second round trip, measured) — takes the 4 x 10 alignment of one pair from 0.411 to 0.392 ms and the
// reference schedule from 0.126 to 0.120 ms (profiles/r06/EXPERIMENTS.md 10).  Batches coarsen the slicing in enqueue_estimate.
// The f64 partial sums group differently with the slicing — 1e-16 relative, far below the f32 rounding of A and b.
constexpr int kGroupsPerThread = 1;

int fail(uwt_ctx* c, int code, const std::string& msg) {
  if (c) c->last_error = msg;
  return code;
}

#define HIPCHK(ctx, expr)                                                                                   \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess)                                                                                   \
      return fail(ctx, UWT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                     \
  } while (0)

// the defaults of uwt_tuning (include/uwt.h); measured choices, see the comments at their uses
uwt_tuning default_tuning() {
  uwt_tuning t;
  std::memset(&t, 0, sizeof(t));
  t.split = 2;
  t.split_min = 8;
  t.split_min_px = 32LL * 640 * 480;
  t.stream_bytes = 200LL << 20;
  t.tail_update = 1;
  t.target_blocks = 0;
  t.coarse = 1;
  t.coarse_batch_px = kCoarseMaxPixels;   // e.g. level 3 of 640x480: +0.8 % on the default batch; larger levels lose (2 waves / SIMD)
  t.coarse_weighted = 1;
  t.overlap_gradients = 1;
  t.first_poll = 3;
  t.chained = -1;
  t.speculation = 1;
  t.fused_stages = 1;
  t.pyramid_batch = 1;
  t.typed_loads = 1;
  return t;
}

int ensure_scratch(uwt_ctx* c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return UWT_OK;
  if (c->scratch) HIPCHK(c, hipFree(c->scratch));
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  HIPCHK(c, hipMalloc(&c->scratch, bytes));
  c->scratch_bytes = bytes;
  return UWT_OK;
}

bool slot_range_ok(const uwt_ctx* c, int first, int n) { return first >= 0 && n >= 0 && (long long)first + n <= c->p.max_frames; }

// Tracker::InitializePyramid (src/Tracker.cpp:297-340): fx halves in double then narrows (:317);
// cx_l = (cx0 + 0.5) / 2^l - 0.5 evaluated in double (:319); invfx = 1 / fx in float (:328).
void init_levels(uwt_ctx* c) {
  const uwt_params& p = c->p;
  float fx = p.fx, fy = p.fy;
  for (int l = 0; l < p.n_levels; l++) {
    if (l > 0) {
      fx = (float)((double)fx * 0.5);
      fy = (float)((double)fy * 0.5);
    }
    uwt_level& I = c->info[l];
    I.w = p.width >> l;    // the point grid: w_[lvl], h_[lvl] (src/Tracker.cpp:312-313)
    I.h = p.height >> l;
    // the level's image: the cv::resize(.., Size(), 0.5, 0.5) chain of src/System.cpp:246-251, dsize = cvRound(ssize * 0.5)
    I.img_w = l == 0 ? p.width : (int32_t)std::lrint((double)c->info[l - 1].img_w * 0.5);   // (lrint: half to even, as cvRound)
    I.img_h = l == 0 ? p.height : (int32_t)std::lrint((double)c->info[l - 1].img_h * 0.5);
    I.pitch = (I.img_w + 3) & ~3;
    I.fx = fx;
    I.fy = fy;
    I.cx = l == 0 ? p.cx : (float)(((double)p.cx + 0.5) / (double)(1 << l) - 0.5);
    I.cy = l == 0 ? p.cy : (float)(((double)p.cy + 0.5) / (double)(1 << l) - 0.5);
    I.invfx = 1.0f / fx;
    I.invfy = 1.0f / fy;
    LevelK& L = c->lv[l];
    L.pitch = I.pitch; L.iw = I.img_w; L.ih = I.img_h; L.gw = I.w; L.gh = I.h;
    L.n = I.pitch * I.img_h;
    L.ng = I.pitch * I.h;
    L.fx = I.fx; L.fy = I.fy; L.cx = I.cx; L.cy = I.cy; L.invfx = I.invfx; L.invfy = I.invfy;
    // "(col - cx) * invfx" (src/Tracker.cpp:1439): MatOp_AddEx::multiply scales s = -cx by invfx in double, convertTo narrows
    L.bx = (float)(-(double)I.cx * (double)I.invfx);
    L.by = (float)(-(double)I.cy * (double)I.invfy);
    L.zscale = (float)((double)p.depth_scale / std::pow(2.0, (double)l));  // src/Tracker.cpp:1266
    L.magic = (uint32_t)((0x100000000ull + (uint64_t)I.pitch - 1) / (uint64_t)I.pitch);
  }
  const int div = 1 << (p.n_levels - 1);
  c->whole = p.width % div == 0 && p.height % div == 0;
}

// One step of the resize chain: level plane `src` (sw x sh, rows of src_pitch) -> `dst` (dw x dh = cvRound halves, rows of
// dst_pitch).  src/dst point at slot 0 of the level planes; the frames processed are slots[0..n) if given, else first_slot..+n.
// Whole cells in tight rows of whole groups of four: k_halve; every other size: k_resize_half.
template <typename T>
int launch_resize(uwt_ctx* c, const T* src, T* dst, int sw, int sh, int src_pitch, int dw, int dh, int dst_pitch, size_t sfs,
                  size_t dfs, int n_frames, const int* d_slots = nullptr, int first_slot = 0) {
  if (n_frames == 0) return UWT_OK;
  if (sw == 2 * dw && sh == 2 * dh && dw % 4 == 0) {
    const int groups = (dw / 4) * dh;
    hipLaunchKernelGGL((k_halve<T, 4>), dim3((groups + kBlock - 1) / kBlock, n_frames), dim3(kBlock), 0, c->stream, src,
                       dst, dw, dh, src_pitch, dst_pitch, sfs, dfs, d_slots, first_slot);
  } else {
    const int groups = (dst_pitch / 4) * dh;
    hipLaunchKernelGGL((k_resize_half<T>), dim3((groups + kBlock - 1) / kBlock, n_frames), dim3(kBlock), 0, c->stream, src,
                       dst, sw, sh, src_pitch, dw, dh, dst_pitch, sfs, dfs, d_slots, first_slot);
  }
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

// rows of w elements, src_pitch elements apart, into rows dst_pitch elements apart (pad columns are never read): four elements per thread
template <typename T>
__global__ __launch_bounds__(kBlock) void k_spread_rows(const T* __restrict__ src, T* __restrict__ dst, int w, size_t src_pitch, size_t dst_pitch,
                                                        size_t rows) {
  const int per_row = (w + 3) >> 2;
  const size_t total = rows * (size_t)per_row;
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const size_t r = i / (size_t)per_row;
    const int x = (int)(i - r * (size_t)per_row) * 4;
    const T* s = src + r * src_pitch + x;
    T* d = dst + r * dst_pitch + x;
    if (x + 3 < w) {   // a whole group: the device rows start on multiples of four elements (dst_pitch % 4 == 0), so one store
      const T v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
      if constexpr (sizeof(T) == 1) {
        *reinterpret_cast<uint32_t*>(d) = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
      } else {
        *reinterpret_cast<uint2*>(d) = make_uint2((uint32_t)v0 | ((uint32_t)v1 << 16), (uint32_t)v2 | ((uint32_t)v3 << 16));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (x + j < w) d[j] = s[j];
    }
  }
}

// the staging area of a stream's uploads (which: 0 the context stream, 1 the copy stream); it only ever grows, and growing waits for
// the stream that may still read the old one
static int ensure_stage(uwt_ctx* c, int which, size_t bytes, hipStream_t s) {
  if (bytes <= c->stage_bytes[which]) return UWT_OK;
  if (c->stage[which]) {
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipFree(c->stage[which]));
  }
  c->stage[which] = nullptr;
  c->stage_bytes[which] = 0;
  HIPCHK(c, hipMalloc(&c->stage[which], bytes));
  c->stage_bytes[which] = bytes;
  return UWT_OK;
}

// `rows` host rows of w elements, src_stride BYTES apart, into device rows dst_pitch elements apart: ONE linear copy of the host span
// (first byte of the first row to last byte of the last; what lies between the rows of a strided view belongs to its parent image)
// into the stream's staging area, then k_spread_rows.  A 2-D copy costs about 5 us PER ROW on this runtime (measured: 138
// alignments/s streamed at 725 x 465 against 50 k through this path).  reserve: bytes the staging area should hold at least.
static int rows_in(uwt_ctx* c, void* dst, size_t dst_pitch, const void* host, size_t elem, size_t w, size_t src_stride, size_t rows,
                   hipStream_t s, int which, size_t reserve) {
  const size_t span = (rows - 1) * src_stride + w * elem;
  int st = ensure_stage(c, which, std::max(span, reserve), s);
  if (st) return st;
  HIPCHK(c, hipMemcpyAsync(c->stage[which], host, span, hipMemcpyHostToDevice, s));
  const size_t work = rows * ((w + 3) / 4);
  const unsigned blocks = (unsigned)std::min<size_t>((work + kBlock - 1) / kBlock, 1u << 16);
  if (elem == 1)
    hipLaunchKernelGGL(k_spread_rows<uint8_t>, dim3(blocks), dim3(kBlock), 0, s, (const uint8_t*)c->stage[which], (uint8_t*)dst, (int)w, src_stride,
                       dst_pitch, rows);
  else
    hipLaunchKernelGGL(k_spread_rows<uint16_t>, dim3(blocks), dim3(kBlock), 0, s, (const uint16_t*)c->stage[which], (uint16_t*)dst, (int)w,
                       src_stride / 2, dst_pitch, rows);
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

// n tightly packed level-0 frames (width x height) into slots first_slot.. of a level-0 plane: one linear copy where the device
// rows are tight too (width a multiple of 4), through the staging area otherwise (a slot is pitch * height)
static int copy_frames_in(uwt_ctx* c, void* plane0, const void* host, size_t elem, int first_slot, int n, hipStream_t s, int which) {
  const size_t w = c->p.width, h = c->p.height, pitch = c->lv[0].pitch;
  unsigned char* dst = (unsigned char*)plane0 + (size_t)first_slot * c->lv[0].n * elem;
  if (pitch == w) {
    HIPCHK(c, hipMemcpyAsync(dst, host, w * h * elem * n, hipMemcpyHostToDevice, s));
    return UWT_OK;
  }
  return rows_in(c, dst, pitch, host, elem, w, w * elem, h * (size_t)n, s, which, w * h * n * (c->p.has_depth ? 2 : 1));   // (the depth frames of the call follow)
}

// one level-0 frame whose host rows are row_stride BYTES apart (a cv::Mat view: Frame::images_[0] = distortion(ROI), src/System.cpp:235)
static int copy_strided_frame_in(uwt_ctx* c, void* plane0, const void* host, size_t elem, size_t row_stride, int slot, hipStream_t s) {
  const size_t w = c->p.width, h = c->p.height, pitch = c->lv[0].pitch;
  unsigned char* dst = (unsigned char*)plane0 + (size_t)slot * c->lv[0].n * elem;
  if (row_stride == w * elem) return copy_frames_in(c, plane0, host, elem, slot, 1, s, 0);
  if (row_stride % elem == 0 && row_stride <= 4 * w * elem)   // the span crosses whole: at most four times the frame's bytes
    return rows_in(c, dst, pitch, host, elem, w, row_stride, h, s, 0, 0);
  HIPCHK(c, hipMemcpy2DAsync(dst, pitch * elem, host, row_stride, w * elem, h, hipMemcpyHostToDevice, s));   // a column out of a very wide parent
  return UWT_OK;
}

// cv::resize(src, dst, Size(), 0.5, 0.5) of one host image of any size through pitched scratch planes (rows padded to whole
// groups of four, as the context's level planes are); dst is dw x dh
template <typename T>
int resize_half_host(uwt_ctx* c, const T* src, int sw, int sh, T* dst, int dw, int dh) {
  const size_t sp = ((size_t)sw + 3) & ~(size_t)3, dp = ((size_t)dw + 3) & ~(size_t)3, es = sizeof(T);
  const size_t off = (sp * sh * es + 255) & ~(size_t)255;
  int st = ensure_scratch(c, off + dp * dh * es + 64);
  if (st) return st;
  unsigned char* d = (unsigned char*)c->scratch;
  HIPCHK(c, hipMemcpy2DAsync(d, sp * es, src, (size_t)sw * es, (size_t)sw * es, sh, hipMemcpyHostToDevice, c->stream));
  st = launch_resize<T>(c, (const T*)d, (T*)(d + off), sw, sh, (int)sp, dw, dh, (int)dp, sp * sh, dp * dh, 1);
  if (st) return st;
  HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)dw * es, d + off, dp * es, (size_t)dw * es, dh, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWT_OK;
}

constexpr int kFewFrames = 8;   // up to here a frame set takes the one-launch forms (k_pyramid_all, k_scharr3_levels)

template <typename T>
void launch_pyramid_all(uwt_ctx* c, T* const* planes_in, T* const* planes, int n, const int* d_slots, int first_slot) {
  PyramidArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.src = planes_in[0];
  for (int l = 0; l < c->p.n_levels; l++) {
    a.dst[l] = planes[l];
    a.stride[l] = c->lv[l].n;
    a.pitch[l] = c->lv[l].pitch;
  }
  a.w = c->lv[0].iw;
  a.h = c->lv[0].ih;
  a.n_levels = c->p.n_levels;
  a.slots = d_slots;
  a.first_slot = first_slot;
  const int tiles = ((a.w + 63) / 64) * ((a.h + 63) / 64);
  hipLaunchKernelGGL(k_pyramid_all<T>, dim3(tiles, n), dim3(kBlock), 0, c->stream, a);
}

// src/gx/gy point at slot 0 of the level planes; the frames processed are slots[0..n) if given, else first_slot..+n
int launch_scharr(uwt_ctx* c, const uint8_t* src, int16_t* gx, int16_t* gy, int w, int h, int pitch, size_t fs, int n_frames,
                  const int* d_slots = nullptr, int first_slot = 0, hipStream_t on = nullptr) {
  if (n_frames == 0) return UWT_OK;
  hipStream_t stream = on ? on : c->stream;
  if (h >= 8 * kGradVRows) {  // four rows per thread on the tall levels
    const int tiles = ((w + kGradVW - 1) / kGradVW) * ((h + 4 * kGradVRows - 1) / (4 * kGradVRows));
    hipLaunchKernelGGL(k_scharr3_v4<4>, dim3(tiles, n_frames), dim3(kBlock), 0, stream, src, gx, gy, w, h, pitch, fs, d_slots,
                       first_slot);
  } else {
    const int tiles = ((w + kGradVW - 1) / kGradVW) * ((h + kGradVRows - 1) / kGradVRows);
    hipLaunchKernelGGL(k_scharr3_v4<1>, dim3(tiles, n_frames), dim3(kBlock), 0, stream, src, gx, gy, w, h, pitch, fs, d_slots,
                       first_slot);
  }
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

LaunchSel launch_sel(const uwt_ctx* c) {
  LaunchSel sel;
  sel.arith = c->p.arith == UWT_ARITH_LEGACY ? kArithLegacy : kArithOpenCV;
  sel.depth = c->p.has_depth != 0;
  sel.acc64 = c->p.accumulate_f64 != 0;
  sel.compute_only = c->compute_only;
  return sel;
}

int launch_residual(uwt_ctx* c, const ResidualArgs& a, int n_pairs, bool dump) {
  uwt::launch_residual(c->stream, launch_sel(c), a, n_pairs, dump);
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

ResidualArgs residual_args(uwt_ctx* c, int lvl) {
  ResidualArgs a;
  std::memset(&a, 0, sizeof(a));
  a.img = c->img[lvl];
  a.gx = c->gx[lvl];
  a.gy = c->gy[lvl];
  a.depth = c->depth[lvl];
  a.ref_slots = c->d_ref;
  a.tgt_slots = c->d_tgt;
  a.state = c->state;
  a.L = c->lv[lvl];
  a.zf = c->p.z_factor;
  a.af = c->p.angle_factor;
  a.groups_per_block = c->groups_per_block[lvl];
  a.slices = c->slices[lvl];
  a.partials = c->partials;
  a.scale = c->scale;
  a.gain = c->p.gain;
  a.typed_loads = c->tn.typed_loads;
  return a;
}

int prof_begin(uwt_ctx* c, size_t* idx, int lvl = 0, int evaluations = 1) {
  if (c->prof_ev_level.size() < c->ev_used / 2 + 1) { c->prof_ev_level.resize(c->ev_used / 2 + 1); c->prof_ev_evals.resize(c->ev_used / 2 + 1); }
  c->prof_ev_level[c->ev_used / 2] = lvl;
  c->prof_ev_evals[c->ev_used / 2] = evaluations;   // evaluations the bracketed launch stands for (k_coarse: a whole level)
  if (c->ev_used + 2 > c->ev_pool.size()) {
    for (int i = 0; i < 64; i++) {
      hipEvent_t e;
      HIPCHK(c, hipEventCreate(&e));
      c->ev_pool.push_back(e);
    }
  }
  *idx = c->ev_used;
  c->ev_used += 2;
  HIPCHK(c, hipEventRecord(c->ev_pool[*idx], c->stream));
  return UWT_OK;
}

int prof_collect(uwt_ctx* c) {  // after a stream sync
  for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i], c->ev_pool[i + 1]));
    c->prof_ms += ms;
    const int lvl = c->prof_ev_level[i / 2];
    if (lvl >= 0 && lvl < UWT_MAX_LEVELS) { c->prof_level_ms[lvl] += ms; c->prof_level_launches[lvl] += c->prof_ev_evals[i / 2]; }
  }
  c->ev_used = 0;
  return UWT_OK;
}

GeneralArgs general_args(uwt_ctx* c) {
  GeneralArgs ga;
  ga.sampler = c->p.sampler;
  ga.weights = c->p.weights;
  ga.stage = 0;
  ga.gain = c->p.gain;
  ga.hist = c->hist;
  ga.scale = c->scale;
  return ga;
}

// One residual evaluation on the general path (robust weights and/or bilinear sampler): with weights on, one histogram pass
// estimates the scale first (MedianMat / MedianAbsoluteDeviation, src/Tracker.cpp:1571-1619), then the weighted accumulation runs.
int launch_general(uwt_ctx* c, const ResidualArgs& ra, int n_pairs) {
  uwt::launch_general(c->stream, launch_sel(c), ra, n_pairs, c->p.sampler, c->p.weights, c->hist, c->scale);
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

// The per-stage (dump-capable) form of the same evaluation: k_residual_general, one pixel per thread step.  Records use one
// pixel per point and 8192 per block.
int launch_general_dump(uwt_ctx* c, ResidualArgs ra, int n_pairs, int* slices_out = nullptr) {
  GeneralArgs ga = general_args(c);
  // pixels per record: 8192, or more where the level's create-time slicing (whose record count sized `partials`) is
  // coarser than that — very large levels, where init raises the groups per thread to stay under kMaxSlices
  int lvl = 0;
  while (lvl + 1 < c->p.n_levels && c->lv[lvl].n != ra.L.n) lvl++;
  const int per_slice = (ra.L.ng + c->slices[lvl] - 1) / c->slices[lvl];
  ra.groups_per_block = std::max(kBlock * 32, (per_slice + kBlock - 1) / kBlock * kBlock);
  ra.slices = (ra.L.ng + ra.groups_per_block - 1) / ra.groups_per_block;
  if ((size_t)ra.slices * n_pairs > c->partial_records) return fail(c, UWT_ERR_CAPACITY, "per-stage dump needs more partial records than the context holds");
  if (slices_out) *slices_out = ra.slices;
  if (ga.weights)
    HIPCHK(c, hipMemsetAsync(c->hist + (size_t)ra.pair_base * kHistBins, 0, sizeof(unsigned int) * kHistBins * n_pairs, c->stream));
  uwt::launch_general_dump(c->stream, launch_sel(c), ra, ga, n_pairs);
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

UpdateArgs update_args(uwt_ctx* c, int lvl) {
  UpdateArgs ua;
  std::memset(&ua, 0, sizeof(ua));
  ua.partials = c->partials;
  ua.state = c->state;
  ua.slices = c->slices[lvl];
  ua.max_iters = c->p.max_iters;
  ua.early_exit = c->p.early_exit;
  ua.epsilon = c->p.epsilon;
  ua.gain = c->p.gain;
  ua.legacy_solve = c->p.arith == UWT_ARITH_LEGACY ? 1 : 0;
  return ua;
}

// the update of the evaluation `ra` launches, in that launch's tail (tail_update_wave) instead of k_gn_update(ua)
void arm_tail(uwt_ctx* c, ResidualArgs& ra, const UpdateArgs& ua) {
  ra.tail.on = 1;
  ra.tail.tickets = c->d_tickets;
  ra.tail.state = ua.state;
  ra.tail.active = ua.active;
  ra.tail.k = ua.k;
  ra.tail.max_iters = ua.max_iters;
  ra.tail.early_exit = ua.early_exit;
  ra.tail.general = ua.general;
  ra.tail.legacy_solve = ua.legacy_solve;
  ra.tail.epsilon = ua.epsilon;
  ra.tail.gain = ua.gain;
}

int launch_iterate(uwt_ctx* c, const ResidualArgs& a, const IterArgs& ia, int n_pairs) {
  uwt::launch_iterate(c->stream, launch_sel(c), a, ia, n_pairs);
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

// The chained form of Tracker::EstimatePose for a batch (dense points, nearest-neighbour sampler, identity weights, any level
// size): one k_iterate launch per iteration — each block first applies the update of the previous
// evaluation (and the level hand-off when a level begins), then evaluates — and one k_finish at the end; levels x
// iterations + 1 launches instead of 2 x levels x iterations + levels + 2.
int enqueue_estimate_chained(uwt_ctx* c, int n_pairs, float* d_poses, StatsOut* d_stats, const hipEvent_t* level_ready) {
  const uwt_params& p = c->p;
  uint32_t* recs[2] = {c->partials, c->partials2};
  PairState* states[2] = {c->state, c->state2};
  int rp = 0, sp = 0;            // parity of the records / states the NEXT launch writes
  IterArgs ia;
  std::memset(&ia, 0, sizeof(ia));
  ia.u.max_iters = p.max_iters;
  ia.u.early_exit = p.early_exit;
  ia.u.epsilon = p.epsilon;
  ia.u.gain = p.gain;
  ia.u.legacy_solve = p.arith == UWT_ARITH_LEGACY ? 1 : 0;
  ia.scale_t = p.handoff_scale_t;
  ia.initial_error = p.initial_error;
  bool first = true;
  int prev_slices = 0, prev_k = 0, prev_lvl = p.first_level;
  // Speculative launching (early-exit schedules, the synchronous one- or two-pair call): every read-back of "who is still
  // iterating" costs a host round trip of about two evaluations; instead each level gets the evaluations levels usually
  // take plus one, nothing is read back, and the level switch notes on the device whether a pair was cut short — the
  // caller looks once, behind the results, and redoes the alignment the careful way in that (rare) case.
  const bool speculate = c->speculate && p.early_exit;
  const int spec_iters = std::min(p.max_iters, std::max(c->spec_budget, c->tn.first_poll + 1));
  ia.cut_short = speculate ? &c->d_small->cut : nullptr;
  if (speculate) c->h_small->cut = 0;   // host store into page-locked memory, ahead of the launches that may set it
  ia.inline_pairs = c->inline_pairs ? 1 : 0;
  for (int i = 0; i < 4; i++) ia.pair_slots[i] = c->pair_slots[i];
  // The coarsest levels — those a single block evaluates — run to their end in one launch (k_coarse), at least one finer level
  // left for k_iterate.
  int start_lvl = p.first_level;
  bool after_coarse = false;
  {
    int nc = 0;
    while (nc < kCoarseMaxLevels && start_lvl - nc > p.last_level && c->lv[start_lvl - nc].ng <= kCoarseMaxPixels) nc++;
    if (nc > 0 && c->tn.coarse && !c->profiling && !c->compute_only) {
      CoarseArgs ca;
      std::memset(&ca, 0, sizeof(ca));
      for (int i = 0; i < nc; i++) {
        const int lvl = p.first_level - i;
        if (level_ready && lvl != p.first_level) HIPCHK(c, hipStreamWaitEvent(c->stream, level_ready[lvl], 0));  // its gradients
        ca.lv[i] = residual_args(c, lvl);
        ca.lv[i].state = nullptr;
        ca.level_id[i] = lvl;
      }
      ca.n_levels = nc;
      ca.u = ia.u;
      ca.state_out = states[sp ^ 1];     // where the first k_iterate launch looks for its state
      ca.scale_t = ia.scale_t;
      ca.initial_error = ia.initial_error;
      ca.inline_pairs = ia.inline_pairs;
      for (int i = 0; i < 4; i++) ca.pair_slots[i] = ia.pair_slots[i];
      uwt::launch_coarse_chain(c->stream, launch_sel(c), ca, n_pairs);
      HIPCHK(c, hipGetLastError());
      start_lvl = p.first_level - nc;
      after_coarse = true;
    }
  }
  for (int lvl = start_lvl; lvl >= p.last_level; lvl--) {
    if (level_ready && lvl != p.first_level) HIPCHK(c, hipStreamWaitEvent(c->stream, level_ready[lvl], 0));  // its gradients
    ResidualArgs ra = residual_args(c, lvl);
    ra.state = nullptr;
    {  // slicing follows the batch, as in enqueue_estimate
      const int n_groups = c->lv[lvl].ng / c->vecl[lvl];
      int want = ((c->tn.target_blocks ? c->tn.target_blocks : 4096) + n_pairs - 1) / n_pairs;
      want = std::max(1, std::min(want, c->slices[lvl]));
      const int gpt = (n_groups + want * kBlock - 1) / (want * kBlock);
      ra.groups_per_block = gpt * kBlock;
      ra.slices = (n_groups + ra.groups_per_block - 1) / ra.groups_per_block;
    }
    int next_poll = c->tn.first_poll;   // see enqueue_estimate
    int k = 0;
    for (; k < (speculate ? spec_iters : p.max_iters); k++) {
      ia.mode = first ? (after_coarse ? 3 : 0) : (k == 0 ? 2 : 1);
      ia.u.partials = recs[rp ^ 1];
      ia.u.slices = prev_slices;
      ia.u.k = prev_k;
      ia.prev_lvl = prev_lvl;
      ia.state_in = states[sp ^ 1];
      ia.state_out = states[sp];
      ra.partials = recs[rp];
      // the update inside launch k belongs to evaluation k - 1: a poll at launch k sees what the separate-kernel flow saw
      // after its update k - 1
      const bool poll = p.early_exit && !speculate && k == next_poll && k < p.max_iters;
      ia.u.active = poll ? c->d_active : nullptr;
      if (poll) HIPCHK(c, hipMemsetAsync(c->d_active, 0, sizeof(int), c->stream));
      size_t ev = 0;
      if (c->profiling) {
        int st = prof_begin(c, &ev, lvl);
        if (st) return st;
        ra.probe = 1;
        c->prof_slices = ra.slices;
        c->prof_pairs = n_pairs;
        c->prof_records = recs[rp];
      }
      int st = launch_iterate(c, ra, ia, n_pairs);
      if (st) return st;
      if (c->profiling) {
        HIPCHK(c, hipEventRecord(c->ev_pool[ev + 1], c->stream));
        c->prof_launches += 1;
        c->prof_pixels += (long long)n_pairs * c->lv[lvl].gw * c->lv[lvl].gh;
      }
      first = false;
      prev_slices = ra.slices;
      prev_k = k;
      prev_lvl = lvl;
      rp ^= 1;
      sp ^= 1;
      if (poll) {  // reference-mode early exit: stop launching once every pair has left this level
        HIPCHK(c, hipMemcpyAsync(c->h_active, c->d_active, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (*c->h_active == 0) { k++; break; }
        next_poll *= 2;
      }
    }
  }
  // the last evaluation's update, the last level's hand-off, results
  ia.mode = 2;
  ia.u.partials = recs[rp ^ 1];
  ia.u.slices = prev_slices;
  ia.u.k = prev_k;
  ia.u.active = nullptr;
  ia.prev_lvl = prev_lvl;
  ia.state_in = states[sp ^ 1];
  // The final state lands in whichever of the two buffers the last evaluation did not read (that depends on the parity of
  // the launch count); nothing reads either buffer after a chained call — results leave through d_poses / d_stats.
  ia.state_out = ia.state_in == c->state ? c->state2 : c->state;
  uwt::launch_finish(c->stream, ia, n_pairs, d_poses, d_stats);
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

// Tracker::EstimatePose for a batch, enqueued on the context's stream (src/Tracker.cpp:362-597)
// The chained flow pays where an alignment is bound by kernel boundaries and dependent round trips, not by arithmetic: a
// few pairs on their own (the drop-in call).  In a batch every block would repeat its pair's update.  Measured at 640x480
// (round 2, profiles/r03/DESIGN_lab_notes_r01-r03.md): ahead up to 6 pairs in fixed schedules, up to 16 in early-exit schedules
// (half the launches between two read-backs), level from there on (uwt_tuning::chained = 1 / 0 force it on / off).
static bool takes_chained_flow(const uwt_ctx* c, int n_pairs) {
  const int few = c->p.early_exit ? 16 : 6;
  return c->p.accumulate_f64 != 0 && c->p.sampler == 0 && c->p.weights == 0 && (c->tn.chained > 0 || (c->tn.chained < 0 && n_pairs <= few));
}

// The chained flow under robust weights (round 6): a few pairs per call — the drop-in use with the Tukey / Huber weighting on.  An
// evaluation is two launches instead of three (scale pass, weighted sums, update): the update of evaluation k — and the level
// hand-off where a level ends — runs at the head of evaluation k + 1's scale pass (k_hist_iterate), the last one in k_finish.
// 640 x 480, one pair, 4 x 10: 91 launches -> 61.  Same device functions as the launches it replaces: the same poses bit for bit.
static bool takes_chained_general(const uwt_ctx* c, int n_pairs) {
  return c->p.accumulate_f64 != 0 && c->p.weights != 0 && !c->profiling && !c->compute_only &&
         (c->tn.chained > 0 || (c->tn.chained < 0 && n_pairs <= 4));
}

int enqueue_estimate_chained_general(uwt_ctx* c, int n_pairs, float* d_poses, StatsOut* d_stats, const hipEvent_t* level_ready) {
  const uwt_params& p = c->p;
  uint32_t* recs[2] = {c->partials, c->partials2};
  PairState* states[2] = {c->state, c->state2};
  int rp = 0, sp = 0;            // parity of the records / states the NEXT evaluation writes
  IterArgs ia;
  std::memset(&ia, 0, sizeof(ia));
  ia.u.max_iters = p.max_iters;
  ia.u.early_exit = p.early_exit;
  ia.u.epsilon = p.epsilon;
  ia.u.gain = p.gain;
  ia.u.general = 1;
  ia.u.legacy_solve = p.arith == UWT_ARITH_LEGACY ? 1 : 0;
  ia.scale_t = p.handoff_scale_t;
  ia.initial_error = p.initial_error;
  // the per-pair residual histograms (and the ticket word of each) start an alignment all-zero; every scale pass leaves them so
  HIPCHK(c, hipMemsetAsync(c->hist, 0, sizeof(unsigned int) * kHistBins * n_pairs, c->stream));
  bool first = true, after_coarse = false;
  int prev_slices = 0, prev_k = 0, prev_lvl = p.first_level;
  int start_lvl = p.first_level;
  // the coarsest levels a single block evaluates, each to its end in one launch (k_coarse_weighted), one finer level at least
  // left for the chained launches
  while (c->tn.coarse_weighted && p.sampler == 0 && start_lvl > p.last_level && c->lv[start_lvl].ng <= kCoarseMaxPixels) {
    if (level_ready && start_lvl != p.first_level) HIPCHK(c, hipStreamWaitEvent(c->stream, level_ready[start_lvl], 0));  // its gradients
    CoarseArgs ca;
    std::memset(&ca, 0, sizeof(ca));
    ca.lv[0] = residual_args(c, start_lvl);
    ca.lv[0].state = nullptr;
    ca.level_id[0] = start_lvl;
    ca.n_levels = 1;
    ca.u = ia.u;
    ca.state_out = states[sp ^ 1];     // where the first k_hist_iterate launch looks for its state
    ca.scale_t = ia.scale_t;
    ca.initial_error = ia.initial_error;
    ca.resume = after_coarse ? 1 : 0;
    uwt::launch_coarse_level(c->stream, launch_sel(c), ca, n_pairs, p.weights);
    HIPCHK(c, hipGetLastError());
    start_lvl--;
    after_coarse = true;
  }
  for (int lvl = start_lvl; lvl >= p.last_level; lvl--) {
    if (level_ready && lvl != p.first_level) HIPCHK(c, hipStreamWaitEvent(c->stream, level_ready[lvl], 0));  // its gradients
    ResidualArgs ra = residual_args(c, lvl);
    {  // slicing follows the batch, as in enqueue_estimate
      const int n_groups = c->lv[lvl].ng / c->vecl[lvl];
      int want = ((c->tn.target_blocks ? c->tn.target_blocks : 4096) + n_pairs - 1) / n_pairs;
      want = std::max(1, std::min(want, c->slices[lvl]));
      const int gpt = (n_groups + want * kBlock - 1) / (want * kBlock);
      ra.groups_per_block = gpt * kBlock;
      ra.slices = (n_groups + ra.groups_per_block - 1) / ra.groups_per_block;
    }
    int next_poll = c->tn.first_poll;
    for (int k = 0; k < p.max_iters; k++) {
      ia.mode = first ? (after_coarse ? 3 : 0) : (k == 0 ? 2 : 1);
      ia.u.partials = recs[rp ^ 1];
      ia.u.slices = prev_slices;
      ia.u.k = prev_k;
      ia.prev_lvl = prev_lvl;
      ia.state_in = states[sp ^ 1];
      ia.state_out = states[sp];
      ra.partials = recs[rp];
      ra.state = states[sp];           // the weighted launch reads the state the scale pass has just published
      // the update inside launch k belongs to evaluation k - 1 (see enqueue_estimate_chained)
      const bool poll = p.early_exit && k == next_poll && k < p.max_iters;
      ia.u.active = poll ? c->d_active : nullptr;
      if (poll) HIPCHK(c, hipMemsetAsync(c->d_active, 0, sizeof(int), c->stream));
      uwt::launch_hist_iterate(c->stream, launch_sel(c), ra, ia, n_pairs, p.sampler, p.weights, c->hist, c->scale);
      uwt::launch_weighted(c->stream, launch_sel(c), ra, n_pairs, p.sampler, p.weights);
      HIPCHK(c, hipGetLastError());
      first = false;
      prev_slices = ra.slices;
      prev_k = k;
      prev_lvl = lvl;
      rp ^= 1;
      sp ^= 1;
      if (poll) {  // reference-mode early exit: stop launching once every pair has left this level
        HIPCHK(c, hipMemcpyAsync(c->h_active, c->d_active, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (*c->h_active == 0) break;
        next_poll *= 2;
      }
    }
  }
  // the last evaluation's update, the last level's hand-off, results
  ia.mode = 2;
  ia.u.partials = recs[rp ^ 1];
  ia.u.slices = prev_slices;
  ia.u.k = prev_k;
  ia.u.active = nullptr;
  ia.prev_lvl = prev_lvl;
  ia.state_in = states[sp ^ 1];
  ia.state_out = ia.state_in == c->state ? c->state2 : c->state;
  uwt::launch_finish(c->stream, ia, n_pairs, d_poses, d_stats);
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

int enqueue_estimate(uwt_ctx* c, int n_pairs, float* d_poses, StatsOut* d_stats, const hipEvent_t* level_ready = nullptr) {
  const uwt_params& p = c->p;
  if (takes_chained_flow(c, n_pairs)) return enqueue_estimate_chained(c, n_pairs, d_poses, d_stats, level_ready);
  if (takes_chained_general(c, n_pairs)) return enqueue_estimate_chained_general(c, n_pairs, d_poses, d_stats, level_ready);
  const int tb = 128;
  const bool general = p.sampler != 0 || p.weights != 0;
  // Slicing follows the batch: the create-time slicing (kGroupsPerThread) gives a single pair enough blocks to spread
  // over the chip; a batch that fills it alone runs fewer, longer blocks (less reduction overhead per pixel, fewer
  // records to fold), still at least target_blocks per launch.
  // (early-exit schedules stay on one stream: interleaving the two halves' read-backs was built and gave +1.7 %)
  const int parts = (p.early_exit || c->profiling || (long long)n_pairs * c->lv[0].ng < c->tn.split_min_px)
                        ? 1 : std::min(c->tn.split, n_pairs / std::max(1, c->tn.split_min));
  const int target_blocks = c->tn.target_blocks ? c->tn.target_blocks : (parts >= 2 ? 1024 : 4096);
  auto slicing = [&](int lvl, int& groups_per_block, int& slices) {
    const int n_groups = c->lv[lvl].ng / c->vecl[lvl];
    int want = (target_blocks + n_pairs - 1) / n_pairs;
    want = std::max(1, std::min(want, c->slices[lvl]));
    // Fixed schedules (every pair stays to the level's end): a block should also be long enough to carry its fixed costs — the
    // matrix set-up, the two-pass LDS fold, the record, the ticket — i.e. 16 groups per thread, as long as the launch still fills
    // the chip once (1024 resident blocks).  Level 2 of a 1024-pair batch: 1 slice of 19 groups per thread instead of 4 of 5,
    // +4.6 % on that level's launches.  Early-exit schedules keep the finer slicing: pairs leave a level at different
    // evaluations and the blocks of those that stay have to fill the chip (coarser: 437 k -> 314 k alignments/s, measured).
    if (!p.early_exit) {
      const int by_work = std::max(1, n_groups / (kBlock * 16));
      want = std::min(want, std::max(by_work, (1024 + n_pairs - 1) / n_pairs));
    }
    const int gpt = (n_groups + want * kBlock - 1) / (want * kBlock);
    groups_per_block = gpt * kBlock;
    slices = (n_groups + groups_per_block - 1) / groups_per_block;
  };
  // the accumulation kernels stream a level's reference planes past the caches (ResidualArgs::stream_planes) when the batch's
  // planes of that level — u8 + 2 x i16 [+ u16] per reference pixel, the target's u8 — exceed what the 256 MB memory-side cache
  // holds across an evaluation; a smaller batch finds them there again at the next evaluation (uwt_tuning::stream_bytes)
  auto streams = [&](int lvl) -> int {
    return (long long)n_pairs * c->lv[lvl].n * (p.has_depth ? 8 : 6) > c->tn.stream_bytes ? 1 : 0;
  };
  int smax = 1;
  for (int lvl = p.first_level; lvl >= p.last_level; lvl--) {
    int gpb, sl;
    slicing(lvl, gpb, sl);
    smax = std::max(smax, sl);
  }
  // robust weights: the per-pair residual histograms (and the ticket word of each) start an alignment all-zero; every scale
  // pass leaves them so (k_resid_hist_v)
  if (general && p.weights)
    HIPCHK(c, hipMemsetAsync(c->hist, 0, sizeof(unsigned int) * kHistBins * n_pairs, c->stream));
  // The coarsest levels of a batch in ONE launch each (round 3): one block per pair runs the level to its end — evaluation,
  // update, exit test and hand-off on the device, the record in LDS (four blocks per CU) — instead of a residual and an update
  // launch per evaluation whose blocks, a few pixel groups long, run 25-50 % below the level-0 rate.  f64 sums, levels of up to
  // coarse_batch_px pixels (rows of whole groups of four or not: the kernels' VEC switch); identity weights: k_coarse_w4;
  // robust weights over the nearest sampler: k_coarse_weighted (histogram, scale, weighted sums and update of a whole level in
  // one block); the bilinear sampler stays on the launches.  Square pixels with unit factors or not: the kernels' PLAIN switch.
  bool coarse_lvl[UWT_MAX_LEVELS] = {};
  if (p.accumulate_f64 != 0 && (!general || (p.sampler == 0 && p.weights != 0 && c->tn.coarse_weighted)) && c->tn.coarse_batch_px > 0 &&
      !c->compute_only && !(c->profiling && p.early_exit))
    for (int lvl = p.first_level; lvl >= p.last_level; lvl--)
      coarse_lvl[lvl] = c->lv[lvl].ng <= c->tn.coarse_batch_px;
  // one coarse level of pairs [base, base + cnt) on stream s; resume: the pairs' states exist (a level ran before this one)
  auto run_coarse = [&](int base, int cnt, hipStream_t s, int lvl, bool resume) -> int {
    CoarseArgs ca;
    std::memset(&ca, 0, sizeof(ca));
    ca.lv[0] = residual_args(c, lvl);
    ca.lv[0].state = nullptr;
    ca.level_id[0] = lvl;
    ca.n_levels = 1;
    UpdateArgs u = update_args(c, lvl);
    u.pair_base = base;
    ca.u = u;
    ca.state_out = c->state;
    ca.scale_t = p.handoff_scale_t;
    ca.initial_error = p.initial_error;
    ca.resume = resume ? 1 : 0;
    size_t ev = 0;
    if (c->profiling) {
      int st = prof_begin(c, &ev, lvl, p.max_iters);
      if (st) return st;
    }
    uwt::launch_coarse_level(s, launch_sel(c), ca, cnt, general ? p.weights : 0);
    HIPCHK(c, hipGetLastError());
    if (c->profiling) {   // (fixed schedules only: the level runs max_iters evaluations; the launch stands for that many)
      HIPCHK(c, hipEventRecord(c->ev_pool[ev + 1], s));
      c->prof_launches += p.max_iters;
      c->prof_pixels += (long long)cnt * c->lv[lvl].gw * c->lv[lvl].gh * p.max_iters;
    }
    return UWT_OK;
  };
  // the schedule for pairs [base, base + cnt) of a batch of n_pairs, on c->stream
  auto run = [&](int base, int cnt) -> int {
    if (!coarse_lvl[p.first_level]) {
      hipLaunchKernelGGL(k_init_state, dim3((cnt + tb - 1) / tb), dim3(tb), 0, c->stream, c->state + base, cnt, p.initial_error);
      HIPCHK(c, hipGetLastError());
    }
    for (int lvl = p.first_level; lvl >= p.last_level; lvl--) {
      if (level_ready && lvl != p.first_level) HIPCHK(c, hipStreamWaitEvent(c->stream, level_ready[lvl], 0));  // its gradients
      if (coarse_lvl[lvl]) {
        int stc = run_coarse(base, cnt, c->stream, lvl, lvl != p.first_level);
        if (stc) return stc;
        continue;
      }
      ResidualArgs ra = residual_args(c, lvl);
      UpdateArgs ua = update_args(c, lvl);
      ra.pair_base = base;
      ua.pair_base = base;
      ra.stream_planes = streams(lvl);
      slicing(lvl, ra.groups_per_block, ra.slices);
      ua.slices = ra.slices;
      // A pair's records sit at (pair * slices + slice): the place depends on the level's slice count, and the parts of a
      // split batch are at different levels at times.  Part `base` is shifted so that its records start at base * smax
      // whatever the level — behind everything the parts before it can touch, inside the buffer (slices <= smax).
      const size_t shift = (size_t)base * (size_t)(smax - ra.slices) * kRecWords;
      ra.partials = c->partials + shift;
      ua.partials = c->partials + shift;
      // Early exit: the host reads back how many pairs are still iterating after the update of evaluation first_poll - 1,
      // then after twice as many, ...  With the reference's constants a level ends at its third evaluation as a rule
      // (error rises or stalls, src/Tracker.cpp:508), so the first look comes after three (uwt_tuning::first_poll).
      // The read-back is taken one evaluation late (round 3): the count of evaluation k is copied to page-locked memory
      // behind its update, evaluation k + 1 is enqueued, and only then does the host wait for the copy — the GPU works on
      // k + 1 meanwhile instead of idling for the host's round trip (~25 us per look).  When the count says "nobody left",
      // evaluation k + 1 has been enqueued for nothing: its blocks see level_done and return at once (a few us).
      int next_poll = c->tn.first_poll;
      int pending = -1;   // slot of the look not yet taken
      for (int k = 0; k < p.max_iters; k++) {
        size_t ev = 0;
        if (c->profiling) {
          int st = prof_begin(c, &ev, lvl);
          if (st) return st;
          ra.probe = 1;
          c->prof_slices = ra.slices;
          c->prof_pairs = cnt;
          c->prof_records = c->partials;
        }
        if (general) ua.general = 1;
        ua.k = k;
        const bool poll = p.early_exit && (k + 1 == next_poll) && (k + 1 < p.max_iters);
        const int slot = c->poll_seq & 1;
        ua.active = poll ? c->d_active + slot : nullptr;
        if (poll) HIPCHK(c, hipMemsetAsync(c->d_active + slot, 0, sizeof(int), c->stream));
        const bool tail = c->tn.tail_update >= 2 && !c->compute_only;   // (one stream: only when forced, see tail_update)
        if (tail) arm_tail(c, ra, ua);
        int st = general ? launch_general(c, ra, cnt) : launch_residual(c, ra, cnt, false);
        if (st) return st;
        if (c->profiling) {
          HIPCHK(c, hipEventRecord(c->ev_pool[ev + 1], c->stream));
          c->prof_launches += 1;
          c->prof_pixels += (long long)cnt * c->lv[lvl].gw * c->lv[lvl].gh;
        }
        if (!tail) {
          hipLaunchKernelGGL(k_gn_update, dim3(cnt), dim3(kUpdateBlock), 0, c->stream, ua);
          HIPCHK(c, hipGetLastError());
        }
        if (pending >= 0) {   // the look at the evaluation before this one, taken while this one runs
          HIPCHK(c, hipEventSynchronize(c->ev_poll[pending]));
          const bool nobody_left = c->h_active[pending] == 0;
          pending = -1;
          if (nobody_left) break;   // reference-mode early exit: every pair has left this level
        }
        if (poll) {
          HIPCHK(c, hipMemcpyAsync(c->h_active + slot, c->d_active + slot, sizeof(int), hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, hipEventRecord(c->ev_poll[slot], c->stream));
          pending = slot;
          c->poll_seq++;
          next_poll *= 2;
        }
      }
      hipLaunchKernelGGL(k_level_end, dim3((cnt + tb - 1) / tb), dim3(tb), 0, c->stream, c->state + base, cnt, lvl,
                         p.handoff_scale_t, p.initial_error);
      HIPCHK(c, hipGetLastError());
    }
    hipLaunchKernelGGL(k_write_out, dim3((cnt + tb - 1) / tb), dim3(tb), 0, c->stream, c->state + base, cnt, d_poses + 7 * (size_t)base,
                       d_stats ? d_stats + base : nullptr);
    HIPCHK(c, hipGetLastError());
    return UWT_OK;
  };
  // A batch is cut into two parts that run the same schedule on streams of their own (fixed schedules; 16 pairs and the
  // pixels of 32 640x480 pairs or more): the launches of one part run in the gaps of the other's — the tail of a residual
  // launch, the update launch, the kernel boundaries: +2..4 % at 256..1024 pairs, +7..9 % at 32..64 in a pipeline of calls;
  // three parts gain nothing more, four lose (measured).  Results do not depend on it (a pair's blocks, records and state
  // are its own; the slicing is the whole batch's).
  if (parts < 2) {
    if (c->tn.tail_update >= 2) HIPCHK(c, hipMemsetAsync(c->d_tickets, 0, sizeof(unsigned int) * (size_t)n_pairs, c->stream));
    return run(0, n_pairs);
  }
  // (fixed schedule, not profiled: no read-backs, no events around launches).  The parts' launches are enqueued in turns,
  // iteration by iteration, so that both streams have work from the start.
  struct Part { int base, cnt; hipStream_t s; ResidualArgs ra; UpdateArgs ua; };
  Part pt[uwt_ctx::kMaxParts];
  hipStream_t main_stream = c->stream;
  // every early return below (a failed launch or event call) leaves part streams forked and not joined: drain them before
  // the error reaches the caller, so that uwt_sync / uwt_destroy on the main stream really mean "nothing is running"
  struct JoinGuard {
    uwt_ctx* c; int parts; hipStream_t main; bool joined = false;
    ~JoinGuard() {
      if (joined) return;
      c->stream = main;
      for (int i = 1; i < parts; i++) (void)hipStreamSynchronize(c->part_stream[i]);
    }
  } guard{c, parts, main_stream};
  // (the pairs' ticket counters are zero between launches by construction; a call that an error cut short may have left some)
  if (c->tn.tail_update >= 1) HIPCHK(c, hipMemsetAsync(c->d_tickets, 0, sizeof(unsigned int) * (size_t)n_pairs, main_stream));
  HIPCHK(c, hipEventRecord(c->ev_fork, main_stream));
  for (int i = 0; i < parts; i++) {
    Part& q = pt[i];
    q.base = (int)((long long)n_pairs * i / parts);
    q.cnt = (int)((long long)n_pairs * (i + 1) / parts) - q.base;
    q.s = i ? c->part_stream[i] : main_stream;
    if (i) HIPCHK(c, hipStreamWaitEvent(q.s, c->ev_fork, 0));
    if (!coarse_lvl[p.first_level])
      hipLaunchKernelGGL(k_init_state, dim3((q.cnt + tb - 1) / tb), dim3(tb), 0, q.s, c->state + q.base, q.cnt, p.initial_error);
  }
  HIPCHK(c, hipGetLastError());
  int st = UWT_OK;
  for (int lvl = p.first_level; lvl >= p.last_level && st == UWT_OK; lvl--) {
    if (coarse_lvl[lvl]) {
      for (int i = 0; i < parts; i++) {
        if (level_ready && lvl != p.first_level) HIPCHK(c, hipStreamWaitEvent(pt[i].s, level_ready[lvl], 0));  // its gradients
        int stc = run_coarse(pt[i].base, pt[i].cnt, pt[i].s, lvl, lvl != p.first_level);
        if (stc) return stc;
      }
      continue;
    }
    for (int i = 0; i < parts; i++) {
      Part& q = pt[i];
      if (level_ready && lvl != p.first_level) HIPCHK(c, hipStreamWaitEvent(q.s, level_ready[lvl], 0));  // its gradients
      q.ra = residual_args(c, lvl);
      q.ua = update_args(c, lvl);
      q.ra.pair_base = q.ua.pair_base = q.base;
      q.ra.stream_planes = streams(lvl);
      slicing(lvl, q.ra.groups_per_block, q.ra.slices);
      q.ua.slices = q.ra.slices;
      // A pair's records sit at (pair * slices + slice): the place depends on the level's slice count, and the parts are at
      // different levels at times.  A part's records are shifted so that they start at base * smax whatever the level —
      // behind everything the parts before it can touch, inside the buffer (slices <= smax).
      const size_t shift = (size_t)q.base * (size_t)(smax - q.ra.slices) * kRecWords;
      q.ra.partials = c->partials + shift;
      q.ua.partials = c->partials + shift;
      if (general) q.ua.general = 1;
    }
    for (int k = 0; k < p.max_iters && st == UWT_OK; k++)
      for (int i = 0; i < parts && st == UWT_OK; i++) {
        Part& q = pt[i];
        q.ua.k = k;
        q.ua.active = nullptr;
        const bool tail = c->tn.tail_update >= 1 && !c->compute_only;   // the update in the tail of the evaluation's launch
        if (tail) arm_tail(c, q.ra, q.ua);
        c->stream = q.s;      // every launch helper enqueues on c->stream
        st = general ? launch_general(c, q.ra, q.cnt) : launch_residual(c, q.ra, q.cnt, false);
        c->stream = main_stream;
        if (st) break;
        if (!tail) hipLaunchKernelGGL(k_gn_update, dim3(q.cnt), dim3(kUpdateBlock), 0, q.s, q.ua);
      }
    if (st) return st;
    for (int i = 0; i < parts; i++)
      hipLaunchKernelGGL(k_level_end, dim3((pt[i].cnt + tb - 1) / tb), dim3(tb), 0, pt[i].s, c->state + pt[i].base, pt[i].cnt, lvl,
                         p.handoff_scale_t, p.initial_error);
    HIPCHK(c, hipGetLastError());
  }
  for (int i = 0; i < parts; i++) {
    hipLaunchKernelGGL(k_write_out, dim3((pt[i].cnt + tb - 1) / tb), dim3(tb), 0, pt[i].s, c->state + pt[i].base, pt[i].cnt,
                       d_poses + 7 * (size_t)pt[i].base, d_stats ? d_stats + pt[i].base : nullptr);
    if (i) {
      HIPCHK(c, hipEventRecord(c->ev_join[i], pt[i].s));
      HIPCHK(c, hipStreamWaitEvent(main_stream, c->ev_join[i], 0));
    }
  }
  HIPCHK(c, hipGetLastError());
  guard.joined = true;
  return UWT_OK;
}

// ---- slot-range dependencies between the context stream and the copy stream --------------------------------------
int dep_note(uwt_ctx* c, uwt_ctx::SlotDep* ring, int& next, bool& dropped, hipStream_t on, int first, int n) {
  uwt_ctx::SlotDep& d = ring[next];
  if (d.used) dropped = true;
  d.first = first; d.n = n; d.used = true;
  HIPCHK(c, hipEventRecord(d.ev, on));
  next = (next + 1) % uwt_ctx::kDeps;
  return UWT_OK;
}
int dep_wait(uwt_ctx* c, const uwt_ctx::SlotDep* ring, int next, bool dropped, hipStream_t waiter, int first, int n) {
  for (int i = 0; i < uwt_ctx::kDeps; i++) {
    const uwt_ctx::SlotDep& d = ring[i];
    const bool oldest = dropped && i == next;   // ring[next] is the oldest survivor once the ring has wrapped
    if (d.used && (oldest || (d.first < first + n && first < d.first + d.n))) HIPCHK(c, hipStreamWaitEvent(waiter, d.ev, 0));
  }
  return UWT_OK;
}
// a compute call on slots [first, first + n): ordered behind the uploads into them ...
int compute_begin(uwt_ctx* c, int first, int n) { return dep_wait(c, c->fresh, c->fresh_next, c->fresh_dropped, c->stream, first, n); }
// ... and remembered, so that a later upload into them waits for it
int compute_end(uwt_ctx* c, int first, int n) {
  c->busy_seq[c->busy_next] = ++c->ticket_seq;
  return dep_note(c, c->busy, c->busy_next, c->busy_dropped, c->stream, first, n);
}

// The caller's lists are copied before this returns (they may be temporaries): into a pinned staging buffer, then
// asynchronously to the device.  Unchanged lists (the steady state of a resident batch) are not re-sent.
int upload_pairs(uwt_ctx* c, int n_pairs, const int32_t* ref_slots, const int32_t* tgt_slots) {
  if (!ref_slots || !tgt_slots || n_pairs < 1) return fail(c, UWT_ERR_INVALID_ARG, "null pair lists or n_pairs < 1");
  if (n_pairs > c->p.max_pairs) return fail(c, UWT_ERR_CAPACITY, "n_pairs exceeds max_pairs");
  for (int i = 0; i < n_pairs; i++)
    if (ref_slots[i] < 0 || ref_slots[i] >= c->p.max_frames || tgt_slots[i] < 0 || tgt_slots[i] >= c->p.max_frames)
      return fail(c, UWT_ERR_INVALID_ARG, "pair slot out of range");
  const size_t block = 2 * (size_t)c->p.max_pairs;
  {
    const int* h_ref = c->h_pairs + c->pair_stage * block;
    const int* h_tgt = h_ref + c->p.max_pairs;
    if (n_pairs == c->n_pairs_cached && !std::memcmp(h_ref, ref_slots, sizeof(int) * n_pairs) &&
        !std::memcmp(h_tgt, tgt_slots, sizeof(int) * n_pairs))
      return UWT_OK;
  }
  const int stage = (c->pair_stage + 1) % uwt_ctx::kPairStages;
  HIPCHK(c, hipEventSynchronize(c->ev_pairs[stage]));  // the copy that last read this block (kPairStages lists ago)
  int* h_ref = c->h_pairs + stage * block;
  int* h_tgt = h_ref + c->p.max_pairs;
  std::memcpy(h_ref, ref_slots, sizeof(int) * n_pairs);
  std::memcpy(h_tgt, tgt_slots, sizeof(int) * n_pairs);
  c->pair_stage = stage;
  c->n_pairs_cached = n_pairs;
  HIPCHK(c, hipMemcpyAsync(c->d_ref, h_ref, sizeof(int) * n_pairs, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_tgt, h_tgt, sizeof(int) * n_pairs, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->ev_pairs[stage], c->stream));
  return UWT_OK;
}

int run_se3_op(uwt_ctx* c, int op, const float* a, int na, const float* b, int nb, float* out, int nout, int* flag) {
  if (!c) return UWT_ERR_INVALID_ARG;
  int st = ensure_scratch(c, 4096);
  if (st) return st;
  float* d = (float*)c->scratch;  // [0,64) a, [64,128) b, [128,256) out, [256] flag
  HIPCHK(c, hipMemcpyAsync(d, a, sizeof(float) * na, hipMemcpyHostToDevice, c->stream));
  if (b) HIPCHK(c, hipMemcpyAsync(d + 64, b, sizeof(float) * nb, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_se3_ops, dim3(1), dim3(1), 0, c->stream, op, d, d + 64, d + 128, (int*)(d + 256));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(out, d + 128, sizeof(float) * nout, hipMemcpyDeviceToHost, c->stream));
  int f = 1;
  HIPCHK(c, hipMemcpyAsync(&f, d + 256, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (flag) *flag = f;
  return UWT_OK;
}

}  // namespace

extern "C" {

int uwt_abi_version(void) { return UWT_ABI_VERSION; }

#ifndef UWT_SOURCE_SHA256
#define UWT_SOURCE_SHA256 "unknown (built outside csrc/Makefile)"
#endif
const char* uwt_source_id(void) { return UWT_SOURCE_SHA256; }

const char* uwt_status_string(int status) {
  switch (status) {
    case UWT_OK: return "ok";
    case UWT_ERR_INVALID_ARG: return "invalid argument";
    case UWT_ERR_NO_VALID_POINTS: return "no valid points";
    case UWT_ERR_HIP: return "HIP error";
    case UWT_ERR_NO_DEVICE: return "no gfx950 device";
    case UWT_ERR_CAPACITY: return "capacity exceeded";
    case UWT_ERR_PAIR_FAILED: return "at least one pair failed";
    default: return "unknown status";
  }
}

const char* uwt_last_error(const uwt_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int uwt_default_params(uwt_params* p, int32_t width, int32_t height, float fx, float fy, float cx, float cy) {
  if (!p) return UWT_ERR_INVALID_ARG;
  std::memset(p, 0, sizeof(*p));
  p->width = width; p->height = height;
  p->fx = fx; p->fy = fy; p->cx = cx; p->cy = cy;
  p->n_levels = 5;          // src/Options.cpp:26
  p->first_level = 4;       // src/Tracker.cpp:368
  p->last_level = 1;        // :369
  p->max_iters = 50;        // :366
  p->epsilon = 0.001f;      // :364
  p->gain = 50.0f;          // :559
  p->z_factor = 1.0f;       // :371
  p->angle_factor = 1.0f;   // :372
  p->depth_scale = 0.0002f; // :1261
  p->initial_error = 50000.0f;  // :393
  p->early_exit = 1;
  p->has_depth = 0;
  p->handoff_scale_t = 0;
  p->accumulate_f64 = 1;
  p->sampler = 0;           // nearest neighbour, round() (src/Tracker.cpp:472)
  p->weights = 0;           // IdentityWeights (src/Tracker.cpp:495)
  p->max_frames = 2;
  p->max_pairs = 1;
  p->device = 0;
  p->arith = UWT_ARITH_OPENCV;
  return UWT_OK;
}

int uwt_create(const uwt_params* p, uwt_ctx** out) {
  if (!p || !out) return UWT_ERR_INVALID_ARG;
  *out = nullptr;
  if (p->n_levels < 1 || p->n_levels > UWT_MAX_LEVELS || p->width < 1 || p->height < 1) return UWT_ERR_INVALID_ARG;
  // any size whose coarsest level still has a point grid (w_[lvl] = width >> lvl >= 1, src/Tracker.cpp:312-313)
  if ((p->width >> (p->n_levels - 1)) < 1 || (p->height >> (p->n_levels - 1)) < 1) return UWT_ERR_INVALID_ARG;
  if (p->first_level >= p->n_levels || p->last_level < 0 || p->last_level > p->first_level) return UWT_ERR_INVALID_ARG;
  if (p->max_iters < 1 || p->max_frames < 1 || p->max_pairs < 1) return UWT_ERR_INVALID_ARG;
  {   // the kernels divide a level's linear index by its row pitch with one multiply (LevelK::magic): index * pitch < 2^32
    const uint64_t pitch0 = ((uint64_t)p->width + 3) & ~3ull;
    if (pitch0 * p->height * pitch0 >= 0x100000000ull) return UWT_ERR_INVALID_ARG;
  }
  if (p->sampler < 0 || p->sampler > 1 || p->weights < 0 || p->weights > 2) return UWT_ERR_INVALID_ARG;
  if (p->sampler == 1 && p->weights == 1) return UWT_ERR_INVALID_ARG;  // the reference's Tukey medians are defined on integer residuals
  if (p->arith != UWT_ARITH_OPENCV && p->arith != UWT_ARITH_LEGACY) return UWT_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || p->device < 0 || p->device >= ndev) return UWT_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, p->device) != hipSuccess) return UWT_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return UWT_ERR_NO_DEVICE;  // code objects are gfx950 only

  uwt_ctx* c = new (std::nothrow) uwt_ctx();
  if (!c) return UWT_ERR_CAPACITY;
  c->p = *p;
  init_levels(c);
  c->tn = default_tuning();
  size_t max_slices = 1;
  for (int l = 0; l < p->n_levels; l++) {
    c->vecl[l] = 4;
    const int n_groups = c->lv[l].ng / c->vecl[l];
    const int gpt = std::max(kGroupsPerThread, (n_groups + kMaxSlices * kBlock - 1) / (kMaxSlices * kBlock));
    c->groups_per_block[l] = kBlock * gpt;
    c->slices[l] = (n_groups + c->groups_per_block[l] - 1) / c->groups_per_block[l];
    if ((size_t)c->slices[l] > max_slices) max_slices = c->slices[l];
  }
  c->partial_records = max_slices * (size_t)p->max_pairs;

#define CREATE_CHK(expr)                                                             \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "uwt_create: %s: %s\n", #expr, hipGetErrorString(e_));    \
      uwt_destroy(c);                                                                \
      return UWT_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)
  CREATE_CHK(hipSetDevice(p->device));
  CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  CREATE_CHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  for (int i = 1; i < uwt_ctx::kMaxParts; i++) {
    CREATE_CHK(hipStreamCreateWithFlags(&c->part_stream[i], hipStreamNonBlocking));
    CREATE_CHK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
  }
  CREATE_CHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  CREATE_CHK(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
  for (int i = 0; i < uwt_ctx::kDeps; i++) {
    CREATE_CHK(hipEventCreateWithFlags(&c->busy[i].ev, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&c->fresh[i].ev, hipEventDisableTiming));
  }
  for (int i = 0; i < uwt_ctx::kPairStages; i++) CREATE_CHK(hipEventCreateWithFlags(&c->ev_pairs[i], hipEventDisableTiming));
  CREATE_CHK(hipEventCreateWithFlags(&c->ev_pyramids, hipEventDisableTiming));
  CREATE_CHK(hipEventCreateWithFlags(&c->ev_side_done, hipEventDisableTiming));
  for (int l = 0; l < UWT_MAX_LEVELS; l++) CREATE_CHK(hipEventCreateWithFlags(&c->ev_level[l], hipEventDisableTiming));
  for (int l = 0; l < p->n_levels; l++) {
    const size_t n = (size_t)c->lv[l].n * p->max_frames;
    CREATE_CHK(hipMalloc((void**)&c->img[l], n + 4096));
    CREATE_CHK(hipMalloc((void**)&c->gx[l], n * 2));
    CREATE_CHK(hipMalloc((void**)&c->gy[l], n * 2));
    if (p->has_depth) CREATE_CHK(hipMalloc((void**)&c->depth[l], n * 2));
    if (c->lv[l].pitch != c->lv[l].iw) {   // the pad columns of pitched rows: never part of a result, defined all the same
      CREATE_CHK(hipMemset(c->img[l], 0, n + 4096));
      CREATE_CHK(hipMemset(c->gx[l], 0, n * 2));
      CREATE_CHK(hipMemset(c->gy[l], 0, n * 2));
      if (p->has_depth) CREATE_CHK(hipMemset(c->depth[l], 0, n * 2));
    }
  }
  CREATE_CHK(hipMalloc((void**)&c->state, sizeof(PairState) * p->max_pairs));
  CREATE_CHK(hipMalloc((void**)&c->d_ref, sizeof(int) * p->max_pairs));
  CREATE_CHK(hipMalloc((void**)&c->d_tgt, sizeof(int) * p->max_pairs));
  CREATE_CHK(hipMalloc((void**)&c->partials, c->partial_records * kRecWords * sizeof(uint32_t)));
  CREATE_CHK(hipMalloc((void**)&c->partials2, c->partial_records * kRecWords * sizeof(uint32_t)));
  CREATE_CHK(hipMalloc((void**)&c->state2, sizeof(PairState) * p->max_pairs));
  CREATE_CHK(hipMalloc((void**)&c->d_poses, sizeof(float) * 7 * p->max_pairs));
  CREATE_CHK(hipMalloc((void**)&c->d_stats, sizeof(StatsOut) * p->max_pairs));
  CREATE_CHK(hipMalloc((void**)&c->d_active, 2 * sizeof(int)));
  CREATE_CHK(hipMalloc((void**)&c->d_tickets, sizeof(unsigned int) * (size_t)c->p.max_pairs));
  CREATE_CHK(hipMemset(c->d_tickets, 0, sizeof(unsigned int) * (size_t)c->p.max_pairs));
  for (int i = 0; i < 2; i++) CREATE_CHK(hipEventCreateWithFlags(&c->ev_poll[i], hipEventDisableTiming));
  CREATE_CHK(hipHostMalloc((void**)&c->h_small, sizeof(uwt_ctx::SmallResults)));
  CREATE_CHK(hipHostGetDevicePointer((void**)&c->d_small, c->h_small, 0));
  if (p->sampler || p->weights) {
    CREATE_CHK(hipMalloc((void**)&c->hist, sizeof(unsigned int) * kHistBins * p->max_pairs));
    CREATE_CHK(hipMalloc((void**)&c->scale, sizeof(PairScale) * p->max_pairs));
    CREATE_CHK(hipMemset(c->scale, 0, sizeof(PairScale) * p->max_pairs));
  }
  CREATE_CHK(hipHostMalloc((void**)&c->h_active, 2 * sizeof(int)));
  CREATE_CHK(hipHostMalloc((void**)&c->h_pairs, sizeof(int) * 2 * p->max_pairs * uwt_ctx::kPairStages));
  std::memset(c->h_pairs, 0xff, sizeof(int) * 2 * p->max_pairs * uwt_ctx::kPairStages);
  // hipMemset returns before the device has run it, and it runs on the NULL stream, which the context's streams (non-blocking) do not
  // wait for: without this the zeros of the pitched planes could land on top of the first upload's rows (seen as rare parity events
  // once uploads into pitched rows became a kernel: profiles/r06/EXPERIMENTS.md 14)
  CREATE_CHK(hipStreamSynchronize(nullptr));
#undef CREATE_CHK
  *out = c;
  return UWT_OK;
}

int uwt_destroy(uwt_ctx* c) {
  if (!c) return UWT_ERR_INVALID_ARG;
  (void)hipSetDevice(c->p.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->side) (void)hipStreamSynchronize(c->side);  // an aborted uwt_track_batch_async may have left work there
  if (c->copy) (void)hipStreamSynchronize(c->copy);
  for (int i = 1; i < uwt_ctx::kMaxParts; i++)
    if (c->part_stream[i]) (void)hipStreamSynchronize(c->part_stream[i]);
  for (int l = 0; l < UWT_MAX_LEVELS; l++) {
    if (c->img[l]) (void)hipFree(c->img[l]);
    if (c->depth[l]) (void)hipFree(c->depth[l]);
    if (c->gx[l]) (void)hipFree(c->gx[l]);
    if (c->gy[l]) (void)hipFree(c->gy[l]);
  }
  if (c->state) (void)hipFree(c->state);
  if (c->d_ref) (void)hipFree(c->d_ref);
  if (c->d_tgt) (void)hipFree(c->d_tgt);
  if (c->partials) (void)hipFree(c->partials);
  if (c->partials2) (void)hipFree(c->partials2);
  if (c->state2) (void)hipFree(c->state2);
  if (c->d_poses) (void)hipFree(c->d_poses);
  if (c->d_stats) (void)hipFree(c->d_stats);
  if (c->d_active) (void)hipFree(c->d_active);
  if (c->d_tickets) (void)hipFree(c->d_tickets);
  for (int i = 0; i < 2; i++)
    if (c->ev_poll[i]) (void)hipEventDestroy(c->ev_poll[i]);
  if (c->h_small) (void)hipHostFree(c->h_small);
  if (c->hist) (void)hipFree(c->hist);
  if (c->scale) (void)hipFree(c->scale);
  if (c->h_active) (void)hipHostFree(c->h_active);
  if (c->h_pairs) (void)hipHostFree(c->h_pairs);
  if (c->scratch) (void)hipFree(c->scratch);
  for (void* p : c->stage)
    if (p) (void)hipFree(p);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->side) (void)hipStreamDestroy(c->side);
  for (int i = 1; i < uwt_ctx::kMaxParts; i++) {
    if (c->part_stream[i]) (void)hipStreamDestroy(c->part_stream[i]);
    if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->copy) (void)hipStreamDestroy(c->copy);
  for (int i = 0; i < uwt_ctx::kDeps; i++) {
    if (c->busy[i].ev) (void)hipEventDestroy(c->busy[i].ev);
    if (c->fresh[i].ev) (void)hipEventDestroy(c->fresh[i].ev);
  }
  for (int i = 0; i < uwt_ctx::kPairStages; i++)
    if (c->ev_pairs[i]) (void)hipEventDestroy(c->ev_pairs[i]);
  if (c->ev_pyramids) (void)hipEventDestroy(c->ev_pyramids);
  if (c->ev_side_done) (void)hipEventDestroy(c->ev_side_done);
  for (int l = 0; l < UWT_MAX_LEVELS; l++)
    if (c->ev_level[l]) (void)hipEventDestroy(c->ev_level[l]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return UWT_OK;
}

int uwt_get_params(const uwt_ctx* c, uwt_params* out) {
  if (!c || !out) return UWT_ERR_INVALID_ARG;
  *out = c->p;
  return UWT_OK;
}

int uwt_get_tuning(const uwt_ctx* c, uwt_tuning* out) {
  if (!c || !out) return UWT_ERR_INVALID_ARG;
  *out = c->tn;
  return UWT_OK;
}

int uwt_set_tuning(uwt_ctx* c, const uwt_tuning* t) {
  if (!c || !t) return UWT_ERR_INVALID_ARG;
  if (t->split < 1 || t->split > uwt_ctx::kMaxParts || t->split_min < 1 || t->split_min_px < 1 || t->stream_bytes < 0 ||
      t->tail_update < 0 || t->tail_update > 2 || t->target_blocks < 0 || t->target_blocks > (1 << 20) || t->coarse_batch_px < 0 ||
      t->coarse_batch_px > (1 << 24) || t->first_poll < 1 || t->first_poll > (1 << 20) || t->chained < -1 || t->chained > 1)
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_set_tuning: value out of range");   // (first_poll is doubled and incremented by the schedulers: bounded well inside int)
  (void)hipSetDevice(c->p.device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->tn = *t;
  c->spec_budget = c->spec_calm = 0;
  for (int32_t* b : {&c->tn.coarse, &c->tn.coarse_weighted, &c->tn.overlap_gradients, &c->tn.speculation, &c->tn.fused_stages,
                     &c->tn.pyramid_batch, &c->tn.typed_loads})
    *b = *b != 0;
  std::memset(c->tn.reserved, 0, sizeof(c->tn.reserved));
  return UWT_OK;
}

int uwt_update_params(uwt_ctx* c, const uwt_params* p) {
  if (!c || !p) return UWT_ERR_INVALID_ARG;
  const uwt_params& o = c->p;
  if (p->width != o.width || p->height != o.height || p->fx != o.fx || p->fy != o.fy || p->cx != o.cx || p->cy != o.cy ||
      p->n_levels != o.n_levels || p->has_depth != o.has_depth || p->max_frames != o.max_frames ||
      p->max_pairs != o.max_pairs || p->device != o.device || p->depth_scale != o.depth_scale)
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_update_params: geometry / capacity fields differ from the context's");
  if (p->first_level >= p->n_levels || p->last_level < 0 || p->last_level > p->first_level || p->max_iters < 1 ||
      p->sampler < 0 || p->sampler > 1 || p->weights < 0 || p->weights > 2 || (p->sampler == 1 && p->weights == 1) ||
      (p->arith != UWT_ARITH_OPENCV && p->arith != UWT_ARITH_LEGACY))
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_update_params: bad solver constants");
  (void)hipSetDevice(o.device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if ((p->sampler || p->weights) && !c->hist) {
    HIPCHK(c, hipMalloc((void**)&c->hist, sizeof(unsigned int) * kHistBins * o.max_pairs));
    HIPCHK(c, hipMalloc((void**)&c->scale, sizeof(PairScale) * o.max_pairs));
    HIPCHK(c, hipMemset(c->scale, 0, sizeof(PairScale) * o.max_pairs));
    HIPCHK(c, hipStreamSynchronize(nullptr));   // (the NULL stream's memset against the context's non-blocking streams: see uwt_create)
  }
  c->p = *p;
  c->spec_budget = c->spec_calm = 0;   // a new schedule: the speculative budget starts over
  return UWT_OK;
}

int uwt_level_info(const uwt_ctx* c, int32_t lvl, uwt_level* out) {
  if (!c || !out || lvl < 0 || lvl >= c->p.n_levels) return UWT_ERR_INVALID_ARG;
  *out = c->info[lvl];
  return UWT_OK;
}

int uwt_set_frame(uwt_ctx* c, int32_t slot, const uint8_t* gray, size_t row_stride, const uint16_t* depth,
                  size_t depth_row_stride) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !gray || !slot_range_ok(c, slot, 1)) return fail(c, UWT_ERR_INVALID_ARG, "uwt_set_frame: bad slot/pointer");
  const int w = c->p.width;
  if (row_stride < (size_t)w) return fail(c, UWT_ERR_INVALID_ARG, "uwt_set_frame: row stride < width");
  int st0 = compute_begin(c, slot, 1);
  if (st0) return st0;
  if (c->p.has_depth && (!depth || depth_row_stride < (size_t)w * 2)) return fail(c, UWT_ERR_INVALID_ARG, "uwt_set_frame: depth required");
  // tight or strided host rows alike: one linear copy of the span the rows cover, and a kernel that spreads them where either side
  // is pitched (a 2-D copy is issued row by row: 2.3 ms for a 725 x 465 view)
  if ((st0 = copy_strided_frame_in(c, c->img[0], gray, 1, row_stride, slot, c->stream))) return st0;
  if (c->p.has_depth && (st0 = copy_strided_frame_in(c, c->depth[0], depth, 2, depth_row_stride, slot, c->stream))) return st0;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWT_OK;
}

int uwt_host_alloc(size_t bytes, void** out) {
  if (!out || !bytes) return UWT_ERR_INVALID_ARG;
  *out = nullptr;
  return hipHostMalloc(out, bytes) == hipSuccess ? UWT_OK : UWT_ERR_HIP;
}

int uwt_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? UWT_OK : UWT_ERR_HIP; }

int uwt_upload_frames_async(uwt_ctx* c, int32_t first_slot, int32_t n, const uint8_t* gray, const uint16_t* depth) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !gray || !slot_range_ok(c, first_slot, n)) return fail(c, UWT_ERR_INVALID_ARG, "uwt_upload_frames_async: bad range");
  if (n == 0) return UWT_OK;
  // behind the compute work that still reads or writes these slots, beside everything else on the context stream
  int st = dep_wait(c, c->busy, c->busy_next, c->busy_dropped, c->copy, first_slot, n);
  if (st) return st;
  st = copy_frames_in(c, c->img[0], gray, 1, first_slot, n, c->copy, 1);
  if (st) return st;
  if (c->p.has_depth && depth && (st = copy_frames_in(c, c->depth[0], depth, 2, first_slot, n, c->copy, 1))) return st;
  return dep_note(c, c->fresh, c->fresh_next, c->fresh_dropped, c->copy, first_slot, n);
}

int uwt_upload_frames(uwt_ctx* c, int32_t first_slot, int32_t n, const uint8_t* gray, const uint16_t* depth) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !gray || !slot_range_ok(c, first_slot, n)) return fail(c, UWT_ERR_INVALID_ARG, "uwt_upload_frames: bad range");
  int st0 = compute_begin(c, first_slot, n);   // on the context stream: behind asynchronous uploads into the same slots
  if (st0) return st0;
  if (n && (st0 = copy_frames_in(c, c->img[0], gray, 1, first_slot, n, c->stream, 0))) return st0;
  if (c->p.has_depth) {
    if (!depth) return fail(c, UWT_ERR_INVALID_ARG, "uwt_upload_frames: depth required");
    if (n && (st0 = copy_frames_in(c, c->depth[0], depth, 2, first_slot, n, c->stream, 0))) return st0;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWT_OK;
}

static int plane_ptr(uwt_ctx* c, int slot, int lvl, int plane, void** out, size_t* elem) {
  if (!c || !out || !slot_range_ok(c, slot, 1) || lvl < 0 || lvl >= c->p.n_levels) return UWT_ERR_INVALID_ARG;
  const size_t n = c->lv[lvl].n;
  switch (plane) {
    case UWT_PLANE_IMAGE: *out = c->img[lvl] + slot * n; *elem = 1; break;
    case UWT_PLANE_DEPTH:
      if (!c->p.has_depth) return UWT_ERR_INVALID_ARG;
      *out = c->depth[lvl] + slot * n; *elem = 2; break;
    case UWT_PLANE_GRADX: *out = c->gx[lvl] + slot * n; *elem = 2; break;
    case UWT_PLANE_GRADY: *out = c->gy[lvl] + slot * n; *elem = 2; break;
    default: return UWT_ERR_INVALID_ARG;
  }
  return UWT_OK;
}

int uwt_plane_device_ptr(uwt_ctx* c, int32_t slot, int32_t lvl, int32_t plane, void** out) {
  size_t elem;
  int st = plane_ptr(c, slot, lvl, plane, out, &elem);
  return st ? fail(c, st, "uwt_plane_device_ptr: bad slot/level/plane") : UWT_OK;
}

int uwt_get_plane(uwt_ctx* c, int32_t slot, int32_t lvl, int32_t plane, void* host_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  void* d;
  size_t elem;
  int st = plane_ptr(c, slot, lvl, plane, &d, &elem);
  if (st || !host_out) return fail(c, UWT_ERR_INVALID_ARG, "uwt_get_plane: bad slot/level/plane");
  const LevelK& L = c->lv[lvl];   // the level's image, img_w x img_h, without the pad columns of its rows
  HIPCHK(c, hipMemcpy2DAsync(host_out, (size_t)L.iw * elem, d, (size_t)L.pitch * elem, (size_t)L.iw * elem, L.ih, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWT_OK;
}

// System::AddFrame's pyramid loop for slots first_slot..+n.  depth_slots (device list of n_depth slots), when given,
// restricts the depth pyramids to those slots: only a pair's reference frame is ever read through its depth
// (src/Tracker.cpp:1266-1272).
static int enqueue_pyramids(uwt_ctx* c, int first_slot, int n, const int* depth_slots = nullptr, int n_depth = 0) {
  const bool fused = n <= kFewFrames && c->p.n_levels >= 3 && c->p.n_levels <= kPyrMaxLevels && c->whole && c->lv[0].iw % 4 == 0 &&
                     c->lv[0].ih % 4 == 0 && c->tn.fused_stages;
  if (fused) {   // the whole pyramid of each plane in one launch
    if (n) launch_pyramid_all<uint8_t>(c, c->img, c->img, n, nullptr, first_slot);
    if (c->p.has_depth) {
      if (depth_slots) { if (n_depth) launch_pyramid_all<uint16_t>(c, c->depth, c->depth, n_depth, depth_slots, 0); }
      else if (n) launch_pyramid_all<uint16_t>(c, c->depth, c->depth, n, nullptr, first_slot);
    }
    HIPCHK(c, hipGetLastError());
    return UWT_OK;
  }
  // batches: levels 1..3 in one pass over level 0 (k_pyramid_batch), the levels beyond by the per-level chain
  int l0 = 1;
  if (c->p.n_levels >= 4 && c->lv[0].iw % 16 == 0 && c->lv[0].ih % 8 == 0 && c->tn.pyramid_batch) {
    const int tiles = ((c->lv[0].iw + 127) / 128) * ((c->lv[0].ih + 63) / 64);
    if (n) {
      PyramidBatchArgs<uint8_t> a;
      a.src = c->img[0];
      for (int l = 0; l < 4; l++) { a.stride[l] = c->lv[l].n; a.pitch[l] = c->lv[l].pitch; if (l) a.dst[l - 1] = c->img[l]; }
      a.w = c->lv[0].iw; a.h = c->lv[0].ih; a.slots = nullptr; a.first_slot = first_slot;
      hipLaunchKernelGGL(k_pyramid_batch<uint8_t>, dim3(tiles, n), dim3(kBlock), 0, c->stream, a);
    }
    const int nd16 = !c->p.has_depth ? 0 : (depth_slots ? n_depth : n);
    if (nd16) {
      PyramidBatchArgs<uint16_t> a;
      a.src = c->depth[0];
      for (int l = 0; l < 4; l++) { a.stride[l] = c->lv[l].n; a.pitch[l] = c->lv[l].pitch; if (l) a.dst[l - 1] = c->depth[l]; }
      a.w = c->lv[0].iw; a.h = c->lv[0].ih; a.slots = depth_slots; a.first_slot = depth_slots ? 0 : first_slot;
      hipLaunchKernelGGL(k_pyramid_batch<uint16_t>, dim3(tiles, nd16), dim3(kBlock), 0, c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    l0 = 4;
  }
  for (int l = l0; l < c->p.n_levels; l++) {
    const LevelK& S = c->lv[l - 1];
    const LevelK& D = c->lv[l];
    int st = launch_resize<uint8_t>(c, c->img[l - 1], c->img[l], S.iw, S.ih, S.pitch, D.iw, D.ih, D.pitch, S.n, D.n, n, nullptr, first_slot);
    if (st) return st;
    if (c->p.has_depth) {
      st = depth_slots ? launch_resize<uint16_t>(c, c->depth[l - 1], c->depth[l], S.iw, S.ih, S.pitch, D.iw, D.ih, D.pitch, S.n, D.n, n_depth, depth_slots)
                       : launch_resize<uint16_t>(c, c->depth[l - 1], c->depth[l], S.iw, S.ih, S.pitch, D.iw, D.ih, D.pitch, S.n, D.n, n, nullptr, first_slot);
      if (st) return st;
    }
  }
  return UWT_OK;
}

static int enqueue_gradient_level(uwt_ctx* c, int l, int first_slot, int n, const int* d_slots, hipStream_t on = nullptr) {
  return launch_scharr(c, c->img[l], c->gx[l], c->gy[l], c->lv[l].iw, c->lv[l].ih, c->lv[l].pitch, c->lv[l].n, n, d_slots, first_slot, on);
}

static int enqueue_gradients(uwt_ctx* c, int first_slot, int n, const int* d_slots = nullptr) {
  if (n && n <= kFewFrames && c->p.n_levels <= kGradMaxLevels && c->tn.fused_stages) {   // every level in one launch
    GradLevelsArgs a;
    std::memset(&a, 0, sizeof(a));
    int tiles = 0;
    for (int l = 0; l < c->p.n_levels; l++) {
      a.src[l] = c->img[l]; a.gx[l] = c->gx[l]; a.gy[l] = c->gy[l];
      a.w[l] = c->lv[l].iw; a.h[l] = c->lv[l].ih; a.pitch[l] = c->lv[l].pitch; a.stride[l] = c->lv[l].n;
      tiles += ((a.w[l] + kGradVW - 1) / kGradVW) * ((a.h[l] + kGradVRows - 1) / kGradVRows);
      a.tile_end[l] = tiles;
    }
    a.n_levels = c->p.n_levels;
    a.slots = d_slots;
    a.first_slot = first_slot;
    hipLaunchKernelGGL(k_scharr3_levels, dim3(tiles, n), dim3(kBlock), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return UWT_OK;
  }
  for (int l = 0; l < c->p.n_levels; l++) {
    int st = enqueue_gradient_level(c, l, first_slot, n, d_slots);
    if (st) return st;
  }
  return UWT_OK;
}

int uwt_set_deferred(uwt_ctx* c, int32_t on) {
  if (!c) return UWT_ERR_INVALID_ARG;
  c->deferred = on != 0;
  return UWT_OK;
}

int uwt_build_pyramids(uwt_ctx* c, int32_t first_slot, int32_t n) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !slot_range_ok(c, first_slot, n)) return fail(c, UWT_ERR_INVALID_ARG, "uwt_build_pyramids: bad range");
  int st = compute_begin(c, first_slot, n);
  if (st) return st;
  st = enqueue_pyramids(c, first_slot, n);
  if (st) return st;
  if (!c->deferred) HIPCHK(c, hipStreamSynchronize(c->stream));
  else return compute_end(c, first_slot, n);   // still in flight: a later asynchronous upload into these slots waits for it
  return UWT_OK;
}

int uwt_apply_gradient(uwt_ctx* c, int32_t first_slot, int32_t n) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !slot_range_ok(c, first_slot, n)) return fail(c, UWT_ERR_INVALID_ARG, "uwt_apply_gradient: bad range");
  int st = compute_begin(c, first_slot, n);
  if (st) return st;
  st = enqueue_gradients(c, first_slot, n);
  if (st) return st;
  if (!c->deferred) HIPCHK(c, hipStreamSynchronize(c->stream));
  else return compute_end(c, first_slot, n);   // still in flight: a later asynchronous upload into these slots waits for it
  return UWT_OK;
}

int uwt_estimate_pose_batch(uwt_ctx* c, int32_t n_pairs, const int32_t* ref_slots, const int32_t* tgt_slots,
                            float* poses_out, uwt_stats* stats_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !poses_out) return fail(c, UWT_ERR_INVALID_ARG, "uwt_estimate_pose_batch: null argument");
  if (!ref_slots || !tgt_slots || n_pairs < 1) return fail(c, UWT_ERR_INVALID_ARG, "null pair lists or n_pairs < 1");
  if (n_pairs > c->p.max_pairs) return fail(c, UWT_ERR_CAPACITY, "n_pairs exceeds max_pairs");
  int st;
  c->inline_pairs = n_pairs <= 2 && takes_chained_flow(c, n_pairs);
  if (c->inline_pairs) {   // the slots travel in the kernel arguments
    for (int i = 0; i < n_pairs; i++) {
      if (ref_slots[i] < 0 || ref_slots[i] >= c->p.max_frames || tgt_slots[i] < 0 || tgt_slots[i] >= c->p.max_frames) {
        c->inline_pairs = false;
        return fail(c, UWT_ERR_INVALID_ARG, "pair slot out of range");
      }
      c->pair_slots[2 * i] = ref_slots[i];
      c->pair_slots[2 * i + 1] = tgt_slots[i];
    }
  } else {
    st = upload_pairs(c, n_pairs, ref_slots, tgt_slots);
    if (st) return st;
  }
  st = compute_begin(c, 0, c->p.max_frames);
  if (st) { c->inline_pairs = false; return st; }
  std::vector<uwt_stats> tmp(n_pairs);
  // A small batch has its results written by the last kernel straight into page-locked host memory: nothing to copy back.
  const bool small = n_pairs <= uwt_ctx::kSmallBatch;
  static_assert(sizeof(StatsOut) == sizeof(uwt_stats), "stats are copied as they are");
  // One or two pairs under an early-exit schedule are launched speculatively (see enqueue_estimate_chained): no read-back
  // inside the alignment, one look at the "cut short" flag behind the results.  If it is set the alignment is run again with
  // twice the evaluations per level (the budget stays: the next frames of a sequence tend to need what this one needed; it
  // comes down again after kSpecCalm calls that were not cut), and the careful way — read-backs — if that is cut short too.
  c->speculate = n_pairs <= 2 && c->p.early_exit && !c->profiling && takes_chained_flow(c, n_pairs) && c->tn.speculation;
  for (int attempt = 0; attempt < 3; attempt++) {
    st = enqueue_estimate(c, n_pairs, small ? c->d_small->poses : c->d_poses, small ? c->d_small->stats : c->d_stats);
    if (st) { c->speculate = false; c->inline_pairs = false; return st; }
    if (!small) {
      HIPCHK(c, hipMemcpyAsync(poses_out, c->d_poses, sizeof(float) * 7 * n_pairs, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(tmp.data(), c->d_stats, sizeof(uwt_stats) * n_pairs, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (small) {
      std::memcpy(poses_out, c->h_small->poses, sizeof(float) * 7 * n_pairs);
      std::memcpy(tmp.data(), c->h_small->stats, sizeof(uwt_stats) * n_pairs);
    }
    if (!c->speculate) break;
    const int base = c->tn.first_poll + 1;
    if (c->h_small->cut == 0) {
      if (c->spec_budget > base && ++c->spec_calm >= uwt_ctx::kSpecCalm) { c->spec_budget = std::max(base, c->spec_budget / 2); c->spec_calm = 0; }
      break;
    }
    c->spec_calm = 0;
    c->spec_budget = std::min(c->p.max_iters, 2 * std::max(c->spec_budget, base));
    if (attempt == 1) c->speculate = false;   // cut short twice: the third run reads back
  }
  c->speculate = false;
  c->inline_pairs = false;
  if (c->profiling) {
    st = prof_collect(c);
    if (st) return st;
  }
  int worst = UWT_OK;
  for (int i = 0; i < n_pairs; i++) {
    if (stats_out) stats_out[i] = tmp[i];
    if (tmp[i].status != UWT_OK && worst == UWT_OK) worst = tmp[i].status;
  }
  if (worst)
    return fail(c, UWT_ERR_PAIR_FAILED, std::string("uwt_estimate_pose_batch: at least one pair failed, first status: ") +
                                            uwt_status_string(worst));
  return UWT_OK;
}

static int track_batch_enqueue(uwt_ctx* c, int32_t first_slot, int32_t n_frames, int32_t grad_refs_only, int32_t n_pairs,
                               const int32_t* ref_slots, const int32_t* tgt_slots, float* d_poses_out, uwt_stats* d_stats_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !d_poses_out || !slot_range_ok(c, first_slot, n_frames))
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_track_batch_async: bad argument");
  c->inline_pairs = false;   // (only the synchronous small call hands slots over in kernel arguments or speculates)
  c->speculate = false;
  int st = upload_pairs(c, n_pairs, ref_slots, tgt_slots);
  if (st) return st;
  // The alignment reads every slot the pair lists name, and with grad_refs_only the reference slots lie "wherever they
  // lie" (uwt.h): the dependency range of the call is the union of the prepared range and the pairs' slots, so that an
  // asynchronous upload into ANY slot this call reads waits for it, and this call for any upload into them.
  int lo = first_slot, hi = first_slot + n_frames;
  for (int i = 0; i < n_pairs; i++) {
    lo = std::min(lo, std::min(ref_slots[i], tgt_slots[i]));
    hi = std::max(hi, std::max(ref_slots[i], tgt_slots[i]) + 1);
  }
  c->dep_first = lo;
  c->dep_n = hi - lo;
  st = compute_begin(c, c->dep_first, c->dep_n);   // behind the asynchronous uploads into these slots
  if (st) return st;
  // The tracker reads gradients and depth of the previous (reference) frame only (src/Tracker.cpp:407-408, 1266-1272).
  // grad_refs_only computes those planes — gradients of every level, depth levels 1.. — for the pairs' reference slots
  // alone; otherwise every prepared frame gets them, as System::AddFrame / System::Tracking do for each new frame
  // (src/System.cpp:246-251, 197-213).
  const int* g_slots = grad_refs_only ? c->d_ref : nullptr;   // gradient frames: a slot list or the range
  const int g_first = g_slots ? 0 : first_slot, g_n = g_slots ? n_pairs : n_frames;
  st = grad_refs_only ? enqueue_pyramids(c, first_slot, n_frames, c->d_ref, n_pairs) : enqueue_pyramids(c, first_slot, n_frames);
  if (st) return st;
  // The side stream pays from ~768 pairs on (+1.7 % at 1024); below, its events cost more than the overlap returns
  // (one pair: +14 % latency).  A profiled call times its kernels alone: nothing runs beside them.
  if (!c->tn.overlap_gradients || c->profiling || n_pairs < 768) {
    st = enqueue_gradients(c, g_first, g_n, g_slots);
    if (st) return st;
    return enqueue_estimate(c, n_pairs, d_poses_out, reinterpret_cast<StatsOut*>(d_stats_out));
  }
  // The alignment starts at the coarsest iterated level and only then needs the finer gradients: the first level's are
  // computed here, the others on the side stream, each level's event gating the iterations that read it.
  HIPCHK(c, hipEventRecord(c->ev_pyramids, c->stream));
  st = enqueue_gradient_level(c, c->p.first_level, g_first, g_n, g_slots);
  if (st) return st;
  HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_pyramids, 0));
  for (int l = c->p.first_level - 1; l >= 0; l--) {
    st = enqueue_gradient_level(c, l, g_first, g_n, g_slots, c->side);
    if (st) return st;
    HIPCHK(c, hipEventRecord(c->ev_level[l], c->side));
  }
  for (int l = c->p.first_level + 1; l < c->p.n_levels; l++) {  // levels the solver never iterates: ApplyGradient still fills them
    st = enqueue_gradient_level(c, l, g_first, g_n, g_slots, c->side);
    if (st) return st;
  }
  HIPCHK(c, hipEventRecord(c->ev_side_done, c->side));
  st = enqueue_estimate(c, n_pairs, d_poses_out, reinterpret_cast<StatsOut*>(d_stats_out), c->ev_level);
  // whatever follows on the context stream (the next call's pyramids, a plane read-back) is ordered after the side work
  const hipError_t e = hipStreamWaitEvent(c->stream, c->ev_side_done, 0);
  if (st) return st;
  if (e != hipSuccess) return fail(c, UWT_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
  return UWT_OK;
}

int uwt_track_batch_async(uwt_ctx* c, int32_t first_slot, int32_t n_frames, int32_t grad_refs_only, int32_t n_pairs,
                          const int32_t* ref_slots, const int32_t* tgt_slots, float* d_poses_out, uwt_stats* d_stats_out) {
  int st = track_batch_enqueue(c, first_slot, n_frames, grad_refs_only, n_pairs, ref_slots, tgt_slots, d_poses_out, d_stats_out);
  if (st) return st;
  return compute_end(c, c->dep_first, c->dep_n);
}

int uwt_track_batch_host_async(uwt_ctx* c, int32_t first_slot, int32_t n_frames, int32_t grad_refs_only, int32_t n_pairs,
                               const int32_t* ref_slots, const int32_t* tgt_slots, float* h_poses_out, uwt_stats* h_stats_out,
                               int64_t* ticket_out) {
  if (!c || !h_poses_out || !ticket_out) return fail(c, UWT_ERR_INVALID_ARG, "uwt_track_batch_host_async: null argument");
  int st = track_batch_enqueue(c, first_slot, n_frames, grad_refs_only, n_pairs, ref_slots, tgt_slots, c->d_poses,
                               reinterpret_cast<uwt_stats*>(c->d_stats));
  if (st) return st;
  // results follow the alignment on the context stream into the caller's (page-locked) buffers; the context's own device
  // buffers are free again before the next call's write-out because the stream is in order
  HIPCHK(c, hipMemcpyAsync(h_poses_out, c->d_poses, sizeof(float) * 7 * n_pairs, hipMemcpyDeviceToHost, c->stream));
  if (h_stats_out)
    HIPCHK(c, hipMemcpyAsync(h_stats_out, c->d_stats, sizeof(uwt_stats) * n_pairs, hipMemcpyDeviceToHost, c->stream));
  st = compute_end(c, c->dep_first, c->dep_n);
  if (st) return st;
  *ticket_out = c->ticket_seq;
  return UWT_OK;
}

int uwt_wait_ticket(uwt_ctx* c, int64_t ticket) {
  if (!c) return UWT_ERR_INVALID_ARG;
  (void)hipSetDevice(c->p.device);
  if (ticket < 1 || ticket > c->ticket_seq) return fail(c, UWT_ERR_INVALID_ARG, "uwt_wait_ticket: unknown ticket");
  for (int i = 0; i < uwt_ctx::kDeps; i++)
    if (c->busy[i].used && c->busy_seq[i] == ticket) {
      HIPCHK(c, hipEventSynchronize(c->busy[i].ev));
      return UWT_OK;
    }
  // older than the ring: everything recorded before the oldest surviving entry is complete when that entry is
  HIPCHK(c, hipEventSynchronize(c->busy[c->busy_next].ev));
  return UWT_OK;
}


int uwt_sync(uwt_ctx* c) {
  if (!c) return UWT_ERR_INVALID_ARG;
  (void)hipSetDevice(c->p.device);
  HIPCHK(c, hipStreamSynchronize(c->copy));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->profiling) {
    int st = prof_collect(c);
    if (st) return st;
  }
  return UWT_OK;
}

int uwt_stream(uwt_ctx* c, void** out) {
  if (!c || !out) return UWT_ERR_INVALID_ARG;
  *out = (void*)c->stream;
  return UWT_OK;
}

int uwt_profile_enable(uwt_ctx* c, int32_t on) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c) return UWT_ERR_INVALID_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->ev_used = 0;
  c->profiling = (on & 1) != 0;
  c->compute_only = (on & 2) != 0;
  c->prof_ms = 0.0;
  c->prof_launches = 0;
  c->prof_pixels = 0;
  for (int l = 0; l < UWT_MAX_LEVELS; l++) { c->prof_level_ms[l] = 0.0; c->prof_level_launches[l] = 0; }
  return UWT_OK;
}

int uwt_profile_read_levels(uwt_ctx* c, double* ms_by_level, int64_t* launches_by_level, int32_t n_levels) {
  if (!c || !ms_by_level || !launches_by_level || n_levels < 1 || n_levels > UWT_MAX_LEVELS) return UWT_ERR_INVALID_ARG;
  for (int l = 0; l < n_levels; l++) { ms_by_level[l] = c->prof_level_ms[l]; launches_by_level[l] = c->prof_level_launches[l]; }
  return UWT_OK;
}

int uwt_profile_read(uwt_ctx* c, double* ms_total, int64_t* launches, int64_t* pixels) {
  if (!c) return UWT_ERR_INVALID_ARG;
  if (ms_total) *ms_total = c->prof_ms;
  if (launches) *launches = c->prof_launches;
  if (pixels) *pixels = c->prof_pixels;
  return UWT_OK;
}


int uwt_profile_clock(uwt_ctx* c, double* shader_ghz) {
  if (c) (void)hipSetDevice(c->p.device);
  if (!c || !shader_ghz) return UWT_ERR_INVALID_ARG;
  *shader_ghz = 0.0;
  if (!c->prof_slices || !c->prof_pairs) return fail(c, UWT_ERR_INVALID_ARG, "no profiled residual launch yet");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int n = std::min(c->prof_pairs, 64);
  double cyc = 0.0, sec = 0.0;
  for (int p = 0; p < n; p++) {  // slice 0 of the first pairs of the last profiled launch
    uint32_t w[2];
    HIPCHK(c, hipMemcpy(w, (c->prof_records ? c->prof_records : c->partials) + ((size_t)p * c->prof_slices) * kRecWords + 60, sizeof(w), hipMemcpyDeviceToHost));
    cyc += (double)w[0];
    sec += (double)w[1] * 1e-8;  // s_memrealtime: 100 MHz
  }
  if (sec > 0.0) *shader_ghz = cyc / sec * 1e-9;
  return UWT_OK;
}

/* ---- per-stage entry points ---------------------------------------------------------------------------------- */

int uwt_halve_u8(uwt_ctx* c, const uint8_t* src, int32_t w, int32_t h, uint8_t* dst) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !src || !dst || w < 2 || h < 2 || (w & 1) || (h & 1)) return fail(c, UWT_ERR_INVALID_ARG, "uwt_halve_u8");
  return resize_half_host<uint8_t>(c, src, w, h, dst, w / 2, h / 2);
}

int uwt_halve_u16(uwt_ctx* c, const uint16_t* src, int32_t w, int32_t h, uint16_t* dst) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !src || !dst || w < 2 || h < 2 || (w & 1) || (h & 1)) return fail(c, UWT_ERR_INVALID_ARG, "uwt_halve_u16");
  return resize_half_host<uint16_t>(c, src, w, h, dst, w / 2, h / 2);
}

int uwt_half_size(int32_t n) { return (int)std::lrint((double)n * 0.5); }   // cvRound(n * 0.5): half to even

int uwt_resize_half_u8(uwt_ctx* c, const uint8_t* src, int32_t w, int32_t h, uint8_t* dst) {
  if (c) (void)hipSetDevice(c->p.device);
  if (!c || !src || !dst || w < 1 || h < 1 || uwt_half_size(w) < 1 || uwt_half_size(h) < 1) return fail(c, UWT_ERR_INVALID_ARG, "uwt_resize_half_u8");
  return resize_half_host<uint8_t>(c, src, w, h, dst, uwt_half_size(w), uwt_half_size(h));
}

int uwt_resize_half_u16(uwt_ctx* c, const uint16_t* src, int32_t w, int32_t h, uint16_t* dst) {
  if (c) (void)hipSetDevice(c->p.device);
  if (!c || !src || !dst || w < 1 || h < 1 || uwt_half_size(w) < 1 || uwt_half_size(h) < 1) return fail(c, UWT_ERR_INVALID_ARG, "uwt_resize_half_u16");
  return resize_half_host<uint16_t>(c, src, w, h, dst, uwt_half_size(w), uwt_half_size(h));
}

int uwt_scharr3(uwt_ctx* c, const uint8_t* src, int32_t w, int32_t h, int16_t* gx, int16_t* gy) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !src || !gx || !gy || w < 1 || h < 1) return fail(c, UWT_ERR_INVALID_ARG, "uwt_scharr3");
  const size_t pitch = ((size_t)w + 3) & ~(size_t)3;   // rows padded to whole groups of four, as the context's level planes are
  const size_t n = pitch * h, off = (n + 255) & ~(size_t)255;
  int st = ensure_scratch(c, off + n * 4);
  if (st) return st;
  uint8_t* d = (uint8_t*)c->scratch;
  int16_t* dgx = (int16_t*)(d + off);
  int16_t* dgy = dgx + n;
  HIPCHK(c, hipMemcpy2DAsync(d, pitch, src, w, w, h, hipMemcpyHostToDevice, c->stream));
  st = launch_scharr(c, d, dgx, dgy, w, h, (int)pitch, n, 1);
  if (st) return st;
  HIPCHK(c, hipMemcpy2DAsync(gx, (size_t)w * 2, dgx, pitch * 2, (size_t)w * 2, h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(gy, (size_t)w * 2, dgy, pitch * 2, (size_t)w * 2, h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWT_OK;
}

int uwt_warp(uwt_ctx* c, int32_t lvl, const float* pts, int32_t n, const float pose[7], float* warped_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !pts || !pose || !warped_out || n < 1 || lvl < 0 || lvl >= c->p.n_levels)
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_warp");
  const size_t bytes = sizeof(float) * 4 * (size_t)n;
  int st = ensure_scratch(c, bytes * 2);
  if (st) return st;
  float4* din = (float4*)c->scratch;
  float4* dout = din + n;
  HIPCHK(c, hipMemcpyAsync(din, pts, bytes, hipMemcpyHostToDevice, c->stream));
  Pose P;
  for (int k = 0; k < 4; k++) P.q[k] = pose[k];
  for (int k = 0; k < 3; k++) P.t[k] = pose[4 + k];
  UWT_WITH_AR(c->p.arith == UWT_ARITH_LEGACY ? kArithLegacy : kArithOpenCV, hipLaunchKernelGGL(k_warp_table<AR>, dim3((n + 255) / 256), dim3(256), 0, c->stream, din, dout, n, P, c->lv[lvl]));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(warped_out, dout, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWT_OK;
}

int uwt_residual_jacobian(uwt_ctx* c, int32_t ref_slot, int32_t tgt_slot, int32_t lvl, const float pose[7],
                          uwt_accum* acc_out, float* J_out, float* r_out, uint8_t* valid_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !pose || !acc_out || lvl < 0 || lvl >= c->p.n_levels || !slot_range_ok(c, ref_slot, 1) || !slot_range_ok(c, tgt_slot, 1))
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_residual_jacobian");
  int st = upload_pairs(c, 1, &ref_slot, &tgt_slot);
  if (st) return st;
  const LevelK& L = c->lv[lvl];
  const size_t n = L.ng;   // device dumps are indexed like the planes (pitch x gh positions); the host receives the gw x gh grid
  const bool dump = J_out || r_out || valid_out;
  if (dump) {
    st = ensure_scratch(c, n * (6 * 4 + 4 + 1) + 512);
    if (st) return st;
  }
  ResidualArgs a = residual_args(c, lvl);
  a.state = nullptr;
  for (int k = 0; k < 4; k++) a.pose.q[k] = pose[k];
  for (int k = 0; k < 3; k++) a.pose.t[k] = pose[4 + k];
  if (dump) {
    a.dumpJ = (float*)c->scratch;
    a.dumpR = a.dumpJ + 6 * n;
    a.dumpV = (uint8_t*)(a.dumpR + n);
  }
  st = launch_residual(c, a, 1, dump);
  if (st) return st;
  std::vector<uint32_t> recs((size_t)a.slices * kRecWords);
  HIPCHK(c, hipMemcpyAsync(recs.data(), c->partials, recs.size() * 4, hipMemcpyDeviceToHost, c->stream));
  if (J_out) HIPCHK(c, hipMemcpy2DAsync(J_out, (size_t)L.gw * 24, a.dumpJ, (size_t)L.pitch * 24, (size_t)L.gw * 24, L.gh, hipMemcpyDeviceToHost, c->stream));
  if (r_out) HIPCHK(c, hipMemcpy2DAsync(r_out, (size_t)L.gw * 4, a.dumpR, (size_t)L.pitch * 4, (size_t)L.gw * 4, L.gh, hipMemcpyDeviceToHost, c->stream));
  if (valid_out) HIPCHK(c, hipMemcpy2DAsync(valid_out, (size_t)L.gw, a.dumpV, (size_t)L.pitch, (size_t)L.gw, L.gh, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memset(acc_out, 0, sizeof(*acc_out));
  for (int s = 0; s < a.slices; s++) {  // same slice-ordered f64 fold as k_gn_update
    const uint32_t* r = recs.data() + (size_t)s * kRecWords;
    double d[27];
    std::memcpy(d, r, sizeof(d));
    for (int k = 0; k < 21; k++) acc_out->A[k] += d[k];
    for (int k = 0; k < 6; k++) acc_out->jtr[k] += d[21 + k];
    acc_out->n_valid += (int32_t)r[54];
    int64_t sr2;
    std::memcpy(&sr2, r + 56, 8);
    acc_out->sum_r2 += sr2;
  }
  return UWT_OK;
}

int uwt_residual_jacobian_weighted(uwt_ctx* c, int32_t ref_slot, int32_t tgt_slot, int32_t lvl, const float pose[7],
                                   uwt_accum* acc_out, double* err_num_out, float* inv_mad_out, float* J_out, float* r_out,
                                   uint8_t* valid_out, float* w_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !pose || !acc_out || lvl < 0 || lvl >= c->p.n_levels || !slot_range_ok(c, ref_slot, 1) || !slot_range_ok(c, tgt_slot, 1))
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_residual_jacobian_weighted");
  if (!c->p.sampler && !c->p.weights)
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_residual_jacobian_weighted: context uses the nearest/identity fast path");
  int st = upload_pairs(c, 1, &ref_slot, &tgt_slot);
  if (st) return st;
  const LevelK& L = c->lv[lvl];
  const size_t n = L.ng;   // see uwt_residual_jacobian
  st = ensure_scratch(c, n * (6 * 4 + 4 + 4 + 1) + 512);
  if (st) return st;
  Pose P;
  for (int k = 0; k < 4; k++) P.q[k] = pose[k];
  for (int k = 0; k < 3; k++) P.t[k] = pose[4 + k];
  hipLaunchKernelGGL(k_set_pose, dim3(1), dim3(64), 0, c->stream, c->state, P, c->p.initial_error);
  HIPCHK(c, hipGetLastError());
  ResidualArgs a = residual_args(c, lvl);
  a.dumpJ = (float*)c->scratch;
  a.dumpR = a.dumpJ + 6 * n;
  a.dumpW = a.dumpR + n;
  a.dumpV = (uint8_t*)(a.dumpW + n);
  int slices = 0;
  st = launch_general_dump(c, a, 1, &slices);
  if (st) return st;
  std::vector<uint32_t> recs((size_t)slices * kRecWords);
  HIPCHK(c, hipMemcpyAsync(recs.data(), c->partials, recs.size() * 4, hipMemcpyDeviceToHost, c->stream));
  PairScale sc;
  std::memset(&sc, 0, sizeof(sc));
  sc.inv_mad = 1.f;
  if (c->p.weights) HIPCHK(c, hipMemcpyAsync(&sc, c->scale, sizeof(sc), hipMemcpyDeviceToHost, c->stream));
  if (J_out) HIPCHK(c, hipMemcpy2DAsync(J_out, (size_t)L.gw * 24, a.dumpJ, (size_t)L.pitch * 24, (size_t)L.gw * 24, L.gh, hipMemcpyDeviceToHost, c->stream));
  if (r_out) HIPCHK(c, hipMemcpy2DAsync(r_out, (size_t)L.gw * 4, a.dumpR, (size_t)L.pitch * 4, (size_t)L.gw * 4, L.gh, hipMemcpyDeviceToHost, c->stream));
  if (w_out) HIPCHK(c, hipMemcpy2DAsync(w_out, (size_t)L.gw * 4, a.dumpW, (size_t)L.pitch * 4, (size_t)L.gw * 4, L.gh, hipMemcpyDeviceToHost, c->stream));
  if (valid_out) HIPCHK(c, hipMemcpy2DAsync(valid_out, (size_t)L.gw, a.dumpV, (size_t)L.pitch, (size_t)L.gw, L.gh, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memset(acc_out, 0, sizeof(*acc_out));
  double err = 0.0;
  for (int s = 0; s < slices; s++) {
    const uint32_t* r = recs.data() + (size_t)s * kRecWords;
    double d[30];
    std::memcpy(d, r, sizeof(d));
    for (int k = 0; k < 21; k++) acc_out->A[k] += d[k];
    for (int k = 0; k < 6; k++) acc_out->jtr[k] += d[21 + k];
    err += d[29];
    acc_out->n_valid += (int32_t)r[54];
    int64_t sr2;
    std::memcpy(&sr2, r + 56, 8);
    acc_out->sum_r2 += sr2;
  }
  if (err_num_out) *err_num_out = err;
  if (inv_mad_out) *inv_mad_out = sc.inv_mad;
  return UWT_OK;
}

static int ls_accumulate_impl(uwt_ctx* c, const float* J, const float* r, const float* w, int32_t n, int32_t divide, bool sse,
                              int32_t count, float A[36], float b[6], float* error, int32_t* num_constraints) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !J || !r || !A || !b || !error || !num_constraints || n < 0) return fail(c, UWT_ERR_INVALID_ARG, "uwt_ls_accumulate");
  const size_t fl = (size_t)n * 8 + 128 + 64;
  int st = ensure_scratch(c, fl * 4);
  if (st) return st;
  float* dJ = (float*)c->scratch;
  float* dr = dJ + (size_t)n * 6;
  float* dw = dr + n;
  float* dp = dw + n;
  if (n) {
    HIPCHK(c, hipMemcpyAsync(dJ, J, sizeof(float) * 6 * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dr, r, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    if (w) HIPCHK(c, hipMemcpyAsync(dw, w, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
  }
  // one thread per accumulator chain (per lane chain in the SSE form), each in the reference's order: k_ls_sequential
  if (sse) hipLaunchKernelGGL(k_ls_sequential<true>, dim3(1), dim3(128), 0, c->stream, dJ, dr, w ? dw : nullptr, n, dp);
  else hipLaunchKernelGGL(k_ls_sequential<false>, dim3(1), dim3(128), 0, c->stream, dJ, dr, w ? dw : nullptr, n, dp);
  HIPCHK(c, hipGetLastError());
  float parts[112];
  HIPCHK(c, hipMemcpyAsync(parts, dp, sizeof(float) * (sse ? 112 : 28), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float s[28];
  for (int k = 0; k < 28; k++)   // LS::finishNoDivide (:39-139): the four lanes folded left to right
    s[k] = sse ? ((parts[4 * k] + parts[4 * k + 1]) + parts[4 * k + 2]) + parts[4 * k + 3] : parts[k];
  int q = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++, q++) { A[6 * i + j] = s[q]; A[6 * j + i] = s[q]; }
  for (int i = 0; i < 6; i++) b[i] = -s[21 + i];  // LS stores b = -Σ w r J (src/LeastSquares.cpp:206; 0 - x - y = -(x + y) in IEEE)
  *error = s[27];
  *num_constraints = count;
  if (divide) {          // LS::finish (:141-146)
    const float nf = (float)count;
    for (int i = 0; i < 36; i++) A[i] = A[i] / nf;
    for (int i = 0; i < 6; i++) b[i] = b[i] / nf;
    *error = *error / nf;
  }
  return UWT_OK;
}

int uwt_ls_accumulate(uwt_ctx* c, const float* J, const float* r, const float* w, int32_t n, int32_t divide, float A[36],
                      float b[6], float* error, int32_t* num_constraints) {
  return ls_accumulate_impl(c, J, r, w, n, divide, false, n /* one per LS::update call (:208) */, A, b, error, num_constraints);
}

int uwt_ls_accumulate_sse(uwt_ctx* c, const float* J, const float* r, const float* w, int32_t n, int32_t divide,
                          int32_t count_quirk, float A[36], float b[6], float* error, int32_t* num_constraints) {
  if (n < 0 || (n & 3)) return fail(c, UWT_ERR_INVALID_ARG, "uwt_ls_accumulate_sse: n must be a multiple of 4");
  return ls_accumulate_impl(c, J, r, w, n, divide, true, count_quirk ? (n / 4) * 6 : n, A, b, error, num_constraints);
}

int uwt_se3_exp(uwt_ctx* c, const float xi[6], float pose_out[7]) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!xi || !pose_out) return UWT_ERR_INVALID_ARG;
  return run_se3_op(c, 0, xi, 6, nullptr, 0, pose_out, 7, nullptr);
}

int uwt_se3_mul(uwt_ctx* c, const float a[7], const float b[7], float out[7]) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!a || !b || !out) return UWT_ERR_INVALID_ARG;
  return run_se3_op(c, 1, a, 7, b, 7, out, 7, nullptr);
}

int uwt_se3_matrix(uwt_ctx* c, const float pose[7], float T_out[16]) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!pose || !T_out) return UWT_ERR_INVALID_ARG;
  return run_se3_op(c, 2, pose, 7, nullptr, 0, T_out, 16, nullptr);
}

int uwt_se3_handoff(uwt_ctx* c, float pose[7], int32_t scale_t) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!pose) return UWT_ERR_INVALID_ARG;
  float out[7];
  int flag = 1;
  int st = run_se3_op(c, scale_t ? 4 : 3, pose, 7, nullptr, 0, out, 7, &flag);
  if (st) return st;
  if (!flag) return fail(c, UWT_ERR_INVALID_ARG, "uwt_se3_handoff: quaternion close to zero (SOPHUS_ENSURE)");
  std::memcpy(pose, out, sizeof(out));
  return UWT_OK;
}

int uwt_solve_delta(uwt_ctx* c, const float A[36], const float b[6], float delta_out[6], float* Ainv_out, int32_t* nonsingular) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!A || !b || !delta_out) return UWT_ERR_INVALID_ARG;
  float out[42];
  int flag = 0;
  int st = run_se3_op(c, (c && c->p.arith == UWT_ARITH_LEGACY) ? 6 : 5, A, 36, b, 6, out, 42, &flag);
  if (st) return st;
  std::memcpy(delta_out, out, 6 * sizeof(float));
  if (Ainv_out) std::memcpy(Ainv_out, out + 6, 36 * sizeof(float));
  if (nonsingular) *nonsingular = flag;
  return UWT_OK;
}

int uwt_estimate_pose_points(uwt_ctx* c, int32_t ref_slot, int32_t tgt_slot, const float* const* tables,
                             const int32_t* n_points, float pose_out[7], uwt_stats* stats_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !tables || !n_points || !pose_out) return fail(c, UWT_ERR_INVALID_ARG, "uwt_estimate_pose_points: null argument");
  const uwt_params& p = c->p;
  size_t total = 0;
  for (int l = p.last_level; l <= p.first_level; l++) {
    // one partial record per 8192 points; the context owns max_slices x max_pairs records
    if (n_points[l] < 0 || (size_t)n_points[l] > c->partial_records * (size_t)(kBlock * 32) || (n_points[l] > 0 && !tables[l]))
      return fail(c, UWT_ERR_INVALID_ARG, "uwt_estimate_pose_points: bad table (null, negative or too many points)");
    total += (size_t)n_points[l];
  }
  int st = upload_pairs(c, 1, &ref_slot, &tgt_slot);
  if (st) return st;
  st = ensure_scratch(c, std::max<size_t>(16, total * 16));
  if (st) return st;
  float4* d_tab[UWT_MAX_LEVELS] = {};
  size_t off = 0;
  for (int l = p.last_level; l <= p.first_level; l++) {
    d_tab[l] = (float4*)c->scratch + off;
    if (n_points[l]) HIPCHK(c, hipMemcpyAsync(d_tab[l], tables[l], (size_t)n_points[l] * 16, hipMemcpyHostToDevice, c->stream));
    off += (size_t)n_points[l];
  }
  const int tb = 64;
  hipLaunchKernelGGL(k_init_state, dim3(1), dim3(tb), 0, c->stream, c->state, 1, p.initial_error);
  HIPCHK(c, hipGetLastError());
  const int per_block = kBlock * 32;
  for (int lvl = p.first_level; lvl >= p.last_level; lvl--) {
    ResidualArgs ra = residual_args(c, lvl);
    PointsArgs pa;
    pa.pts = d_tab[lvl];
    pa.n_pts = n_points[lvl];
    pa.pts_per_block = per_block;
    ra.slices = std::max(1, (pa.n_pts + per_block - 1) / per_block);
    UpdateArgs ua = update_args(c, lvl);
    ua.slices = ra.slices;
    const bool general = p.sampler || p.weights;   // robust weights / bilinear sampler: the per-stage form over the table
    if (general) ua.general = 1;
    int next_poll = 2;
    for (int k = 0; k < p.max_iters; k++) {
      if (general) {
        if (p.weights) HIPCHK(c, hipMemsetAsync(c->hist + (size_t)ra.pair_base * kHistBins, 0, sizeof(unsigned int) * kHistBins, c->stream));
        uwt::launch_points_general(c->stream, launch_sel(c), ra, pa, general_args(c));
      } else {
        uwt::launch_points(c->stream, launch_sel(c), ra, pa);
      }
      HIPCHK(c, hipGetLastError());
      ua.k = k;
      const bool poll = p.early_exit && (k + 1 == next_poll) && (k + 1 < p.max_iters);
      ua.active = poll ? c->d_active : nullptr;
      if (poll) HIPCHK(c, hipMemsetAsync(c->d_active, 0, sizeof(int), c->stream));
      hipLaunchKernelGGL(k_gn_update, dim3(1), dim3(kUpdateBlock), 0, c->stream, ua);
      HIPCHK(c, hipGetLastError());
      if (poll) {
        HIPCHK(c, hipMemcpyAsync(c->h_active, c->d_active, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (*c->h_active == 0) break;
        next_poll *= 2;
      }
    }
    hipLaunchKernelGGL(k_level_end, dim3(1), dim3(tb), 0, c->stream, c->state, 1, lvl, p.handoff_scale_t, p.initial_error);
    HIPCHK(c, hipGetLastError());
  }
  hipLaunchKernelGGL(k_write_out, dim3(1), dim3(tb), 0, c->stream, c->state, 1, c->d_poses, c->d_stats);
  HIPCHK(c, hipGetLastError());
  uwt_stats tmp;
  HIPCHK(c, hipMemcpyAsync(pose_out, c->d_poses, sizeof(float) * 7, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&tmp, c->d_stats, sizeof(tmp), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (stats_out) *stats_out = tmp;
  if (tmp.status != UWT_OK)
    return fail(c, UWT_ERR_PAIR_FAILED, std::string("uwt_estimate_pose_points: ") + uwt_status_string(tmp.status));
  return UWT_OK;
}

static int mag_to_scratch(uwt_ctx* c, int slot, int lvl, uint8_t** d_mag, unsigned long long** d_sum) {
  const size_t n = c->lv[lvl].n;
  int st = ensure_scratch(c, n + 64 + (size_t)c->lv[lvl].n * 16 + 64);
  if (st) return st;
  *d_sum = (unsigned long long*)c->scratch;
  *d_mag = (uint8_t*)c->scratch + 64;
  HIPCHK(c, hipMemsetAsync(*d_sum, 0, 8, c->stream));
  const int blocks = (int)std::min<size_t>(1024, (n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_grad_mag, dim3(blocks), dim3(kBlock), 0, c->stream, c->gx[lvl] + slot * n, c->gy[lvl] + slot * n, (int)n,
                     c->lv[lvl].pitch, c->lv[lvl].iw, *d_mag, *d_sum);
  HIPCHK(c, hipGetLastError());
  return UWT_OK;
}

int uwt_robust_weights(uwt_ctx* c, const float* residuals, int32_t n, int32_t kind, float* weights_out, float* median_out,
                       float* mad_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !residuals || n < 1 || (kind != kWeightsIdentity && kind != kWeightsTukeyRef))
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_robust_weights: null residuals, n < 1, or a kind other than identity / Tukey");
  const size_t bytes = sizeof(float) * (size_t)n;
  int st = ensure_scratch(c, 2 * bytes + 64);
  if (st) return st;
  float* d_r = (float*)c->scratch;
  float* d_w = d_r + n;
  float* d_stats = d_w + n;
  HIPCHK(c, hipMemcpyAsync(d_r, residuals, bytes, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_robust_weights, dim3(1), dim3(1024), 0, c->stream, d_r, n, kind, weights_out ? d_w : nullptr, d_stats);
  HIPCHK(c, hipGetLastError());
  float stats[2] = {0.f, 0.f};
  if (weights_out) HIPCHK(c, hipMemcpyAsync(weights_out, d_w, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (median_out) *median_out = stats[0];
  if (mad_out) *mad_out = stats[1];
  return UWT_OK;
}

int uwt_gradient_magnitude(uwt_ctx* c, int32_t slot, int32_t lvl, uint8_t* mag_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !mag_out || !slot_range_ok(c, slot, 1) || lvl < 0 || lvl >= c->p.n_levels)
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_gradient_magnitude");
  uint8_t* d_mag;
  unsigned long long* d_sum;
  int st = mag_to_scratch(c, slot, lvl, &d_mag, &d_sum);
  if (st) return st;
  const LevelK& L = c->lv[lvl];   // gradient_[lvl]: the level's image, img_w x img_h
  HIPCHK(c, hipMemcpy2DAsync(mag_out, L.iw, d_mag, L.pitch, L.iw, L.ih, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWT_OK;
}

int uwt_obtain_candidate_points_batch(uwt_ctx* c, int32_t first_slot, int32_t n_frames, int32_t lvl, double threshold,
                                      float* pts_out, int32_t cap, int32_t* counts_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !counts_out || cap < 0 || (cap > 0 && !pts_out) || n_frames < 1 || !slot_range_ok(c, first_slot, n_frames) || lvl < 0 ||
      lvl >= c->p.n_levels)
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_obtain_candidate_points_batch");
  const LevelK& L = c->lv[lvl];
  const int w = L.gw, h = L.gh;   // the point grid the reference's loops walk (src/Tracker.cpp:1334-1335)
  const size_t n = L.n;
  const int kcap = (int)std::min<size_t>((size_t)cap, (size_t)w * h);
  // row bands: enough blocks for a lone frame to spread over the chip, a few rows per thread at least
  const int col_blocks = (w + kBlock - 1) / kBlock;
  int bands = std::max(1, std::min(h / 8, 512 / std::max(1, col_blocks * n_frames)));
  const size_t m = (size_t)w * bands;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_sums = 0, o_mag = up(8 * (size_t)n_frames), o_cnt = o_mag + up(n * n_frames), o_off = o_cnt + up(4 * m * n_frames),
               o_tot = o_off + up(4 * m * n_frames), o_out = o_tot + up(4 * (size_t)n_frames),
               total = o_out + (size_t)n_frames * kcap * 16 + 256;
  int st = ensure_scratch(c, total);
  if (st) return st;
  st = compute_begin(c, first_slot, n_frames);
  if (st) return st;
  uint8_t* base = (uint8_t*)c->scratch;
  unsigned long long* d_sums = (unsigned long long*)(base + o_sums);
  uint8_t* d_mag = base + o_mag;
  int* d_cnt = (int*)(base + o_cnt);
  int* d_off = (int*)(base + o_off);
  int* d_tot = (int*)(base + o_tot);
  float4* d_out = (float4*)(base + o_out);
  const uint16_t* d_depth = c->p.has_depth ? c->depth[lvl] : nullptr;
  HIPCHK(c, hipMemsetAsync(d_sums, 0, 8 * (size_t)n_frames, c->stream));
  const int mag_blocks = (int)std::min<size_t>(256, (n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_grad_mag_batch, dim3(mag_blocks, n_frames), dim3(kBlock), 0, c->stream, c->gx[lvl], c->gy[lvl], (int)n, L.pitch,
                     L.iw, first_slot, d_mag, d_sums);
  const dim3 grid(col_blocks, bands, n_frames);
  hipLaunchKernelGGL(k_candidates_batch<false>, grid, dim3(kBlock), 0, c->stream, d_mag, d_depth, first_slot, L.pitch, L.iw, L.ih, w, h, bands,
                     d_sums, threshold, d_cnt, (const int*)nullptr, (float4*)nullptr, 0);
  hipLaunchKernelGGL(k_scan_counts, dim3(n_frames), dim3(1024), 0, c->stream, d_cnt, (int)m, d_off, d_tot);
  hipLaunchKernelGGL(k_candidates_batch<true>, grid, dim3(kBlock), 0, c->stream, d_mag, d_depth, first_slot, L.pitch, L.iw, L.ih, w, h, bands,
                     d_sums, threshold, (int*)nullptr, d_off, d_out, kcap);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(counts_out, d_tot, 4 * (size_t)n_frames, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int f = 0; f < n_frames; f++) {   // each frame's points are packed at f * cap in the caller's buffer
    const int k = std::min(counts_out[f], kcap);
    if (k > 0)
      HIPCHK(c, hipMemcpyAsync(pts_out + (size_t)f * cap * 4, d_out + (size_t)f * kcap, (size_t)k * 16, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWT_OK;
}

int uwt_obtain_candidate_points(uwt_ctx* c, int32_t slot, int32_t lvl, double threshold, float* pts_out, int32_t cap,
                                int32_t* count_out) {
  if (!count_out) return fail(c, UWT_ERR_INVALID_ARG, "uwt_obtain_candidate_points");
  return uwt_obtain_candidate_points_batch(c, slot, 1, lvl, threshold, pts_out, cap, count_out);
}

int uwt_obtain_patch_points(uwt_ctx* c, int32_t slot, const float* kp, int32_t n_kp, float* pts_out, int32_t cap,
                            int32_t* count_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !count_out || n_kp < 0 || (n_kp > 0 && !kp) || cap < 0 || (cap > 0 && !pts_out) || !slot_range_ok(c, slot, 1))
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_obtain_patch_points");
  const int w = c->lv[0].gw, h = c->lv[0].gh;   // level 0: grid = image
  for (int i = 0; i < std::min(n_kp, 200); i++)
    if (!(kp[2 * i] >= 0.f && kp[2 * i] < (float)w && kp[2 * i + 1] >= 0.f && kp[2 * i + 1] < (float)h))
      return fail(c, UWT_ERR_INVALID_ARG, "uwt_obtain_patch_points: key point outside the image");
  const int nk = std::min(n_kp, 200);
  const int kcap = std::min(cap, 200 * 144);
  int st = ensure_scratch(c, 4096 + (size_t)kcap * 16 + 64);
  if (st) return st;
  float2* d_kp = (float2*)((uint8_t*)c->scratch + 64);
  float4* d_out = (float4*)((uint8_t*)c->scratch + 4096);
  int* d_cnt = (int*)c->scratch;
  if (nk) HIPCHK(c, hipMemcpyAsync(d_kp, kp, (size_t)nk * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_patch_points, dim3(1), dim3(256), 0, c->stream, d_kp, nk,
                     c->p.has_depth ? c->depth[0] + (size_t)slot * c->lv[0].n : nullptr, c->lv[0].pitch, w, h, d_out, kcap, d_cnt);
  HIPCHK(c, hipGetLastError());
  int cnt = 0;
  HIPCHK(c, hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *count_out = cnt;
  const int m = std::min(cnt, kcap);
  if (m > 0) {
    HIPCHK(c, hipMemcpyAsync(pts_out, d_out, (size_t)m * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return UWT_OK;
}

int uwt_add_patch_points(uwt_ctx* c, int32_t lvl, const float* pts, int32_t n_pts, int32_t patch_size, float* pts_out,
                         int32_t cap, int32_t* count_out) {
  if (c) (void)hipSetDevice(c->p.device);
  if (!c || !count_out || lvl < 0 || lvl >= c->p.n_levels || n_pts < 0 || (n_pts > 0 && !pts) || cap < 0 || (cap > 0 && !pts_out) ||
      patch_size < 1)
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_add_patch_points");
  const int start = (patch_size - 1) / 2;   // src/Tracker.cpp:602
  const size_t in_bytes = (size_t)n_pts * 16, out_bytes = (size_t)cap * 16;
  int st = ensure_scratch(c, 4096 + in_bytes + out_bytes + 64);
  if (st) return st;
  int* d_cnt = (int*)c->scratch;
  float4* d_in = (float4*)((uint8_t*)c->scratch + 4096);
  float4* d_out = (float4*)((uint8_t*)c->scratch + 4096 + in_bytes);
  if (n_pts) HIPCHK(c, hipMemcpyAsync(d_in, pts, in_bytes, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_add_patch_points, dim3(1), dim3(256), 0, c->stream, d_in, n_pts, c->lv[lvl].gw, c->lv[lvl].gh, start, d_out, cap,
                     d_cnt);
  HIPCHK(c, hipGetLastError());
  int cnt = 0;
  HIPCHK(c, hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *count_out = cnt;
  const int m = std::min(cnt, (int)cap);
  if (m > 0) {
    HIPCHK(c, hipMemcpyAsync(pts_out, d_out, (size_t)m * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return UWT_OK;
}

/* ---- frame ingest ------------------------------------------------------------------------------------------------- */

struct uwt_ingest {
  int in_w, in_h, out_w, out_h, device;
  double newK[4];
  std::vector<int16_t> h_map1;
  std::vector<uint16_t> h_map2;
  short2* d_map1 = nullptr;
  uint16_t* d_map2 = nullptr;
  uint8_t* d_raw = nullptr;
  uint8_t* d_und = nullptr;
  hipStream_t stream = nullptr;
};

namespace {

// cvUndistortPoints with 5 fixed iterations and no rectification/projection (normalised output), as
// icvGetRectangles calls it from cvGetOptimalNewCameraMatrix (OpenCV 3.2 calib3d).
void undistort_normalised(double u, double v, const double K[4], const double k[4], float* ox, float* oy) {
  double x = (u - K[2]) / K[0], y = (v - K[3]) / K[1];
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = 1.0 / (1.0 + ((0.0 * r2 + k[1]) * r2 + k[0]) * r2);
    const double dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
    const double dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
    x = (x0 - dX) * icdist;
    y = (y0 - dY) * icdist;
  }
  *ox = (float)x;
  *oy = (float)y;
}

// getOptimalNewCameraMatrix(K, dist, Size(in), alpha = 1, Size(out), nullptr, false)  (src/CameraModel.cpp:89)
void optimal_new_camera_matrix(const double K[4], const double k[4], int in_w, int in_h, double alpha, int new_w, int new_h,
                               double newK[4]) {
  const int N = 9;
  float iX0 = -3.4028235e38f, iX1 = 3.4028235e38f, iY0 = -3.4028235e38f, iY1 = 3.4028235e38f;
  float oX0 = 3.4028235e38f, oX1 = -3.4028235e38f, oY0 = 3.4028235e38f, oY1 = -3.4028235e38f;
  for (int y = 0; y < N; y++)
    for (int x = 0; x < N; x++) {
      const float px = (float)x * in_w / (N - 1), py = (float)y * in_h / (N - 1);
      float qx, qy;
      undistort_normalised(px, py, K, k, &qx, &qy);
      oX0 = std::min(oX0, qx); oX1 = std::max(oX1, qx); oY0 = std::min(oY0, qy); oY1 = std::max(oY1, qy);
      if (x == 0) iX0 = std::max(iX0, qx);
      if (x == N - 1) iX1 = std::min(iX1, qx);
      if (y == 0) iY0 = std::max(iY0, qy);
      if (y == N - 1) iY1 = std::min(iY1, qy);
    }
  const float iw = iX1 - iX0, ih = iY1 - iY0, ow = oX1 - oX0, oh = oY1 - oY0;
  const double fx0 = (float)(new_w - 1) / iw, fy0 = (float)(new_h - 1) / ih;
  const double cx0 = -fx0 * iX0, cy0 = -fy0 * iY0;
  const double fx1 = (float)(new_w - 1) / ow, fy1 = (float)(new_h - 1) / oh;
  const double cx1 = -fx1 * oX0, cy1 = -fy1 * oY0;
  newK[0] = fx0 * (1 - alpha) + fx1 * alpha;
  newK[1] = fy0 * (1 - alpha) + fy1 * alpha;
  newK[2] = cx0 * (1 - alpha) + cx1 * alpha;
  newK[3] = cy0 * (1 - alpha) + cy1 * alpha;
}

// initUndistortRectifyMap(K, dist, Mat(), newK, size, CV_16SC2, map1, map2)  (src/CameraModel.cpp:90)
void init_undistort_maps(const double K[4], const double k[4], const double newK[4], int w, int h, int16_t* map1,
                         uint16_t* map2) {
  const double ir0 = 1.0 / newK[0], ir2 = -newK[2] / newK[0], ir4 = 1.0 / newK[1], ir5 = -newK[3] / newK[1];
  for (int i = 0; i < h; i++) {
    double _x = i * 0.0 + ir2;
    const double _y = i * ir4 + ir5, _w = 1.0;
    for (int j = 0; j < w; j++, _x += ir0) {
      const double ww = 1.0 / _w, x = _x * ww, y = _y * ww;
      const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
      const double kr = (1 + ((0.0 * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2);
      const double xd = (x * kr + k[2] * _2xy + k[3] * (r2 + 2 * x2));
      const double yd = (y * kr + k[2] * (r2 + 2 * y2) + k[3] * _2xy);
      const double u = K[0] * xd + K[2], v = K[1] * yd + K[3];
      const long iu = std::lrint(u * 32.0), iv = std::lrint(v * 32.0);
      const int su = std::max(-32768, std::min(32767, (int)iu >> 5)), sv = std::max(-32768, std::min(32767, (int)iv >> 5));
      map1[2 * ((size_t)i * w + j)] = (int16_t)su;
      map1[2 * ((size_t)i * w + j) + 1] = (int16_t)sv;
      map2[(size_t)i * w + j] = (uint16_t)(((int)iv & 31) * 32 + ((int)iu & 31));
    }
  }
}

#define ING_CHK(expr)                              \
  do {                                             \
    if ((expr) != hipSuccess) return UWT_ERR_HIP;  \
  } while (0)

int ingest_remap(uwt_ingest* g, const uint8_t* raw, size_t stride, int x0, int y0, int cw, int ch, uint8_t* d_dst, int dst_pitch) {
  ING_CHK(hipSetDevice(g->device));
  if (stride == (size_t)g->in_w)   // tight rows: one linear copy (a 2-D copy is issued row by row)
    ING_CHK(hipMemcpyAsync(g->d_raw, raw, (size_t)g->in_w * g->in_h, hipMemcpyHostToDevice, g->stream));
  else
    ING_CHK(hipMemcpy2DAsync(g->d_raw, g->in_w, raw, stride, g->in_w, g->in_h, hipMemcpyHostToDevice, g->stream));
  hipLaunchKernelGGL(k_remap_crop, dim3((cw * ch + kBlock - 1) / kBlock), dim3(kBlock), 0, g->stream, g->d_raw, g->in_w,
                     g->in_h, (size_t)g->in_w, g->d_map1, g->d_map2, g->out_w, x0, y0, d_dst, cw, ch, dst_pitch);
  ING_CHK(hipGetLastError());
  return UWT_OK;
}

}  // namespace

int uwt_ingest_create(const float K[4], const float dist[4], int32_t in_w, int32_t in_h, int32_t out_w, int32_t out_h,
                      int32_t device, uwt_ingest** out, float newK_out[4]) {
  if (!K || !dist || !out || in_w < 2 || in_h < 2 || out_w < 1 || out_h < 1) return UWT_ERR_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return UWT_ERR_NO_DEVICE;
  uwt_ingest* g = new (std::nothrow) uwt_ingest();
  if (!g) return UWT_ERR_CAPACITY;
  g->in_w = in_w; g->in_h = in_h; g->out_w = out_w; g->out_h = out_h; g->device = device;
  const double Kd[4] = {K[0], K[1], K[2], K[3]}, kd[4] = {dist[0], dist[1], dist[2], dist[3]};
  optimal_new_camera_matrix(Kd, kd, in_w, in_h, 1.0, out_w, out_h, g->newK);
  g->h_map1.resize((size_t)out_w * out_h * 2);
  g->h_map2.resize((size_t)out_w * out_h);
  init_undistort_maps(Kd, kd, g->newK, out_w, out_h, g->h_map1.data(), g->h_map2.data());
  if (newK_out)
    for (int i = 0; i < 4; i++) newK_out[i] = (float)g->newK[i];
  bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess &&
            hipMalloc((void**)&g->d_map1, g->h_map1.size() * 2) == hipSuccess &&
            hipMalloc((void**)&g->d_map2, g->h_map2.size() * 2) == hipSuccess &&
            hipMalloc((void**)&g->d_raw, (size_t)in_w * in_h) == hipSuccess &&
            hipMalloc((void**)&g->d_und, (size_t)out_w * out_h) == hipSuccess &&
            hipMemcpy(g->d_map1, g->h_map1.data(), g->h_map1.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(g->d_map2, g->h_map2.data(), g->h_map2.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    uwt_ingest_destroy(g);
    return UWT_ERR_HIP;
  }
  *out = g;
  return UWT_OK;
}

int uwt_ingest_destroy(uwt_ingest* g) {
  if (!g) return UWT_ERR_INVALID_ARG;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->d_map1) (void)hipFree(g->d_map1);
  if (g->d_map2) (void)hipFree(g->d_map2);
  if (g->d_raw) (void)hipFree(g->d_raw);
  if (g->d_und) (void)hipFree(g->d_und);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
  return UWT_OK;
}

int uwt_ingest_maps(uwt_ingest* g, int16_t* map1_out, uint16_t* map2_out) {
  if (!g || !map1_out || !map2_out) return UWT_ERR_INVALID_ARG;
  std::memcpy(map1_out, g->h_map1.data(), g->h_map1.size() * 2);
  std::memcpy(map2_out, g->h_map2.data(), g->h_map2.size() * 2);
  return UWT_OK;
}

int uwt_ingest_undistort(uwt_ingest* g, const uint8_t* raw, size_t stride, uint8_t* und_out) {
  if (!g || !raw || !und_out || stride < (size_t)g->in_w) return UWT_ERR_INVALID_ARG;
  int st = ingest_remap(g, raw, stride, 0, 0, g->out_w, g->out_h, g->d_und, g->out_w);
  if (st) return st;
  ING_CHK(hipMemcpyAsync(und_out, g->d_und, (size_t)g->out_w * g->out_h, hipMemcpyDeviceToHost, g->stream));
  ING_CHK(hipStreamSynchronize(g->stream));
  return UWT_OK;
}

int uwt_ingest_calculate_roi(uwt_ingest* g, const uint8_t* raw_first, size_t stride, int32_t roi[4]) {
  if (!g || !roi) return UWT_ERR_INVALID_ARG;
  std::vector<uint8_t> und((size_t)g->out_w * g->out_h);
  int st = uwt_ingest_undistort(g, raw_first, stride, und.data());
  if (st) return st;
  // System::CalculateROI (src/System.cpp:148-191): walk in from the four sides along the middle row / column while the
  // undistorted image is 0, then a 5-pixel margin; Rect(p1, p2) => width = p2.x - p1.x.
  const int w = g->out_w, h = g->out_h;
  const int xm = (int)((w - 1) * 0.5), ym = (int)((h - 1) * 0.5);
  int p1x = 0, p1y = 0, p2x = w - 1, p2y = h - 1;
  while (p1x < w - 1 && und[(size_t)ym * w + p1x] == 0) p1x++;
  while (p2x > 0 && und[(size_t)ym * w + p2x] == 0) p2x--;
  while (p1y < h - 1 && und[(size_t)p1y * w + xm] == 0) p1y++;
  while (p2y > 0 && und[(size_t)p2y * w + xm] == 0) p2y--;
  p1x += 5; p2x -= 5; p1y += 5; p2y -= 5;
  roi[0] = p1x; roi[1] = p1y; roi[2] = p2x - p1x; roi[3] = p2y - p1y;
  return UWT_OK;
}

int uwt_ingest_frame(uwt_ingest* g, uwt_ctx* c, int32_t slot, const uint8_t* raw, size_t stride, int32_t x0, int32_t y0) {
  if (!g || !c || !raw || stride < (size_t)g->in_w || !slot_range_ok(c, slot, 1)) return UWT_ERR_INVALID_ARG;
  const int cw = c->p.width, ch = c->p.height;
  if (x0 < 0 || y0 < 0 || x0 + cw > g->out_w || y0 + ch > g->out_h || g->device != c->p.device)
    return fail(c, UWT_ERR_INVALID_ARG, "uwt_ingest_frame: crop window outside the undistorted frame");
  // The remap runs on the ingest object's stream and writes straight into the tracker's slot: it is ordered behind the
  // tracker work still in flight on that slot (uwt_track_batch_async returns with its kernels queued), and this call
  // returns only when the slot is written, so whatever the tracker enqueues next sees the new frame.
  int st = dep_wait(c, c->busy, c->busy_next, c->busy_dropped, g->stream, slot, 1);
  if (st) return st;
  st = dep_wait(c, c->fresh, c->fresh_next, c->fresh_dropped, g->stream, slot, 1);
  if (st) return st;
  st = ingest_remap(g, raw, stride, x0, y0, cw, ch, c->img[0] + (size_t)slot * c->lv[0].n, c->lv[0].pitch);
  if (st) return st;
  ING_CHK(hipStreamSynchronize(g->stream));
  return UWT_OK;
}

static int accumulate_trajectory_impl(uwt_ctx* c, bool scan, const float* poses, int32_t n, const float start_pose[7], float t_scale,
                              int32_t reference_axes, float* traj_out) {
  if (c) (void)hipSetDevice(c->p.device);  // one context = one device; callers may have switched the thread's device
  if (!c || !poses || !start_pose || !traj_out || n < 0) return fail(c, UWT_ERR_INVALID_ARG, "uwt_accumulate_trajectory");
  if (n == 0) return UWT_OK;
  const size_t bytes = sizeof(float) * 7 * (size_t)n;
  int st = ensure_scratch(c, 2 * bytes);
  if (st) return st;
  float* din = (float*)c->scratch;
  float* dout = din + 7 * (size_t)n;
  HIPCHK(c, hipMemcpyAsync(din, poses, bytes, hipMemcpyHostToDevice, c->stream));
  Pose P;
  for (int k = 0; k < 4; k++) P.q[k] = start_pose[k];
  for (int k = 0; k < 3; k++) P.t[k] = start_pose[4 + k];
  if (scan) hipLaunchKernelGGL(k_trajectory_scan, dim3(1), dim3(1024), 0, c->stream, din, n, P, t_scale, reference_axes, dout);
  else hipLaunchKernelGGL(k_trajectory, dim3(1), dim3(64), 0, c->stream, din, n, P, t_scale, reference_axes, dout);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(traj_out, dout, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return UWT_OK;
}


int uwt_accumulate_trajectory(uwt_ctx* c, const float* poses, int32_t n, const float start_pose[7], float t_scale,
                              int32_t reference_axes, float* traj_out) {
  return accumulate_trajectory_impl(c, false, poses, n, start_pose, t_scale, reference_axes, traj_out);
}

int uwt_accumulate_trajectory_scan(uwt_ctx* c, const float* poses, int32_t n, const float start_pose[7], float t_scale,
                                   int32_t reference_axes, float* traj_out) {
  return accumulate_trajectory_impl(c, true, poses, n, start_pose, t_scale, reference_axes, traj_out);
}

}  // extern "C"
