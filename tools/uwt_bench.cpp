// Native benchmark of the direct-tracking path over the C ABI alone (no Python, no torch in the process):
// deterministic synthetic pairs made here, resident batch, uwt_track_batch_async + uwt_sync per step, one JSON line.
//
//   make -C tools            (g++ against include/uwt.h and uw-slam_amd/libuwt_hip.so; see tools/Makefile)
//   tools/uwt_bench [--pairs 1024] [--unique 32] [--steps 10] [--warmup 3] [--width 640] [--height 480] [--levels 4]
//                   [--iters 10] [--no-depth] [--reference-schedule] [--legacy-arith] [--gpus N] [--rccl]
//
// --gpus N: the batched multi-GPU mode from a native host — one thread and one context per device, --pairs per device,
// global pair i on device i mod N, one ncclAllGather (RCCL over xGMI) of the solved poses per step enqueued on each
// context's stream, the gathered block checked on the host.  --rccl takes that path with one device too.
//
// Synthetic inputs follow SURVEY.md §8(d): a band-limited texture (seeded noise, Gaussian blur sigma = 3 px, min-max to
// u8), the target re-rendered under a small random SE(3) for a fronto-parallel plane at depth z, TUM-like intrinsics.
// bench.py is the contract benchmark (its generator is numpy; the two are not bit-identical inputs); this program shows
// the library needs nothing but its header.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>   // hipMalloc / hipMemcpy / hipSetDevice for the pose buffers (no device code here)
#include <rccl/rccl.h>

#include "uwt.h"
#include "uwt_gen.h"   // the deterministic input generator shared with bench.py (tools/libuwt_gen.so)

namespace {

int arg_int(int argc, char** argv, const char* name, int def) {
  for (int i = 1; i + 1 < argc; i++)
    if (!std::strcmp(argv[i], name)) return std::atoi(argv[i + 1]);
  return def;
}
bool arg_flag(int argc, char** argv, const char* name) {
  for (int i = 1; i < argc; i++)
    if (!std::strcmp(argv[i], name)) return true;
  return false;
}

#define CHK(expr)                                                                      \
  do {                                                                                 \
    const int st_ = (expr);                                                            \
    if (st_ != UWT_OK) {                                                               \
      std::fprintf(stderr, "%s -> %d (%s)\n", #expr, st_, ctx ? uwt_last_error(ctx) : ""); \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

struct Barrier {   // std::barrier is C++20
  std::mutex m;
  std::condition_variable cv;
  int n, waiting = 0, phase = 0;
  bool aborted = false;
  explicit Barrier(int n_) : n(n_) {}
  // false: a device thread failed and called abort() — every waiter (now or later) gives up instead of waiting for it
  bool wait() {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) return false;
    const int ph = phase;
    if (++waiting == n) { waiting = 0; phase++; cv.notify_all(); }
    else cv.wait(lk, [&] { return phase != ph || aborted; });
    return !aborted;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(m);
    aborted = true;
    cv.notify_all();
  }
};

struct Shared {
  int P, U, steps, warmup, w, h, levels, iters, n_dev;
  bool depth, ref_sched, rccl, legacy;
  Barrier* bar;
  ncclComm_t* comms;
  double t_start = 0, t_end = 0;
  std::mutex m;
  std::vector<std::vector<float>> gathered;   // per device: the n_dev * P * 7 floats it received
  std::vector<std::vector<float>> own;        // per device: its own poses
  std::vector<int> rc;
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// one device: its shard of the batch (global pair i = dev + k * n_dev), resident; steps; optional all-gather per step
int run_device(Shared& S, int dev) {
  const int P = S.P, w = S.w, h = S.h;
  const float f = 525.0f * w / 640.0f, cx = w / 2 - 0.5f, cy = h / 2 - 0.5f;
  if (hipSetDevice(dev) != hipSuccess) return 1;
  uwt_ctx* ctx = nullptr;
  uwt_params p;
  uwt_default_params(&p, w, h, f, f, cx, cy);
  if (S.ref_sched) {
    p.n_levels = 5; p.first_level = 4; p.last_level = 1; p.max_iters = 50; p.early_exit = 1;
  } else {
    p.n_levels = S.levels; p.first_level = S.levels - 1; p.last_level = 0; p.max_iters = S.iters; p.early_exit = 0;
  }
  p.has_depth = S.depth ? 1 : 0;
  p.arith = S.legacy ? UWT_ARITH_LEGACY : UWT_ARITH_OPENCV;   // (the default: OpenCV's generic code paths, include/uwt.h)
  p.max_frames = 2 * P;
  p.max_pairs = P;
  p.device = dev;
  CHK(uwt_create(&p, &ctx));

  // U distinct pairs of this device's shard, tiled over its P resident pairs (slot 2i = reference, 2i+1 = target)
  const int U = S.U;
  std::vector<std::vector<uint8_t>> refs(U), tgts(U);
  std::vector<std::vector<uint16_t>> deps(U);
  for (int u = 0; u < U; u++) {
    const int gid = dev + u * S.n_dev;   // global id of the shard's u-th pair
    uwt_gen::gen_pair(w, h, f, f, cx, cy, gid, S.depth, refs[u], tgts[u], deps[u]);
  }
  for (int i = 0; i < P; i++) {
    const int u = i % U;
    CHK(uwt_set_frame(ctx, 2 * i, refs[u].data(), (size_t)w, S.depth ? deps[u].data() : nullptr, S.depth ? (size_t)w * 2 : 0));
    CHK(uwt_set_frame(ctx, 2 * i + 1, tgts[u].data(), (size_t)w, S.depth ? deps[u].data() : nullptr, S.depth ? (size_t)w * 2 : 0));
  }
  std::vector<int32_t> ref_slots(P), tgt_slots(P);
  for (int i = 0; i < P; i++) { ref_slots[i] = 2 * i; tgt_slots[i] = 2 * i + 1; }
  float *d_poses = nullptr, *d_all = nullptr;
  if (hipMalloc((void**)&d_poses, sizeof(float) * 7 * P) != hipSuccess) return 1;
  if (S.rccl && hipMalloc((void**)&d_all, sizeof(float) * 7 * P * S.n_dev) != hipSuccess) return 1;
  void* stream = nullptr;
  CHK(uwt_stream(ctx, &stream));

  auto step = [&]() -> int {
    CHK(uwt_track_batch_async(ctx, 0, 2 * P, 1, P, ref_slots.data(), tgt_slots.data(), d_poses, nullptr));
    // the gather follows the alignment on the context's stream: no host synchronisation per step
    if (S.rccl && ncclAllGather(d_poses, d_all, (size_t)7 * P, ncclFloat, S.comms[dev], (hipStream_t)stream) != ncclSuccess) return 1;
    return 0;
  };
  for (int i = 0; i < S.warmup; i++)
    if (step()) return 1;
  CHK(uwt_sync(ctx));
  if (!S.bar->wait()) return 1;
  if (dev == 0) S.t_start = now_s();
  for (int i = 0; i < S.steps; i++)
    if (step()) return 1;
  CHK(uwt_sync(ctx));
  if (!S.bar->wait()) return 1;       // the job's time is that of its slowest device
  if (dev == 0) S.t_end = now_s();

  S.own[dev].resize((size_t)7 * P);
  if (hipMemcpy(S.own[dev].data(), d_poses, sizeof(float) * 7 * P, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (S.rccl) {
    S.gathered[dev].resize((size_t)7 * P * S.n_dev);
    if (hipMemcpy(S.gathered[dev].data(), d_all, sizeof(float) * 7 * P * S.n_dev, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    (void)hipFree(d_all);
  }
  (void)hipFree(d_poses);
  uwt_destroy(ctx);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  // more hardware queues than the runtime's default of 4: with RCCL's streams in the process the two halves of a batch can
  // otherwise be dealt one queue and run one after the other (bench.py has the measurements); read once, at start-up
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  Shared S;
  S.P = arg_int(argc, argv, "--pairs", 1024);
  const int U0 = arg_int(argc, argv, "--unique", 32);
  S.U = U0 < S.P ? U0 : S.P;
  S.steps = arg_int(argc, argv, "--steps", 10);
  S.warmup = arg_int(argc, argv, "--warmup", 3);
  S.w = arg_int(argc, argv, "--width", 640);
  S.h = arg_int(argc, argv, "--height", 480);
  S.levels = arg_int(argc, argv, "--levels", 4);
  S.iters = arg_int(argc, argv, "--iters", 10);
  S.depth = !arg_flag(argc, argv, "--no-depth");
  S.ref_sched = arg_flag(argc, argv, "--reference-schedule");
  S.legacy = arg_flag(argc, argv, "--legacy-arith");
  S.n_dev = arg_int(argc, argv, "--gpus", 1);
  S.rccl = S.n_dev > 1 || arg_flag(argc, argv, "--rccl");
  if (S.ref_sched) S.levels = 5;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have < S.n_dev || S.n_dev < 1) {
    std::fprintf(stderr, "uwt_bench: --gpus %d but %d device(s) visible\n", S.n_dev, have);
    return 2;
  }
  std::vector<ncclComm_t> comms(S.n_dev);
  if (S.rccl) {
    std::vector<int> devs(S.n_dev);
    for (int d = 0; d < S.n_dev; d++) devs[d] = d;
    if (ncclCommInitAll(comms.data(), S.n_dev, devs.data()) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
  }
  S.comms = comms.data();
  Barrier bar(S.n_dev);
  S.bar = &bar;
  S.gathered.resize(S.n_dev);
  S.own.resize(S.n_dev);
  S.rc.assign(S.n_dev, 0);
  std::vector<std::thread> th;
  // a failing device thread aborts the barrier, so the others return instead of waiting for it; a peer stuck inside a
  // collective the failed device never joins cannot be woken that way — after a grace period the process leaves through
  // _Exit (no destructors racing live HIP / RCCL state)
  std::atomic<int> done{0};
  auto body = [&S, &done](int d) {
    S.rc[d] = run_device(S, d);
    if (S.rc[d]) S.bar->abort();
    done++;
  };
  for (int d = 1; d < S.n_dev; d++) th.emplace_back(body, d);
  body(0);
  bool failed = false;
  for (int d = 0; d < S.n_dev; d++) failed |= S.rc[d] != 0;
  if (failed) {
    for (int w = 0; w < 100 && done.load() < S.n_dev; w++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    if (done.load() < S.n_dev) { std::fprintf(stderr, "a device failed and a peer did not return: leaving\n"); std::_Exit(1); }
  }
  for (auto& t : th) t.join();
  for (int d = 0; d < S.n_dev; d++)
    if (S.rc[d]) return 1;
  if (S.rccl)
    for (int d = 0; d < S.n_dev; d++) ncclCommDestroy(comms[d]);

  const int P = S.P, U = S.U, N = S.n_dev;
  const double sec = S.t_end - S.t_start;
  double tmax = 0;
  int finite = 1, tiled_equal = 1, gather_ok = 1;
  for (int d = 0; d < N; d++) {
    const std::vector<float>& poses = S.own[d];
    for (int i = 0; i < P; i++) {
      const float* q = &poses[7 * i];
      for (int k = 0; k < 7; k++) finite &= std::isfinite(q[k]) ? 1 : 0;
      tmax = std::fmax(tmax, std::sqrt((double)q[4] * q[4] + (double)q[5] * q[5] + (double)q[6] * q[6]));
    }
    // tiled pairs must give identical poses (same inputs, same arithmetic)
    for (int i = U; i < P; i++) tiled_equal &= !std::memcmp(&poses[7 * i], &poses[7 * (i % U)], sizeof(float) * 7);
    // every device received every device's block (rank-major: block r = device r's pairs r, r + N, r + 2N, ...)
    if (S.rccl)
      for (int r = 0; r < N; r++) gather_ok &= !std::memcmp(&S.gathered[d][(size_t)7 * P * r], S.own[r].data(), sizeof(float) * 7 * P);
  }
  std::printf("{\"metric\": \"frame-pair alignments/sec (%dx%d, %d pyr lvls)\", \"value\": %.2f, \"unit\": \"alignments/s\", "
              "\"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"scaling\": \"weak\", \"host\": \"C++ over the C ABI%s\", "
              "\"config\": {\"workload\": \"%s, %d pairs resident per GPU (%d distinct)%s\", \"arithmetic\": \"%s\"}, "
              "\"poses_finite\": %s, \"tiled_pairs_identical\": %s, \"gathered_blocks_match\": %s, \"max_translation_m\": %.6f}\n",
              S.w, S.h, S.levels, (double)P * N * S.steps / sec, N, S.steps, S.warmup, sec * 1e3 / S.steps,
              S.rccl ? " + RCCL all-gather" : "",
              S.ref_sched ? "reference schedule (levels 4..1, <= 50 iterations, early exit)" : "fixed iterations, no early exit", P, U,
              S.depth ? ", u16 depth plane" : "", S.legacy ? "legacy" : "opencv", finite ? "true" : "false", tiled_equal ? "true" : "false",
              !S.rccl ? "null" : (gather_ok ? "true" : "false"), tmax);
  return (finite && tiled_equal && gather_ok) ? 0 : 1;
}
