// uwt_launch.h — internal: the launch dispatchers of the heavy kernel templates, one translation unit per family so that the
// library builds in parallel (uwt_launch_residual.hip, uwt_launch_general.hip, uwt_launch_flow.hip).
// A dispatcher picks the instantiation from run-time facts (arithmetic set, depth plane, level width, intrinsics, factors) and
// enqueues it; it reports nothing — the caller checks hipGetLastError().
#pragma once

#include "uwt_kernels.h"

namespace uwt {

struct LaunchSel {
  int arith;           // kArithOpenCV / kArithLegacy (uwt_params::arith)
  bool depth;          // the context has a depth plane
  bool acc64;          // f64 normal-equation sums (uwt_params::accumulate_f64)
  bool compute_only;   // the diagnostic twin (uwt_profile_enable bit 2)
};

// A level's rows are pitched to whole groups of four pixels and walked in such groups.  level_ragged: the grid's rows are not
// whole groups (46 wide in rows of 48; every level of an odd-sized frame): the RAGGED instantiations mask the positions of a
// row's last group that lie beyond the point grid (residual_core: colm).
inline bool level_ragged(const LevelK& L) { return L.gw != L.pitch; }
inline bool level_plain(const ResidualArgs& a) { return a.zf == 1.0f && a.af == 1.0f && a.L.fx == a.L.fy; }

// k_residual, identity weights / nearest sampler (and the per-stage dump form)
void launch_residual(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, int n_pairs, bool dump);
// the scale pass (weights != 0: k_resid_hist_v) and the weighted / bilinear k_residual of one evaluation
void launch_general(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, int n_pairs, int sampler, int weights,
                    unsigned int* hist, PairScale* scale);
// the per-stage (dump-capable) form: k_resid_hist + k_scale_stage (weights != 0), then k_residual_general
void launch_general_dump(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const GeneralArgs& ga, int n_pairs);
// explicit point tables (k_residual_points)
void launch_points(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const PointsArgs& pa);
// the same on the general path: k_points_hist + k_scale_stage (weights != 0; the caller cleared the pair's bins), k_points_general
void launch_points_general(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const PointsArgs& pa, const GeneralArgs& ga);

// the chained flow of a few pairs: k_iterate, k_coarse (up to kCoarseMaxLevels levels in one launch), k_finish
void launch_iterate(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const IterArgs& ia, int n_pairs);
void launch_coarse_chain(hipStream_t s, const LaunchSel& sel, const CoarseArgs& ca, int n_pairs);
void launch_finish(hipStream_t s, const IterArgs& ia, int n_pairs, float* d_poses, StatsOut* d_stats);
// the chained flow under robust weights: the pending update + the scale pass in one launch (k_hist_iterate); the weighted pass
// behind it is launch_general's second half (launch_weighted)
void launch_hist_iterate(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const IterArgs& ia, int n_pairs, int sampler,
                         int weights, unsigned int* hist, PairScale* scale);
void launch_weighted(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, int n_pairs, int sampler, int weights);
// one coarse level of a batch, one block per pair: k_coarse_w4 (identity weights) / k_coarse_weighted
void launch_coarse_level(hipStream_t s, const LaunchSel& sel, const CoarseArgs& ca, int cnt, int weights);

// runs the statements with AR a compile-time constant
#define UWT_WITH_AR(arith, ...)                                              \
  do {                                                                       \
    if ((arith) == kArithLegacy) { constexpr int AR = kArithLegacy; __VA_ARGS__; }  \
    else { constexpr int AR = kArithOpenCV; __VA_ARGS__; }                   \
  } while (0)

}  // namespace uwt
