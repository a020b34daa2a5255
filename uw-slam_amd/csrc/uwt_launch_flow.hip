// uwt_launch_flow.hip — dispatch of the flow kernels: k_iterate / k_coarse / k_finish (a few pairs per call: the update chained
// into the next evaluation's launch) and k_coarse_w4 / k_coarse_weighted (a coarse level of a batch, one block per pair).
// f64 sums only; PLAIN = square pixels and unit factors (see k_iterate).
#include "uwt_launch.h"

namespace uwt {
namespace {

template <int AR, bool DEPTH>
void launch_iterate_t(hipStream_t s, const ResidualArgs& a, const IterArgs& ia, int n_pairs, bool compute_only) {
  const dim3 grid(a.slices, n_pairs), blk(kBlock);
  const bool plain = level_plain(a);
  const bool wide = n_pairs > 3;   // four blocks per CU (two-pass reduction) instead of blocks alone on their CUs
  if (level_ragged(a.L)) {   // grid rows that are not whole groups of four: the masking instantiations (square pixels with unit factors or the general form)
    if (plain) hipLaunchKernelGGL((k_iterate<AR, 4, DEPTH, true, false, 14, true>), grid, blk, 0, s, a, ia);
    else hipLaunchKernelGGL((k_iterate<AR, 4, DEPTH, false, false, 14, true>), grid, blk, 0, s, a, ia);
    return;
  }
  if (plain && compute_only) hipLaunchKernelGGL((k_iterate<AR, 4, DEPTH, true, true>), grid, blk, 0, s, a, ia);
  else if (plain && wide) hipLaunchKernelGGL((k_iterate<AR, 4, DEPTH, true, false, 14>), grid, blk, 0, s, a, ia);
  else if (plain) hipLaunchKernelGGL((k_iterate<AR, 4, DEPTH, true>), grid, blk, 0, s, a, ia);
  else if (wide) hipLaunchKernelGGL((k_iterate<AR, 4, DEPTH, false, false, 14>), grid, blk, 0, s, a, ia);
  else hipLaunchKernelGGL((k_iterate<AR, 4, DEPTH, false>), grid, blk, 0, s, a, ia);
}

template <int AR, bool DEPTH>
void launch_coarse_level_t(hipStream_t s, const CoarseArgs& ca, int cnt, int weights) {
  const dim3 grid(cnt), blk(kBlock);
  const bool v4 = !level_ragged(ca.lv[0].L);
  const bool plain = v4 && level_plain(ca.lv[0]);   // (a level with ragged grid rows takes the general form: one instantiation)
  if (weights == kWeightsTukeyRef) {
    if (plain) hipLaunchKernelGGL((k_coarse_weighted<AR, DEPTH, kWeightsTukeyRef, true, 1>), grid, blk, 0, s, ca);
    else if (v4) hipLaunchKernelGGL((k_coarse_weighted<AR, DEPTH, kWeightsTukeyRef, false, 1>), grid, blk, 0, s, ca);
    else hipLaunchKernelGGL((k_coarse_weighted<AR, DEPTH, kWeightsTukeyRef, false, 1, true>), grid, blk, 0, s, ca);
  } else if (weights == kWeightsHuber) {
    if (plain) hipLaunchKernelGGL((k_coarse_weighted<AR, DEPTH, kWeightsHuber, true, 1>), grid, blk, 0, s, ca);
    else if (v4) hipLaunchKernelGGL((k_coarse_weighted<AR, DEPTH, kWeightsHuber, false, 1>), grid, blk, 0, s, ca);
    else hipLaunchKernelGGL((k_coarse_weighted<AR, DEPTH, kWeightsHuber, false, 1, true>), grid, blk, 0, s, ca);
  } else {
    if (plain) hipLaunchKernelGGL((k_coarse_w4<AR, DEPTH, true>), grid, blk, 0, s, ca);
    else if (v4) hipLaunchKernelGGL((k_coarse_w4<AR, DEPTH, false>), grid, blk, 0, s, ca);
    else hipLaunchKernelGGL((k_coarse_w4<AR, DEPTH, false, 1>), grid, blk, 0, s, ca);
  }
}

}  // namespace

void launch_iterate(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const IterArgs& ia, int n_pairs) {
  UWT_WITH_AR(sel.arith,
    if (sel.depth) launch_iterate_t<AR, true>(s, a, ia, n_pairs, sel.compute_only);
    else launch_iterate_t<AR, false>(s, a, ia, n_pairs, sel.compute_only));
}

// up to kCoarseMaxLevels coarsest levels of a few pairs, to their end, in one launch; when one of them has ragged grid rows
// (not whole groups of four) the launch takes the general form with the masking decided per level
void launch_coarse_chain(hipStream_t s, const LaunchSel& sel, const CoarseArgs& ca, int n_pairs) {
  const dim3 grid(n_pairs), blk(kBlock);
  bool all4 = true;
  for (int i = 0; i < ca.n_levels; i++) all4 = all4 && !level_ragged(ca.lv[i].L);
  const bool plain = all4 && level_plain(ca.lv[0]);
  UWT_WITH_AR(sel.arith,
    if (sel.depth) {
      if (plain) hipLaunchKernelGGL((k_coarse<AR, true, true>), grid, blk, 0, s, ca);
      else if (all4) hipLaunchKernelGGL((k_coarse<AR, true, false>), grid, blk, 0, s, ca);
      else hipLaunchKernelGGL((k_coarse<AR, true, false, kIteratePass, kCoarseMaxLevels, 0>), grid, blk, 0, s, ca);
    } else {
      if (plain) hipLaunchKernelGGL((k_coarse<AR, false, true>), grid, blk, 0, s, ca);
      else if (all4) hipLaunchKernelGGL((k_coarse<AR, false, false>), grid, blk, 0, s, ca);
      else hipLaunchKernelGGL((k_coarse<AR, false, false, kIteratePass, kCoarseMaxLevels, 0>), grid, blk, 0, s, ca);
    });
}

void launch_finish(hipStream_t s, const IterArgs& ia, int n_pairs, float* d_poses, StatsOut* d_stats) {
  hipLaunchKernelGGL(k_finish, dim3(n_pairs), dim3(kUpdateBlock), 0, s, ia, d_poses, d_stats);
}

// one coarse level (up to kCoarseMaxPixels) of a batch, one block per pair, the level's iterations in one
// launch: identity weights k_coarse_w4, robust weights over the nearest sampler k_coarse_weighted
void launch_coarse_level(hipStream_t s, const LaunchSel& sel, const CoarseArgs& ca, int cnt, int weights) {
  UWT_WITH_AR(sel.arith,
    if (sel.depth) launch_coarse_level_t<AR, true>(s, ca, cnt, weights);
    else launch_coarse_level_t<AR, false>(s, ca, cnt, weights));
}

}  // namespace uwt
