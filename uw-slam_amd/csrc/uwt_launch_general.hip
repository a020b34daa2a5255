// uwt_launch_general.hip — dispatch of the general path: robust weights (scale pass k_resid_hist_v + weighted k_residual) and the
// bilinear sampler, and the per-stage (dump-capable) form k_resid_hist / k_scale_stage / k_residual_general.
#include "uwt_launch.h"

namespace uwt {
namespace {

// One residual evaluation on the general path for pairs [pair_base, +n): the dense kernel specialised for the sampler / weights.
// Same slicing as the fast path.
template <int AR, bool RAGGED, bool DEPTH, bool UNIT>
void launch_general_t(hipStream_t s, const ResidualArgs& a, int n_pairs, int sampler, int weights) {
  const dim3 grid(a.slices, n_pairs), blk(kBlock);
  const int key = sampler * 3 + weights;
  constexpr int VEC = 4;
  if constexpr (!RAGGED && UNIT) {
    if (a.L.fx == a.L.fy) {   // SQUARE: the Jacobian's coinciding products once (pixel_jacobian), as on the identity path
      if (a.stream_planes) {   // the streamed twins (load_group)
        switch (key) {
          case 1: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, true, 0, 1, false, true>), grid, blk, 0, s, a); break;
          case 2: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, true, 0, 2, false, true>), grid, blk, 0, s, a); break;
          case 3: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, true, 1, 0, false, true>), grid, blk, 0, s, a); break;
          default:
            if constexpr (AR == kArithOpenCV) hipLaunchKernelGGL((k_residual_w4<AR, VEC, DEPTH, UNIT, false, double, true, 1, 2, true>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, true, 1, 2, false, true>), grid, blk, 0, s, a);
            break;
        }
        return;
      }
      switch (key) {
        case 1: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, true, 0, 1>), grid, blk, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, true, 0, 2>), grid, blk, 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, true, 1, 0>), grid, blk, 0, s, a); break;
        default:   // bilinear + Huber: 129 registers under the OpenCV set, held to 128 (one value parked in scratch outside the loop)
          if constexpr (AR == kArithOpenCV) hipLaunchKernelGGL((k_residual_w4<AR, VEC, DEPTH, UNIT, false, double, true, 1, 2>), grid, blk, 0, s, a);
          else hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, true, 1, 2>), grid, blk, 0, s, a);
          break;
      }
      return;
    }
  }
  switch (key) {   // the general Jacobian form; RAGGED: with the positions beyond the grid masked
    case 1: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, false, 0, 1, false, 0, RAGGED>), grid, blk, 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, false, 0, 2, false, 0, RAGGED>), grid, blk, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, false, 1, 0, false, 0, RAGGED>), grid, blk, 0, s, a); break;
    default: hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, false, double, false, 1, 2, false, 0, RAGGED>), grid, blk, 0, s, a); break;
  }
}

template <int AR, bool RAGGED, bool DEPTH>
void launch_hist_t(hipStream_t s, const ResidualArgs& a, int n_pairs, int sampler, int weights, unsigned int* hist, PairScale* scale) {
  const dim3 grid(a.slices, n_pairs), blk(kBlock);
  if (sampler) hipLaunchKernelGGL((k_resid_hist_v<AR, 4, DEPTH, 1, RAGGED>), grid, blk, 0, s, a, hist, scale, weights);
  else hipLaunchKernelGGL((k_resid_hist_v<AR, 4, DEPTH, 0, RAGGED>), grid, blk, 0, s, a, hist, scale, weights);
}

}  // namespace

// The alignment loop's launch on the general path: the scale pass (weights only: residual histograms per pair, the scale derived
// in the tail of the pair's last block; the histograms are all-zero before and after — cleared once per alignment call), then
// the dense kernel specialised for the sampler / weights.
void launch_general(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, int n_pairs, int sampler, int weights,
                    unsigned int* hist, PairScale* scale) {
  const bool ragged = level_ragged(a.L);
  if (weights) {
    UWT_WITH_AR(sel.arith,
      if (!ragged) {
        if (sel.depth) launch_hist_t<AR, false, true>(s, a, n_pairs, sampler, weights, hist, scale);
        else launch_hist_t<AR, false, false>(s, a, n_pairs, sampler, weights, hist, scale);
      } else {
        if (sel.depth) launch_hist_t<AR, true, true>(s, a, n_pairs, sampler, weights, hist, scale);
        else launch_hist_t<AR, true, false>(s, a, n_pairs, sampler, weights, hist, scale);
      });
  }
  launch_weighted(s, sel, a, n_pairs, sampler, weights);
}

// the weighted / bilinear sums of one evaluation (the scale, where weights are on, is in place)
void launch_weighted(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, int n_pairs, int sampler, int weights) {
  const bool unit = (a.zf == 1.0f && a.af == 1.0f);
  const bool ragged = level_ragged(a.L);
  const int key = (ragged ? 0 : 4) | (sel.depth ? 2 : 0) | (unit ? 1 : 0);
  UWT_WITH_AR(sel.arith,
    switch (key) {
      case 0: launch_general_t<AR, true, false, false>(s, a, n_pairs, sampler, weights); break;
      case 1: launch_general_t<AR, true, false, true>(s, a, n_pairs, sampler, weights); break;
      case 2: launch_general_t<AR, true, true, false>(s, a, n_pairs, sampler, weights); break;
      case 3: launch_general_t<AR, true, true, true>(s, a, n_pairs, sampler, weights); break;
      case 4: launch_general_t<AR, false, false, false>(s, a, n_pairs, sampler, weights); break;
      case 5: launch_general_t<AR, false, false, true>(s, a, n_pairs, sampler, weights); break;
      case 6: launch_general_t<AR, false, true, false>(s, a, n_pairs, sampler, weights); break;
      default: launch_general_t<AR, false, true, true>(s, a, n_pairs, sampler, weights); break;
    });
}

// the chained flow's first launch of an evaluation under robust weights: pending update + scale pass (k_hist_iterate)
void launch_hist_iterate(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const IterArgs& ia, int n_pairs, int sampler,
                         int weights, unsigned int* hist, PairScale* scale) {
  const dim3 grid(a.slices, n_pairs), blk(kBlock);
  const bool ragged = level_ragged(a.L);
  const int key = (sampler ? 4 : 0) | (sel.depth ? 2 : 0) | (ragged ? 1 : 0);
  UWT_WITH_AR(sel.arith,
    switch (key) {
      case 0: hipLaunchKernelGGL((k_hist_iterate<AR, false, 0, false>), grid, blk, 0, s, a, ia, hist, scale, weights); break;
      case 1: hipLaunchKernelGGL((k_hist_iterate<AR, false, 0, true>), grid, blk, 0, s, a, ia, hist, scale, weights); break;
      case 2: hipLaunchKernelGGL((k_hist_iterate<AR, true, 0, false>), grid, blk, 0, s, a, ia, hist, scale, weights); break;
      case 3: hipLaunchKernelGGL((k_hist_iterate<AR, true, 0, true>), grid, blk, 0, s, a, ia, hist, scale, weights); break;
      case 4: hipLaunchKernelGGL((k_hist_iterate<AR, false, 1, false>), grid, blk, 0, s, a, ia, hist, scale, weights); break;
      case 5: hipLaunchKernelGGL((k_hist_iterate<AR, false, 1, true>), grid, blk, 0, s, a, ia, hist, scale, weights); break;
      case 6: hipLaunchKernelGGL((k_hist_iterate<AR, true, 1, false>), grid, blk, 0, s, a, ia, hist, scale, weights); break;
      default: hipLaunchKernelGGL((k_hist_iterate<AR, true, 1, true>), grid, blk, 0, s, a, ia, hist, scale, weights); break;
    });
}

// The per-stage (dump-capable) form of the same evaluation: k_residual_general, one pixel per thread step; `a` carries the
// dump form's slicing (one pixel per point), `ga.hist` is cleared by the caller.
void launch_general_dump(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const GeneralArgs& ga, int n_pairs) {
  const bool unit = (a.zf == 1.0f && a.af == 1.0f);
  const dim3 grid(a.slices, n_pairs), blk(kBlock);
  if (ga.weights) {
    UWT_WITH_AR(sel.arith,
      if (sel.depth) hipLaunchKernelGGL((k_resid_hist<AR, true>), grid, blk, 0, s, a, ga);
      else hipLaunchKernelGGL((k_resid_hist<AR, false>), grid, blk, 0, s, a, ga));
    hipLaunchKernelGGL(k_scale_stage, dim3((n_pairs + 3) / 4), dim3(256), 0, s, ga, a.state, n_pairs, a.pair_base);
  }
  UWT_WITH_AR(sel.arith,
    if (sel.depth && unit) hipLaunchKernelGGL((k_residual_general<AR, true, true>), grid, blk, 0, s, a, ga);
    else if (sel.depth) hipLaunchKernelGGL((k_residual_general<AR, true, false>), grid, blk, 0, s, a, ga);
    else if (unit) hipLaunchKernelGGL((k_residual_general<AR, false, true>), grid, blk, 0, s, a, ga);
    else hipLaunchKernelGGL((k_residual_general<AR, false, false>), grid, blk, 0, s, a, ga));
}

void launch_points_general(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const PointsArgs& pa, const GeneralArgs& ga) {
  const bool unit = (a.zf == 1.0f && a.af == 1.0f);
  const dim3 grid(a.slices), blk(kBlock);
  if (ga.weights) {
    UWT_WITH_AR(sel.arith, hipLaunchKernelGGL((k_points_hist<AR>), grid, blk, 0, s, a, pa, ga));
    hipLaunchKernelGGL(k_scale_stage, dim3(1), dim3(256), 0, s, ga, a.state, 1, a.pair_base);
  }
  UWT_WITH_AR(sel.arith,
    if (unit) hipLaunchKernelGGL((k_points_general<AR, true>), grid, blk, 0, s, a, pa, ga);
    else hipLaunchKernelGGL((k_points_general<AR, false>), grid, blk, 0, s, a, pa, ga));
}

}  // namespace uwt
