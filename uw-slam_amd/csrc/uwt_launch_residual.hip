// uwt_launch_residual.hip — dispatch of k_residual on the identity-weights / nearest-sampler path (the dominant kernel), its
// per-stage dump form, and k_residual_points.
#include "uwt_launch.h"

namespace uwt {
namespace {

// RAGGED: the level's grid rows are not whole groups of four (level_ragged): the instantiations that mask the positions beyond the
// grid.  They exist for the shapes odd-sized frames run in production — f64 sums with typed plane loads, square pixels or not —
// and as one plain form for everything else (f32 sums, the per-stage dumps, non-unit factors, typed loads switched off).
template <int AR, bool DEPTH, bool UNIT, bool DUMP>
void launch_residual_ragged_t(hipStream_t s, const ResidualArgs& a, int n_pairs, bool acc64) {
  const dim3 grid(a.slices, n_pairs), blk(kBlock);
  if constexpr (UNIT && !DUMP) {
    if (acc64 && a.typed_loads) {
      const bool sq = a.L.fx == a.L.fy;
      if (sq && a.stream_planes) hipLaunchKernelGGL((k_residual<AR, 4, DEPTH, UNIT, DUMP, double, true, 0, 0, false, kLoadsTyped | kLoadsStream, true>), grid, blk, 0, s, a);
      else if (sq) hipLaunchKernelGGL((k_residual<AR, 4, DEPTH, UNIT, DUMP, double, true, 0, 0, false, kLoadsTyped, true>), grid, blk, 0, s, a);
      else if (a.stream_planes) hipLaunchKernelGGL((k_residual<AR, 4, DEPTH, UNIT, DUMP, double, false, 0, 0, false, kLoadsTyped | kLoadsStream, true>), grid, blk, 0, s, a);
      else hipLaunchKernelGGL((k_residual<AR, 4, DEPTH, UNIT, DUMP, double, false, 0, 0, false, kLoadsTyped, true>), grid, blk, 0, s, a);
      return;
    }
  }
  if (acc64) hipLaunchKernelGGL((k_residual<AR, 4, DEPTH, UNIT, DUMP, double, false, 0, 0, false, 0, true>), grid, blk, 0, s, a);
  else hipLaunchKernelGGL((k_residual<AR, 4, DEPTH, UNIT, DUMP, float, false, 0, 0, false, 0, true>), grid, blk, 0, s, a);
}

template <int AR, bool DEPTH, bool UNIT, bool DUMP>
void launch_residual_t(hipStream_t s, const ResidualArgs& a, int n_pairs, bool acc64, bool compute_only) {
  constexpr int VEC = 4;
  const dim3 grid(a.slices, n_pairs), blk(kBlock);
  if constexpr (UNIT && !DUMP) {
    if (acc64 && a.L.fx == a.L.fy) {   // the production instantiation, its diagnostic twin and its streamed twin (load_group)
      if (compute_only) hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, double, true, 0, 0, true>), grid, blk, 0, s, a);
      else if (a.typed_loads && a.stream_planes) hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, double, true, 0, 0, false, kLoadsTyped | kLoadsStream>), grid, blk, 0, s, a);
      else if (a.typed_loads) hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, double, true, 0, 0, false, kLoadsTyped>), grid, blk, 0, s, a);
      else if (a.stream_planes) hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, double, true, 0, 0, false, kLoadsStream>), grid, blk, 0, s, a);
      else hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, double, true>), grid, blk, 0, s, a);
      return;
    }
  }
  if constexpr (!DUMP) {
    // the general Jacobian form (fx != fy — the reference's own EUROC calibration: 458.654 / 457.296 — and / or non-unit
    // factors): its streamed twin too
    if constexpr (UNIT) {   // fx != fy with unit factors (EUROC): typed plane loads here too
      if (acc64 && a.typed_loads) {
        if (a.stream_planes) hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, double, false, 0, 0, false, kLoadsTyped | kLoadsStream>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, double, false, 0, 0, false, kLoadsTyped>), grid, blk, 0, s, a);
        return;
      }
    }
    if (acc64 && a.stream_planes) {
      hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, double, false, 0, 0, false, kLoadsStream>), grid, blk, 0, s, a);
      return;
    }
  }
  if (acc64) hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, double>), grid, blk, 0, s, a);
  else hipLaunchKernelGGL((k_residual<AR, VEC, DEPTH, UNIT, DUMP, float>), grid, blk, 0, s, a);
}

template <int AR, bool DEPTH>
void launch_residual_vd(hipStream_t s, const ResidualArgs& a, int n_pairs, bool unit, bool dump, bool acc64, bool co) {
  if (level_ragged(a.L)) {   // (no compute-only twin of the ragged forms: the diagnostic measures the whole-level production kernel)
    if (unit) {
      if (dump) launch_residual_ragged_t<AR, DEPTH, true, true>(s, a, n_pairs, acc64);
      else launch_residual_ragged_t<AR, DEPTH, true, false>(s, a, n_pairs, acc64);
    } else {
      if (dump) launch_residual_ragged_t<AR, DEPTH, false, true>(s, a, n_pairs, acc64);
      else launch_residual_ragged_t<AR, DEPTH, false, false>(s, a, n_pairs, acc64);
    }
    return;
  }
  if (unit) {
    if (dump) launch_residual_t<AR, DEPTH, true, true>(s, a, n_pairs, acc64, false);
    else launch_residual_t<AR, DEPTH, true, false>(s, a, n_pairs, acc64, co);
  } else {
    if (dump) launch_residual_t<AR, DEPTH, false, true>(s, a, n_pairs, acc64, false);
    else launch_residual_t<AR, DEPTH, false, false>(s, a, n_pairs, acc64, co);
  }
}

}  // namespace

void launch_residual(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, int n_pairs, bool dump) {
  const bool unit = (a.zf == 1.0f && a.af == 1.0f);
  const bool co = sel.compute_only && !dump;
  UWT_WITH_AR(sel.arith,
    if (sel.depth) launch_residual_vd<AR, true>(s, a, n_pairs, unit, dump, sel.acc64, co);
    else launch_residual_vd<AR, false>(s, a, n_pairs, unit, dump, sel.acc64, co));
}

void launch_points(hipStream_t s, const LaunchSel& sel, const ResidualArgs& a, const PointsArgs& pa) {
  const bool unit = (a.zf == 1.0f && a.af == 1.0f);
  const dim3 grid(a.slices), blk(kBlock);
  UWT_WITH_AR(sel.arith,
    if (unit && sel.acc64) hipLaunchKernelGGL((k_residual_points<AR, true, false, double>), grid, blk, 0, s, a, pa);
    else if (unit) hipLaunchKernelGGL((k_residual_points<AR, true, false, float>), grid, blk, 0, s, a, pa);
    else if (sel.acc64) hipLaunchKernelGGL((k_residual_points<AR, false, false, double>), grid, blk, 0, s, a, pa);
    else hipLaunchKernelGGL((k_residual_points<AR, false, false, float>), grid, blk, 0, s, a, pa));
}

}  // namespace uwt
