// Explicit instantiation of a few kernels for quick resource / ISA checks (seconds instead of the library's minutes):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize --cuda-device-only -S \
//         -Rpass-analysis=kernel-resource-usage -I uw-slam_amd/csrc tools/exp/one_kernel.hip -o /tmp/one_kernel.s
#include "uwt_kernels.h"
namespace uwt {
template __global__ void k_residual<kArithOpenCV, 4, true, true, false, double, true, 0, 0, false>(const ResidualArgs);
#ifdef ONE_KERNEL_MORE
template __global__ void k_residual<kArithLegacy, 4, true, true, false, double, true, 0, 0, false>(const ResidualArgs);
template __global__ void k_residual<kArithOpenCV, 4, false, true, false, double, true, 0, 0, false>(const ResidualArgs);
template __global__ void k_residual<kArithOpenCV, 4, true, true, false, double, true, 0, 2, false>(const ResidualArgs);
template __global__ void k_residual<kArithOpenCV, 4, true, true, false, double, true, 1, 2, false>(const ResidualArgs);
template __global__ void k_coarse<kArithOpenCV, true, true, double, true, 14, 1>(const CoarseArgs);
#endif
}
#ifdef ONE_KERNEL_W4
namespace uwt {
template __global__ void k_coarse_w4<kArithOpenCV, true, true, double, true, 14, 1>(const CoarseArgs);
}
#endif
#ifdef ONE_KERNEL_RW4
namespace uwt {
template __global__ void k_residual_w4<kArithOpenCV, 4, true, true, false, double, true, 1, 2>(const ResidualArgs);
template __global__ void k_residual_w4<kArithLegacy, 4, true, true, false, double, true, 1, 2>(const ResidualArgs);
template __global__ void k_residual<kArithLegacy, 4, true, true, false, double, true, 1, 2>(const ResidualArgs);
}
#endif
#ifdef ONE_KERNEL_CW
namespace uwt {
template __global__ void k_coarse_weighted<kArithOpenCV, true, 2, 1>(const CoarseArgs);
template __global__ void k_coarse_weighted<kArithOpenCV, true, 1, 3>(const CoarseArgs);
}
#endif
