#!/bin/bash
# quick resource check of the production kernel(s): tools/exp/one_kernel.sh [-DONE_KERNEL_MORE]
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -disable-machine-licm -Wno-unused-function \
  --cuda-device-only -S -Rpass-analysis=kernel-resource-usage -I uw-slam_amd/csrc "$@" tools/exp/one_kernel.hip -o /tmp/one_kernel.s 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//' | paste - - - - 
