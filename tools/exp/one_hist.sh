#!/bin/bash
# compile tools/exp/one_kernel.hip (seconds) and print the resources + main-loop instruction histogram of one instantiation
#   tools/exp/one_hist.sh [mangled substring, default the production OpenCV-set kernel] [-D...]
cd "$(dirname "$0")/../.."
key=${1:-k_residualILi0ELi4ELb1ELb1ELb0EdLb1ELi0ELi0ELb0ELb0E}; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -disable-machine-licm -Wno-unused-function \
  --cuda-device-only -S -Rpass-analysis=kernel-resource-usage -I uw-slam_amd/csrc "$@" tools/exp/one_kernel.hip -o /tmp/one_kernel.s 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - | grep "$key"
ISA_FILE=/tmp/one_kernel.s python3 tools/isa_hist.py "$key" --loop | head -${HIST_LINES:-12}
