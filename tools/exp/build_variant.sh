#!/bin/bash
# usage: build_variant.sh <name> [-DUWT_EXP_... ...]  -- builds tools/exp/lib_<name>.so from the working tree with extra defines
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
cd "$ROOT/uw-slam_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -disable-machine-licm -Wno-unused-function "$@" -shared -o "$ROOT/tools/exp/lib_$name.so" uwt_capi.hip
if [ -n "$SHOW" ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -disable-machine-licm "$@" -S --cuda-device-only -o /tmp/$name.s uwt_capi.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "Function Name: _ZN3uwt10k_residualILi4ELb1ELb1ELb0EdLb1ELi0ELi0ELb0E" | grep -E "VGPRs|Occ|VGPRs Spill"
fi
