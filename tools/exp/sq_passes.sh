#!/bin/bash
# usage: sq_passes.sh <label> <lib variant in tools/exp or "cur"> -- SQ counter passes on the level-0 residual launches (256 pairs)
label=$1; v=$2
[ "$v" != "cur" ] && cp tools/exp/lib_$v.so uw-slam_amd/libuwt_hip.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$label
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_INSTS_BRANCH" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  (cd $R && rocprofv3 --kernel-trace --pmc $set -d $out/pass$i --output-format csv -- python3 bench.py --pairs 256 --levels 1 --steps 1 --warmup 1 --cpu-pairs 0 --no-profile > $out/pass$i.log 2>&1)
done
cd $R && python3 tools/sq_summary.py $out k_residual $((256*640*480)) $out/summary.csv
