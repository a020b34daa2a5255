#!/bin/bash
# SQ passes only (level-0 launches at 1024 pairs) into gpurun_out/<dir>
R=$(pwd); out=$R/gpurun_out/${1:-prof_r02}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_INSTS_BRANCH"; do
  i=$((i+1))
  (cd $R && rocprofv3 --kernel-trace --pmc $set -d $out/sq0/pass$i --output-format csv -- python3 bench.py --levels 1 --steps 1 --warmup 1 --cpu-pairs 0 --no-profile > $out/sq0_pass$i.log 2>&1)
done
cd $R && python3 tools/sq_summary.py $out/sq0 k_residual $((1024*640*480)) $out/sq_counters_k_residual_level0_p1024.csv
find $out -name "*counter_collection.csv" -size +20M -delete; find $out -name "*kernel_trace.csv" -size +20M -delete
