#!/bin/bash
# SQ counter passes over level-0 launches of one kernel:  r3_sq.sh <label> <kernel substring> <bench args...>
label=$1; pat=$2; shift 2
R=$(pwd); out=$R/gpurun_out/$label; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  (cd $R && rocprofv3 --kernel-trace --pmc $set -d $out/pass$i --output-format csv -- python3 bench.py --levels 1 --steps 1 --warmup 1 --cpu-pairs 0 --no-profile "$@" > $out/pass$i.log 2>&1)
done
cd $R && python3 tools/sq_summary.py $out "$pat" $((256*640*480)) $out/summary_$label.csv
find $out -name "*counter_collection.csv" -size +5M -delete
