#!/bin/bash
# round 4: memory-path counters over the level-0 launches of k_residual (what do the plane loads wait for?):
#   r4_mem_counters.sh <label> [bench args...]
# (two counters of a block per pass: more and rocprofv3 aborts with "exceeds the capabilities of the hardware" and then does not
#  exit — every pass runs under its own timeout)
label=$1; shift
R=$(pwd); out=$R/gpurun_out/$label; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_UTCL2_BUSY"; do
  i=$((i+1))
  (cd $R && timeout -k 5 100 rocprofv3 --kernel-trace --pmc $set -d $out/pass$i --output-format csv -- python3 bench.py --levels 1 --steps 1 --warmup 1 --cpu-pairs 0 --no-profile "$@" > $out/pass$i.log 2>&1)
done
cd $R && python3 tools/sq_summary.py $out "k_residual" $((1024*640*480)) $out/summary_$label.csv
find $out -name "*counter_collection.csv" -size +5M -delete
