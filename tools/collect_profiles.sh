#!/bin/bash
# Collect the round's profiles on the GPU box (run from the repo root through gpurun):
#   tools/collect_profiles.sh <out dir under gpurun_out>
# Round 5: every figure of the default workload under both arithmetic sets (opencv = the default, legacy), the reference's EUROC
# geometry, config 4's single-GPU leg.  Round 6: ROI-like frame sizes (what System::CalculateROI leaves of a EUROC frame), the
# one-pair latency by schedule.
# kernel-trace statistics and the bench line of the default workload and of the other quoted configurations, HBM
# traffic counters (separate --pmc passes, never together with other trace domains), SQ counters of the residual kernel.
set -u
R=$(pwd)
out=$R/gpurun_out/${1:-prof_r06}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args...
  name=$1; shift
  (cd $R && rocprofv3 --kernel-trace --stats -d $out/stats_$name --output-format csv -- python3 bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err)
  (cd $R && python3 bench.py "$@" > $out/bench_${name}_no_profiler.json 2> $out/bench_${name}_no_profiler.err)   # the same line without the profiler
}
stats default_p1024
stats legacy_default_p1024 --arith legacy
stats cfg3_1280x960_l5_p256 --width 1280 --height 960 --levels 5 --pairs 256 --cpu-pairs 8 --unique 8
stats refsched_p1024 --reference-schedule --cpu-pairs 32
stats tukey_p256 --weights tukey --pairs 256 --cpu-pairs 8 --unique 8
stats huber_p256 --weights huber --pairs 256 --cpu-pairs 8 --unique 8
stats bilinear_p256 --bilinear --pairs 256 --cpu-pairs 8 --unique 8
stats nodepth_p1024 --no-depth --cpu-pairs 16
stats bilinear_huber_p256 --bilinear --weights huber --pairs 256 --cpu-pairs 8 --unique 8
stats huber_p1024 --weights huber --cpu-pairs 8 --unique 8
stats legacy_huber_p256 --arith legacy --weights huber --pairs 256 --cpu-pairs 8 --unique 8
E=458.654,457.296,367.215,248.375; E736=458.654,457.296,359.215,248.375
stats euroc_640x480_l4_p1024 --intrinsics $E --cpu-pairs 16
stats euroc_640x480_l4_nodepth_p1024 --intrinsics $E --no-depth --cpu-pairs 16
stats euroc_736x480_l5_p1024 --width 736 --intrinsics $E736 --levels 5 --no-depth --unique 64 --cpu-pairs 16
stats euroc_752x480_l5_p1024 --width 752 --intrinsics $E --levels 5 --no-depth --unique 64 --cpu-pairs 16
stats euroc_736x480_refsched_p1024 --width 736 --intrinsics $E736 --reference-schedule --no-depth --unique 64 --cpu-pairs 16
stats roi_733x471_l5_p1024 --width 733 --height 471 --intrinsics $E736 --levels 5 --no-depth --unique 64 --cpu-pairs 16
stats roi_725x465_l5_p1024 --width 725 --height 465 --intrinsics $E736 --levels 5 --no-depth --unique 64 --cpu-pairs 16
stats roi_735x479_l5_p1024 --width 735 --height 479 --intrinsics $E736 --levels 5 --no-depth --unique 64 --cpu-pairs 16
stats roi_733x471_refsched_p1024 --width 733 --height 471 --intrinsics $E736 --reference-schedule --no-depth --unique 64 --cpu-pairs 16
(cd $R && python3 bench.py --gpus 1 --total-pairs 8192 --steps 5 --warmup 2 > $out/bench_total8192_g1.json 2> $out/bench_total8192_g1.err)
(cd $R && python3 tools/exp/latency_general.py 200 > $out/latency_single_pair.txt 2>/dev/null)
(cd $R && python3 tools/exp/latency_identity.py 300 > $out/latency_identity_by_level.txt 2>/dev/null)
(cd $R && python3 tools/exp/call_overhead.py > $out/call_overhead.txt 2>/dev/null)
(cd $R && python3 tools/exp/stage_timing.py > $out/stage_timing.txt 2>/dev/null)
for a in opencv legacy; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd $R && rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${a}_$c --output-format csv -- python3 bench.py --arith $a --cpu-pairs 0 --steps 2 --warmup 1 --no-profile > $out/pmc_${a}_$c.log 2>&1)
  done
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_INSTS_BRANCH"; do
    i=$((i+1))
    (cd $R && rocprofv3 --kernel-trace --pmc $set -d $out/sq_$a/pass$i --output-format csv -- python3 bench.py --arith $a --levels 1 --steps 1 --warmup 1 --cpu-pairs 0 --no-profile > $out/sq_${a}_pass$i.log 2>&1)
  done
done
cd $R
for a in opencv legacy; do
  python3 tools/pmc_summary.py $out/pmc_${a}_FETCH_SIZE $out/pmc_fetch_$a.csv
  python3 tools/pmc_summary.py $out/pmc_${a}_WRITE_SIZE $out/pmc_write_$a.csv
  python3 tools/sq_summary.py $out/sq_$a k_residual $((1024*640*480)) $out/sq_counters_k_residual_${a}_level0_p1024.csv
done
# the robust-weight path: level-0 launches of the scale pass and of the weighted accumulation at 256 pairs
bash tools/sq_passes.sh ${1:-prof_r06}/sq_huber k_resid_hist_v --pairs 256 --unique 8 --weights huber > /dev/null 2>&1
python3 tools/sq_summary.py $out/sq_huber k_resid_hist_v $((256*640*480)) $out/sq_counters_k_resid_hist_v_level0_p256_huber.csv
python3 tools/sq_summary.py $out/sq_huber "k_residual<" $((256*640*480)) $out/sq_counters_k_residual_weighted_level0_p256_huber.csv
python3 tools/per_level_table.py $(find $out/stats_default_p1024 -name "*kernel_trace.csv" | head -1) > $out/per_level_launch_table_trace.md 2>/dev/null
python3 -c "import ctypes; h = ctypes.CDLL('uw-slam_amd/libuwt_hip.so'); h.uwt_source_id.restype = ctypes.c_char_p; print(h.uwt_source_id().decode())" > $out/library_source_id.txt
# keep what travels back small: the statistics tables, not the raw traces
find $out -name "*kernel_trace.csv" -size +20M -delete
find $out -name "*counter_collection.csv" -size +20M -delete
du -sh $out
