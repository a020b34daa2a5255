#!/bin/bash
# copy the summaries of a tools/collect_profiles.sh run into profiles/<round>/ under the names the README there lists
#   tools/publish_profiles.sh <gpurun_out subdir> <round dir, e.g. r04>
# Refuses when the library in the tree is not the one the counters were collected on (the facts are stamped with its source id).
set -e
src=gpurun_out/$1; dst=profiles/$2
mkdir -p $dst
if [ -f $src/library_source_id.txt ]; then
  want=$(cat $src/library_source_id.txt)
  have=$(python3 -c "import ctypes; h = ctypes.CDLL('uw-slam_amd/libuwt_hip.so'); h.uwt_source_id.restype = ctypes.c_char_p; print(h.uwt_source_id().decode())")
  if [ "$want" != "$have" ]; then echo "library sources changed since the collection ($want vs $have): collect again" >&2; exit 1; fi
fi
for d in $src/stats_*; do
  n=$(basename $d | sed 's/^stats_//')
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $dst/kernel_stats_bench_$n.csv
  [ -f $src/bench_$n.json ] && cp $src/bench_$n.json $dst/bench_$n.json
  [ -f $src/bench_${n}_no_profiler.json ] && cp $src/bench_${n}_no_profiler.json $dst/bench_${n}_no_profiler.json
done
for a in opencv legacy; do
  cp $src/pmc_fetch_$a.csv $dst/pmc_fetch_${a}_bench_default_p1024.csv
  cp $src/pmc_write_$a.csv $dst/pmc_write_${a}_bench_default_p1024.csv
done
cp $src/sq_counters_*.csv $dst/
[ -s $src/per_level_launch_table_trace.md ] && cp $src/per_level_launch_table_trace.md $dst/
for f in bench_total8192_g1.json latency_single_pair.txt latency_identity_by_level.txt call_overhead.txt stage_timing.txt; do [ -s $src/$f ] && cp $src/$f $dst/; done
python3 tools/kernel_resources.py -o $dst/kernel_resources.md > /dev/null
python3 tools/make_profile_facts.py $dst
