#!/bin/bash
# copy the summaries of a tools/collect_profiles.sh run into profiles/<round>/ under the names the README there lists
#   tools/publish_profiles.sh <gpurun_out subdir> <round dir, e.g. r03>
set -e
src=gpurun_out/$1; dst=profiles/$2
mkdir -p $dst
for d in $src/stats_*; do
  n=$(basename $d | sed 's/^stats_//')
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $dst/kernel_stats_bench_$n.csv
  [ -f $src/bench_$n.json ] && cp $src/bench_$n.json $dst/bench_$n.json
  [ -f $src/bench_${n}_no_profiler.json ] && cp $src/bench_${n}_no_profiler.json $dst/bench_${n}_no_profiler.json
done
cp $src/pmc_fetch.csv $dst/pmc_fetch_bench_default_p1024.csv
cp $src/pmc_write.csv $dst/pmc_write_bench_default_p1024.csv
cp $src/sq_counters_*.csv $dst/
python3 tools/make_profile_facts.py $dst
