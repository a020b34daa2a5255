#!/bin/bash
# kernel statistics of one bench configuration:  tools/exp/r3_stats.sh <name> <bench args...>
set -u
R=$(pwd); name=$1; shift
out=$R/gpurun_out/r3stats; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --stats -d $out/stats_$name --output-format csv -- python3 bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err)
f=$(find $out/stats_$name -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats_$name.csv
find $out/stats_$name -name "*kernel_trace.csv" -size +5M -delete
head -12 $out/kernel_stats_$name.csv | cut -c1-200
tail -1 $out/bench_$name.json | cut -c1-300
