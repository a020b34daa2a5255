#!/bin/bash
# single-pair latency (tools/exp/latency_general.py) of library builds, interleaved on one box: latency_ab.sh <rounds> <reps> <name> <name> ...
rounds=$1; reps=$2; shift 2
for r in $(seq 1 $rounds); do for v in "$@"; do
  cp tools/exp/ablibs/lib_$v.so uw-slam_amd/libuwt_hip.so
  python tools/exp/latency_general.py $reps 2>/dev/null | sed "s/^/$v#$r  /"
done; done
