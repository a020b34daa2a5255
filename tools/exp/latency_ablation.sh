#!/bin/bash
# What each stage of a lone pair's evaluation costs, by ablation: library builds with one stage removed (results meaningless), interleaved on
# one box.  latency_ablation.sh <rounds> <reps> <name> ...   (tools/exp/ablibs/lib_<name>.so)
rounds=$1; reps=$2; shift 2
for r in $(seq 1 $rounds); do for v in "$@"; do
  cp tools/exp/ablibs/lib_$v.so uw-slam_amd/libuwt_hip.so
  python tools/exp/latency_identity.py $reps 2>/dev/null | sed "s/^/$v#$r  /"
done; done
