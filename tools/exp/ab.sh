#!/bin/bash
# usage: ab.sh <rounds> <libA> <libB> ...  -- interleaved level-0 timings of library variants in tools/exp/
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/exp/lib_$v.so uw-slam_amd/libuwt_hip.so
    tools/exp/run_l0.sh "$v#$r"
  done
done
