#!/bin/bash
# usage: ab_full.sh <rounds> <libA> <libB> ...  -- interleaved full default bench of library variants in tools/exp/
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/exp/lib_$v.so uw-slam_amd/libuwt_hip.so
    python bench.py --cpu-pairs 0 --steps 8 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v#$r', d['value'], d['ms_per_step'])"
  done
done
