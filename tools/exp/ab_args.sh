#!/bin/bash
# usage: ab_args.sh <rounds> "<bench args>" <libA> <libB> ...  -- interleaved bench.py runs of library variants with given arguments
rounds=$1; args=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/exp/lib_$v.so uw-slam_amd/libuwt_hip.so
    python bench.py --cpu-pairs 0 --no-profile $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v#$r [$args]', d['value'], d['ms_per_step'])"
  done
done
