#!/bin/bash
# interleaved A/B of library builds on one box: ab_libs.sh <rounds> "<bench args>" <name> <name> ...   (tools/exp/ablibs/lib_<name>.so; scratch copies)
rounds=$1; args=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/exp/ablibs/lib_$v.so uw-slam_amd/libuwt_hip.so
    python bench.py --cpu-pairs 0 --no-profile --steps 20 --warmup 5 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v#$r [$args]', d['value'], d['ms_per_step'])"
  done
done
