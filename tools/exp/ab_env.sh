#!/bin/bash
# usage: ab_env.sh <rounds> <lib>:<ENV=val>...  -- interleaved default bench of (library variant, environment) combinations
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    lib=${v%%:*}; envs=${v#*:}
    cp tools/exp/lib_$lib.so uw-slam_amd/libuwt_hip.so
    env $envs python bench.py --cpu-pairs 0 --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v#$r', d['value'], d['ms_per_step'])"
  done
done
