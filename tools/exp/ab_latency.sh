#!/bin/bash
# usage: ab_latency.sh <rounds> <libA> <libB> ...  -- interleaved single-pair latencies (bench's own leg) and small batches
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/exp/lib_$v.so uw-slam_amd/libuwt_hip.so
    python bench.py --cpu-pairs 0 --pairs 16 --unique 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['single_pair_latency']; print('$v#$r p16', d['value'], 'single', s['bench_schedule_ms'], s['reference_schedule_ms'])"
    python bench.py --cpu-pairs 0 --pairs 64 --unique 8 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v#$r p64', d['value'])"
  done
done
