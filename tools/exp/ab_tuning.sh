#!/bin/bash
# interleaved A/B of uwt_tuning settings on one box: ab_tuning.sh <rounds> "<bench args>" <setting> <setting> ...   (setting: k=v[,k=v] or "defaults")
rounds=$1; args=$2; shift 2
for r in $(seq 1 $rounds); do
  for t in "$@"; do
    tt=""; [ "$t" != "defaults" ] && tt="--tuning $t"
    python bench.py --cpu-pairs 0 --no-profile --steps 20 --warmup 5 $args $tt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t#$r [$args]', d['value'], d['ms_per_step'])"
  done
done
