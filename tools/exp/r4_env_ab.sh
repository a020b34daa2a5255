#!/bin/bash
# round 4: launch-shape switches re-measured under the OpenCV arithmetic set (interleaved on one box, the tree's library)
#   tools/exp/r4_env_ab.sh <out file> [bench args]
out=$1; shift
for r in 1 2 3; do
  for e in "X=0" "UWT_TARGET_BLOCKS=512" "UWT_TARGET_BLOCKS=2048" "UWT_COARSE_BATCH_PX=20000" "UWT_COARSE_BATCH_PX=0" "UWT_SPLIT=1" "UWT_SPLIT=3" "UWT_TAIL_UPDATE=0" "UWT_OVERLAP_GRAD=0"; do
    env $e python bench.py --cpu-pairs 0 --no-profile "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e #$r', d['value'], d['ms_per_step'])"
  done
done > $out 2>&1
