#!/bin/bash
# the coarsest level of a batch as one k_coarse launch (default) against per-evaluation launches (UWT_COARSE_BATCH_PX=0)
for r in 1 2 3; do for px in 6144 0; do
  UWT_COARSE_BATCH_PX=$px python bench.py --cpu-pairs 0 --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('coarse_px $px #$r', d['value'], d['ms_per_step'])"
done; done
