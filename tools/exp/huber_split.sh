#!/bin/bash
# Huber, 256 pairs: throughput by the number of parts (streams) the batch is cut into
for r in 1 2; do
for s in 1 2 3 4; do
  UWT_SPLIT=$s python bench.py --pairs 256 --unique 8 --cpu-pairs 0 --weights huber 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $s #$r', d['value'], d['ms_per_step'])"
done; done
