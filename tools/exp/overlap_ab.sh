#!/bin/bash
# finer levels' gradients on the side stream beside the coarse iterations (default at >= 768 pairs) against everything in turn
for r in 1 2 3; do for o in 1 0; do
  UWT_OVERLAP_GRAD=$o python bench.py --cpu-pairs 0 --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap_grad $o #$r', d['value'], d['ms_per_step'])"
done; done
