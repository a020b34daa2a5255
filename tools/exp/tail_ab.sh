#!/bin/bash
# tail update on / off on one box: value, ms per step, roofline.frac and the per-level evaluation times
for r in 1 2; do for t in 0 1; do
  for cfg in "--cpu-pairs 0" "--pairs 256 --unique 8 --cpu-pairs 0 --weights huber"; do
    UWT_TAIL_UPDATE=$t python bench.py $cfg 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('tail $t', '$cfg'[-12:], d['value'], d['ms_per_step'], d['roofline']['frac'], [round(l['avg_ms_per_evaluation']*1e3,1) for l in d['roofline']['per_level']])"
  done
done; done
