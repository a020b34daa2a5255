#!/bin/bash
# blocks per residual launch the slicing aims at (two-stream form), default workload and Huber at 256 pairs
for r in 1 2; do for tb in 768 1024 1536 2048; do
  UWT_TARGET_BLOCKS=$tb python bench.py --cpu-pairs 0 --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('target_blocks $tb #$r default', d['value'], d['ms_per_step'])"
  UWT_TARGET_BLOCKS=$tb python bench.py --pairs 256 --unique 8 --cpu-pairs 0 --weights huber --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('target_blocks $tb #$r huber256', d['value'], d['ms_per_step'])"
done; done
