#!/bin/bash
# parity survey of round 4: every launch form and mode against the oracle, bit for bit, under BOTH arithmetic sets
out=gpurun_out/${1:-r4survey}.txt; : > $out
for ar in opencv legacy; do
  S="python tools/parity_survey.py --arith $ar"
  for mode in fixed reference; do for depth in 0 1; do for single in 0 1; do
    $S --n 300 --w 320 --h 240 --mode $mode --depth $depth --single $single >> $out 2>/dev/null
  done; done; done
  for mode in fixed reference; do $S --n 64 --w 640 --h 480 --mode $mode --depth 1 >> $out 2>/dev/null; done
  $S --n 48 --w 640 --h 480 --mode fixed --depth 1 --single 1 >> $out 2>/dev/null
  for wgt in 1 2; do for depth in 0 1; do $S --n 150 --w 320 --h 240 --mode fixed --depth $depth --weights $wgt >> $out 2>/dev/null; done; done
  $S --n 150 --w 320 --h 240 --mode fixed --depth 1 --sampler 1 >> $out 2>/dev/null
  $S --n 150 --w 320 --h 240 --mode fixed --depth 1 --sampler 1 --weights 2 >> $out 2>/dev/null
  $S --n 32 --w 640 --h 480 --mode fixed --depth 1 --weights 2 >> $out 2>/dev/null
  UWT_SPLIT_MIN_PX=1 $S --n 300 --w 320 --h 240 --mode fixed --depth 1 >> $out 2>/dev/null
done
UWT_FUZZ_SEEDS=600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz 2>&1 | tail -2 >> $out
grep -c "bit-identical" $out; grep "bit-identical" $out | cut -c1-170
