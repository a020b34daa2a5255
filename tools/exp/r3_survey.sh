#!/bin/bash
# parity survey of the round: every launch form and mode against the oracle, bit for bit
out=gpurun_out/r3survey.txt; : > $out
S="python tools/parity_survey.py"
for mode in fixed reference; do for depth in 0 1; do for single in 0 1; do
  $S --n 400 --w 320 --h 240 --mode $mode --depth $depth --single $single >> $out 2>/dev/null
done; done; done
for mode in fixed reference; do $S --n 96 --w 640 --h 480 --mode $mode --depth 1 >> $out 2>/dev/null; done
$S --n 64 --w 640 --h 480 --mode fixed --depth 1 --single 1 >> $out 2>/dev/null
for wgt in 1 2; do for depth in 0 1; do $S --n 200 --w 320 --h 240 --mode fixed --depth $depth --weights $wgt >> $out 2>/dev/null; done; done
$S --n 200 --w 320 --h 240 --mode fixed --depth 1 --sampler 1 >> $out 2>/dev/null
$S --n 200 --w 320 --h 240 --mode fixed --depth 1 --sampler 1 --weights 2 >> $out 2>/dev/null
$S --n 48 --w 640 --h 480 --mode fixed --depth 1 --weights 2 >> $out 2>/dev/null
UWT_FUSED=1 $S --n 100 --w 320 --h 240 --mode fixed --depth 1 --weights 2 >> $out 2>/dev/null
UWT_SPLIT_MIN_PX=1 $S --n 400 --w 320 --h 240 --mode fixed --depth 1 >> $out 2>/dev/null
UWT_FUZZ_SEEDS=1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz 2>&1 | tail -2 >> $out
grep -c "bit-identical" $out; grep "bit-identical" $out | awk '{print $0}' | cut -c1-150
