#!/bin/bash
# The round's parity survey: every launch form and mode against the oracle, bit for bit, under BOTH arithmetic sets, plus the
# shapes of round 5 (EUROC geometry).  tools/parity_survey.sh <name under gpurun_out>
out=gpurun_out/${1:-r5survey}.txt; : > $out
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
  $S --n 300 --w 320 --h 240 --mode fixed --depth 1 --tuning split_min_px=1 >> $out 2>/dev/null
  # the reference's EUROC geometry (fx != fy; 736 x 480 x 5: the 46-wide level pixel by pixel), batch and one pair per call
  E="--intrinsics 458.654,457.296,359.215,248.375 --w 736 --h 480"
  $S $E --n 48 --mode reference >> $out 2>/dev/null
  $S $E --n 32 --mode reference --single 1 >> $out 2>/dev/null
  $S $E --n 32 --mode fixed5 >> $out 2>/dev/null
done
# full size, the OpenCV set: both schedules, one pair per call, robust weights, every level streamed
S="python tools/parity_survey.py --arith opencv --w 640 --h 480 --depth 1"
$S --n 256 --mode fixed --seed0 5000 >> $out 2>/dev/null
$S --n 256 --mode reference --seed0 6000 >> $out 2>/dev/null
$S --n 96 --mode fixed --single 1 --seed0 7000 >> $out 2>/dev/null
$S --n 128 --mode fixed --weights 2 --seed0 8000 >> $out 2>/dev/null
$S --n 128 --mode fixed --weights 1 --seed0 9000 >> $out 2>/dev/null
$S --n 128 --mode fixed --seed0 10000 --tuning stream_bytes=0 >> $out 2>/dev/null
$S --n 128 --mode fixed --seed0 11000 --tuning typed_loads=0 >> $out 2>/dev/null
UWT_FUZZ_SEEDS=400 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz 2>&1 | tail -2 >> $out
grep -c "bit-identical" $out; grep "bit-identical" $out | cut -c1-190
