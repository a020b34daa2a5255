#!/bin/bash
# a larger survey at full size on the round's last library (OpenCV set): 640x480, fixed and reference schedules, depth, batch and
# one-pair-per-call forms, Huber
out=gpurun_out/${1:-r4survey_big}.txt; : > $out
S="python tools/parity_survey.py --arith opencv --w 640 --h 480 --depth 1"
$S --n 256 --mode fixed --seed0 5000 >> $out 2>/dev/null
$S --n 256 --mode reference --seed0 6000 >> $out 2>/dev/null
$S --n 96 --mode fixed --single 1 --seed0 7000 >> $out 2>/dev/null
$S --n 128 --mode fixed --weights 2 --seed0 8000 >> $out 2>/dev/null
$S --n 128 --mode fixed --weights 1 --seed0 9000 >> $out 2>/dev/null
UWT_STREAM_MB=0 $S --n 128 --mode fixed --seed0 10000 >> $out 2>/dev/null
grep "bit-identical" $out | cut -c1-170
