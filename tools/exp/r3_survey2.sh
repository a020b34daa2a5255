#!/bin/bash
# second parity survey of the round (the last kernels: Scharr rewrite, tail update): other seeds, larger batches, the
# two-stream form with the tail update at full size, either update form forced
out=gpurun_out/r3survey2.txt; : > $out
S="python tools/parity_survey.py --seed0 50000"
for mode in fixed reference; do for depth in 0 1; do
  $S --n 1200 --w 320 --h 240 --mode $mode --depth $depth >> $out 2>/dev/null
done; done
for t in 0 2; do UWT_TAIL_UPDATE=$t $S --n 600 --w 320 --h 240 --mode fixed --depth 1 >> $out 2>/dev/null; done
for t in 0 2; do UWT_TAIL_UPDATE=$t $S --n 300 --w 320 --h 240 --mode fixed --depth 1 --weights 2 >> $out 2>/dev/null; done
$S --n 256 --w 640 --h 480 --mode fixed --depth 1 >> $out 2>/dev/null            # two streams, tail update (default)
$S --n 256 --w 640 --h 480 --mode reference --depth 1 >> $out 2>/dev/null
$S --n 128 --w 640 --h 480 --mode fixed --depth 1 --weights 2 >> $out 2>/dev/null
$S --n 128 --w 640 --h 480 --mode fixed --depth 1 --weights 1 >> $out 2>/dev/null
$S --n 128 --w 640 --h 480 --mode fixed --depth 0 --sampler 1 >> $out 2>/dev/null
$S --n 64 --w 1280 --h 960 --mode fixed --depth 1 >> $out 2>/dev/null
UWT_FUZZ_SEEDS=4000 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz 2>&1 | tail -2 >> $out
grep -c "bit-identical" $out; cut -c1-170 $out
