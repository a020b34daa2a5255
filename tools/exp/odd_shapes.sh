#!/bin/bash
# Whole alignments at sizes whose coarse levels are not whole 4-pixel groups (widths 16 k and 8 k with k odd, tiny coarse levels: a size
# must divide by 2^(levels - 1), include/uwt.h), against the oracle, bit for bit: every launch form's scalar-width instantiation.  tools/exp/odd_shapes.sh <name under gpurun_out>
out=gpurun_out/${1:-odd_shapes}.txt; : > $out
for ar in opencv legacy; do
  S="python tools/parity_survey.py --arith $ar --n 40"
  for wh in "112 80" "144 96" "80 48" "208 112" "176 144" "48 32" "16 16" "368 240" "104 72 4" "120 88 4" "72 40 4" "24 8 4"; do
    set -- $wh
    modes="fixed fixed5 reference"; [ -n "$3" ] && modes=fixed
    for mode in $modes; do for depth in 0 1; do
      $S --w $1 --h $2 --mode $mode --depth $depth >> $out 2>&1
    done; done
    $S --w $1 --h $2 --mode fixed --depth 1 --single 1 >> $out 2>&1
    [ -z "$3" ] && $S --w $1 --h $2 --mode reference --depth 0 --single 1 >> $out 2>&1
    $S --w $1 --h $2 --mode fixed --depth 1 --weights 2 >> $out 2>&1
    [ -z "$3" ] && $S --w $1 --h $2 --mode fixed5 --depth 0 --weights 1 >> $out 2>&1
    $S --w $1 --h $2 --mode fixed --depth 1 --sampler 1 >> $out 2>&1
    $S --w $1 --h $2 --mode fixed --depth 1 --sampler 1 --weights 2 >> $out 2>&1
    $S --w $1 --h $2 --mode fixed --depth 1 --intrinsics $(python -c "print('%g,%g,%g,%g' % (0.72*$1, 0.715*$1, $1/2-1.3, $2/2+0.7))") >> $out 2>&1
  done
done
python - $out <<'PY'
import re, sys
tot = bit = st = ok = it = 0
for l in open(sys.argv[1]):
    m = re.search(r"n=(\d+): bit-identical (\d+), status equal (\d+), status 0: (\d+), iterations equal (\d+)", l)
    if not m:
        if "Traceback" in l or "Error" in l: print(l.rstrip())
        continue
    n, b, s, o, i = map(int, m.groups())
    tot += n; bit += b; st += s; ok += o; it += i
    if b != n or s != n or i != o: print("DIFFERS:", l.rstrip())
print("%d alignments: %d poses bit-identical, %d statuses equal; %d with status 0, of which %d with equal iteration counts" % (tot, bit, st, ok, it))
PY
