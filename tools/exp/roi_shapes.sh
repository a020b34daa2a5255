#!/bin/bash
# Whole alignments at frame sizes the reference's own EUROC pipeline produces — the data-dependent ROI crop of src/System.cpp:148-191,
# odd and not divisible by 2^(levels-1): cvRound level sizes (both half-way roundings), partial last columns / rows of cv::resize, point
# grids (size >> lvl) smaller than their images — against the oracle, bit for bit: the RAGGED instantiation of every launch form.
# Companion of odd_shapes.sh (sizes divisible by 2^(levels-1) whose coarse levels are not whole 4-pixel groups).
#   tools/exp/roi_shapes.sh <name under gpurun_out> [alignments per case, default 24]
out=gpurun_out/${1:-roi_shapes}.txt; : > $out
N=${2:-24}
for ar in opencv legacy; do
  S="python tools/parity_survey.py --arith $ar --n $N"
  for wh in "725 465" "733 471" "735 479" "163 99" "161 97" "165 101" "166 98" "167 103" "154 101" "91 57 4" "77 45 4" "39 23 4"; do
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
    $S --w $1 --h $2 --mode fixed --depth 0 --tuning typed_loads=0 >> $out 2>&1
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
