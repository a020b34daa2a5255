#!/bin/bash
# round 6: the fixed 5 x 10 step at ROI-like frame sizes (what System::CalculateROI leaves of a 752 x 480 EUROC frame: odd, data-dependent;
# src/System.cpp:148-191) next to the 736 x 480 line, per pixel.  tools/exp/r6_roi.sh <name under gpurun_out> [extra bench args]
E=458.654,457.296,359.215,248.375
out=gpurun_out/${1:-r6_roi}; mkdir -p $out; shift
run() { name=$1; shift; python bench.py --cpu-pairs 0 --steps 10 --warmup 3 --no-profile --unique 16 --levels 5 --no-depth --intrinsics $E "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json"))
    import re; w,h=map(int,re.search(r"(\d+)x(\d+)", d["metric"]).groups())
    px=sum((w>>l)*(h>>l) for l in range(5))
    print("%-22s %9.1f /s  %7.3f ms/step  %8.2f Gpx-iter/s  parity %s" % ("$name", d["value"], d["ms_per_step"], d["value"]*px*10/1e9, d.get("parity",{}).get("bit_identical")))
except Exception as e:
    print("$name", "FAILED", e); print(open("$out/$name.err").read()[-800:])
PY
}
run w736x480 --width 736 --height 480 "$@"
run w725x465 --width 725 --height 465 "$@"
run w733x471 --width 733 --height 471 "$@"
run w735x479 --width 735 --height 479 "$@"
