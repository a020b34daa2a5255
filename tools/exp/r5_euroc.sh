#!/bin/bash
# round 5: the headline step at the reference's EUROC geometry (calibration/calibrationEUROC.xml) next to the square default
E=458.654,457.296,367.215,248.375
E736=458.654,457.296,359.215,248.375     # principal point moved by the 8 columns cropped on the left (752 -> 736)
out=gpurun_out/${1:-r5_euroc}; mkdir -p $out
run() { name=$1; shift; python bench.py --cpu-pairs 0 --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
d=json.load(open("$out/$name.json"))
r=d.get("roofline",{})
print("%-28s %9.1f /s  frac %s  per level GB/s %s  latency %s" % ("$name", d["value"], r.get("frac"), [e["algorithmic_GBs"] for e in r.get("per_level",[])], {k:v for k,v in (d.get("single_pair_latency") or {}).items() if k.endswith("_ms")}))
PY
}
run square_640x480_l4
run euroc_640x480_l4 --intrinsics $E
run euroc_640x480_l4_nodepth --intrinsics $E --no-depth
run square_640x480_l4_nodepth --no-depth
run euroc_736x480_l5_fixed --width 736 --intrinsics $E736 --levels 5 --no-depth --unique 64
run euroc_752x480_l5_fixed --width 752 --intrinsics $E --levels 5 --no-depth --unique 64
run euroc_736x480_refsched --width 736 --intrinsics $E736 --reference-schedule --no-depth --unique 64
run euroc_752x480_refsched --width 752 --intrinsics $E --reference-schedule --no-depth --unique 64
run square_640x480_refsched --reference-schedule --no-depth
