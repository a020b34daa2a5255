#!/bin/bash
# round 5: the general path's bench lines (robust weights / bilinear sampler), optionally with the residual plane off: r5_general.sh <out> [plane]
out=gpurun_out/${1:-r5_general}; mkdir -p $out
run() { name=$1; shift; python bench.py --cpu-pairs 0 --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
d=json.load(open("$out/$name.json"))
r=d.get("roofline",{})
print("%-24s %9.1f /s  frac %s  per level ms/eval %s" % ("$name", d["value"], r.get("frac"), [e["avg_ms_per_evaluation"] for e in r.get("per_level",[])]))
PY
}
run huber_p1024 --weights huber --unique 16
run huber_p256 --weights huber --pairs 256 --unique 16
run tukey_p256 --weights tukey --pairs 256 --unique 16
run bilinear_huber_p256 --weights huber --bilinear --pairs 256 --unique 16
run bilinear_p256 --bilinear --pairs 256 --unique 16
