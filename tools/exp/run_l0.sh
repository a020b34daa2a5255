#!/bin/bash
# usage: run_l0.sh <label> [env assignments...]   -- level-0-only residual timing via bench.py
label=$1; shift
out=$(env "$@" python bench.py --levels 1 --pairs 1024 --steps 3 --warmup 1 --cpu-pairs 0 2>/dev/null | tail -1)
python - "$label" "$out" <<'PY'
import sys, json
d=json.loads(sys.argv[2]); r=d.get("roofline",{})
print("%-28s value %9.0f/s  k_residual avg %.4f ms  achieved %.0f GB/s" % (sys.argv[1], d["value"], r.get("avg_launch_ms",0), r.get("achieved",0)))
PY
