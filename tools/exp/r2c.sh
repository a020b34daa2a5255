#!/bin/bash
# round-2 A/B: parity tests, level-0 and full bench, slicing sweep
mkdir -p gpurun_out/r2c
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2c/gputest.txt 2>&1; tail -3 gpurun_out/r2c/gputest.txt
for tb in 4096 2048 1024; do
  UWT_TARGET_BLOCKS=$tb python bench.py --cpu-pairs 0 --steps 6 > gpurun_out/r2c/bench_tb$tb.json 2>/dev/null
done
python bench.py --levels 1 --pairs 1024 --steps 3 --warmup 1 --cpu-pairs 0 > gpurun_out/r2c/bench_l0.json 2>/dev/null
python bench.py > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2c/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d.get("parity"))
    except Exception as e: print(f, "ERR", e)
PY
