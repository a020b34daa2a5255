for P in 32 64 96 160 1024; do
  python bench.py --pairs $P --unique 32 --cpu-pairs 0 --steps 20 --warmup 5 > gpurun_out/mall_p$P.json 2> gpurun_out/mall_p$P.err
  python - <<PY
import json
d=json.load(open("gpurun_out/mall_p$P.json"))
r=d["roofline"]
print($P, d["value"], [ (e["level"], e["avg_ms_per_evaluation"], e["algorithmic_GBs"]) for e in r["per_level"]], r["valu"].get("shader_clock_GHz"), r["valu"].get("valu_issue_frac"))
PY
done
