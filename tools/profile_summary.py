#!/usr/bin/env python3
"""One line per bench configuration of a profiles/<round>/ directory (the `…_no_profiler.json` lines): alignments/s, roofline.frac,
per-level GB/s, the other arithmetic set, one-pair latencies — what DESIGN.md §5 and profiles/<round>/README.md quote.
usage: profile_summary.py profiles/r05"""
import glob, json, os, sys
d0 = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05"
files = sorted(glob.glob(os.path.join(d0, "bench_*_no_profiler.json"))) + [f for f in (os.path.join(d0, "bench_total8192_g1.json"), os.path.join(d0, "bench_default_p1024_final.json")) if os.path.exists(f)]
for f in files:
    d = json.load(open(f))
    r = d.get("roofline") or {}
    a = d.get("arith_sets") or {}
    lat = d.get("single_pair_latency") or {}
    st = d.get("streaming") or {}
    other = {k: v["value"] for k, v in a.items() if isinstance(v, dict)}
    print("%-44s %9.1f/s  frac %-6s  GB/s %s" % (os.path.basename(f)[6:-5].replace("_no_profiler", ""), d["value"], r.get("frac"),
          " / ".join("%.0f" % e["algorithmic_GBs"] for e in r.get("per_level", []) if e.get("algorithmic_GBs"))))
    print("    sets %s | one pair %s / %s ms | streaming %s (%s of PCIe) | cpu %s / %s | parity %s/%s | valu %s clock %s GHz"
          % (other, lat.get("bench_schedule_ms"), lat.get("reference_schedule_ms"), st.get("value"), st.get("frac_of_pcie_bound"),
             (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline_all_cores") or {}).get("value"),
             (d.get("parity") or {}).get("bit_identical_poses"), (d.get("parity") or {}).get("pairs"),
             (r.get("valu") or {}).get("frac_of_valu_peak"), (r.get("valu") or {}).get("shader_clock_GHz")))
