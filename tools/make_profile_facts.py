#!/usr/bin/env python3
"""profiles/<round>/k_residual_facts.json from the round's counter summaries and the compiler's resource table: the facts about
the dominant kernel that bench.py quotes with their source, per arithmetic set (opencv / legacy), stamped with the source id
(uwt_source_id(): sha256 of sources + flags) of the libuwt_hip.so they were collected on — bench.py quotes them only while the
library it loaded reports the same id (the binary itself is not byte-reproducible).

  HBM bytes per pixel-iteration   pmc_fetch_<set>_bench_default_p1024.csv + pmc_write_… (2 x FETCH_SIZE + WRITE_SIZE, KiB)
  instruction mix                 sq_counters_k_residual_<set>_level0_p1024.csv (SQ_INSTS_VALU, SQ_INSTS_VALU_FMA_F64 per pixel)
  registers / occupancy / LDS     kernel_resources.md (tools/kernel_resources.py), the production instantiation of the set

usage: make_profile_facts.py <profiles dir> [library.so]
"""
import csv
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (the last argument, 3: typed + non-temporal plane loads — what a batch larger than the caches runs at its fine levels)
PRODUCTION = {"opencv": "k_residual<0, 4, true, true, false, double, true, 0, 0, false, 3>",
              "legacy": "k_residual<1, 4, true, true, false, double, true, 0, 0, false, 3>"}


def per_dispatch(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if "k_residual" in r["kernel"] and r["counter"] == counter]
    return [(int(r["grid"]), float(r["mean_per_dispatch"]), int(r["dispatches"])) for r in rows]


def set_facts(d, name):
    facts = {}
    f_fetch = os.path.join(d, "pmc_fetch_%s_bench_default_p1024.csv" % name)
    f_write = os.path.join(d, "pmc_write_%s_bench_default_p1024.csv" % name)
    if os.path.exists(f_fetch) and os.path.exists(f_write):
        fetch, write = per_dispatch(f_fetch, "FETCH_SIZE"), per_dispatch(f_write, "WRITE_SIZE")
        # KiB per dispatch, weighted by dispatch count = bytes of all residual launches of the run
        kib = sum(2.0 * m * n for _, m, n in fetch) + sum(m * n for _, m, n in write)   # x2: gfx950 FETCH_SIZE correction
        launches = sum(n for _, _, n in fetch)
        # the coarsest level of a batch runs in k_coarse when that kernel appears in the trace: k_residual launches then cover
        # levels 0..2 (an equal number of launches each, whatever their grid sizes), else 0..3
        coarse = any("k_coarse" in r["kernel"] for r in csv.DictReader(open(f_fetch)))
        n_lv = 3 if coarse else 4
        pixels = 1024 * sum((640 >> l) * (480 >> l) for l in range(n_lv)) * launches / float(n_lv)
        facts["hbm_bytes_per_pixel_iteration"] = round(kib * 1024.0 / pixels, 3)
        facts["hbm_bytes_source"] = ("%s/pmc_fetch_%s_bench_default_p1024.csv + pmc_write_%s_bench_default_p1024.csv (2 x FETCH_SIZE + "
                                     "WRITE_SIZE, KiB, separate --pmc passes of bench.py --arith %s at its defaults)"
                                     % (d.rstrip("/"), name, name, name))
    sq = os.path.join(d, "sq_counters_k_residual_%s_level0_p1024.csv" % name)
    if os.path.exists(sq):
        vals = {}
        for line in open(sq):
            if line.startswith("#") or line.startswith("counter"):
                continue
            c, _, _, per_px = line.strip().split(",")
            vals[c] = float(per_px)
        facts["valu_instructions_per_pixel"] = round(vals["SQ_INSTS_VALU"], 2)
        facts["f64_fma_per_pixel"] = round(vals["SQ_INSTS_VALU_FMA_F64"], 2)
        facts["instruction_mix_source"] = ("%s/sq_counters_k_residual_%s_level0_p1024.csv (SQ_INSTS_VALU, SQ_INSTS_VALU_FMA_F64 per "
                                           "lane-pixel, level-0 launches)" % (d.rstrip("/"), name))
    res = os.path.join(d, "kernel_resources.md")
    if os.path.exists(res):
        for line in open(res):
            if "`%s`" % PRODUCTION[name] in line:
                cells = [c.strip() for c in line.strip().strip("|").split("|")]
                facts["vgprs"], facts["lds_bytes_per_block"], facts["waves_per_simd"] = int(cells[1]), int(cells[5]), int(cells[6])
                facts["resource_source"] = "%s/kernel_resources.md (%s)" % (d.rstrip("/"), PRODUCTION[name])
    return facts


def main(d, lib):
    import ctypes
    h = ctypes.CDLL(lib)
    h.uwt_source_id.restype = ctypes.c_char_p
    out = {"library_source_id": h.uwt_source_id().decode(),   # sha256 of the sources + flags the library was built from
           "library": os.path.relpath(lib, ROOT),
           "sets": {name: set_facts(d, name) for name in ("opencv", "legacy")}}
    json.dump(out, open(os.path.join(d, "k_residual_facts.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "uw-slam_amd", "libuwt_hip.so"))
