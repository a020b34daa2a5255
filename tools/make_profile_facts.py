#!/usr/bin/env python3
"""profiles/<round>/k_residual_facts.json from the round's counter summaries: the static facts about the dominant kernel that
bench.py quotes with their source (HBM bytes per pixel-iteration from the FETCH_SIZE / WRITE_SIZE passes, instruction
mix from the SQ passes).

usage: make_profile_facts.py <profiles dir>
"""
import csv
import json
import os
import sys


def per_dispatch(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if "k_residual" in r["kernel"] and r["counter"] == counter]
    return [(int(r["grid"]), float(r["mean_per_dispatch"]), int(r["dispatches"])) for r in rows]


def main(d):
    fetch = per_dispatch(os.path.join(d, "pmc_fetch_bench_default_p1024.csv"), "FETCH_SIZE")
    write = per_dispatch(os.path.join(d, "pmc_write_bench_default_p1024.csv"), "WRITE_SIZE")
    # KiB per dispatch, weighted by dispatch count = bytes of all residual launches of the run
    kib = sum(2.0 * m * n for _, m, n in fetch) + sum(m * n for _, m, n in write)   # x2: gfx950 FETCH_SIZE correction
    launches = sum(n for _, _, n in fetch)
    # levels the k_residual launches cover: 0..3 in rounds 1-2 (level 3 shows as a smaller grid); from round 3 the coarsest
    # level of a batch runs in k_coarse, and k_residual launches cover levels 0..2, all at one grid size
    grids = sorted({g for g, _, _ in fetch})
    n_lv = 4 if len(grids) > 1 else 3
    px_levels = 1024 * sum((640 >> l) * (480 >> l) for l in range(n_lv))
    pixels = px_levels * launches / float(n_lv)
    facts = {
        "hbm_bytes_per_pixel_iteration": round(kib * 1024.0 / pixels, 3),
        "hbm_bytes_source": d.rstrip("/") + "/pmc_fetch_bench_default_p1024.csv + pmc_write_bench_default_p1024.csv "
                            "(2 x FETCH_SIZE + WRITE_SIZE, KiB, separate --pmc passes of bench.py at its defaults)",
    }
    sq = os.path.join(d, "sq_counters_k_residual_level0_p1024.csv")
    if os.path.exists(sq):
        vals = {}
        for line in open(sq):
            if line.startswith("#") or line.startswith("counter"):
                continue
            name, _, _, per_px = line.strip().split(",")
            vals[name] = float(per_px)
        facts["valu_instructions_per_pixel"] = round(vals["SQ_INSTS_VALU"], 2)
        facts["f64_fma_per_pixel"] = round(vals["SQ_INSTS_VALU_FMA_F64"], 2)
        facts["instruction_mix_source"] = d.rstrip("/") + "/sq_counters_k_residual_level0_p1024.csv (SQ_INSTS_VALU, SQ_INSTS_VALU_FMA_F64 per lane-pixel, level-0 launches)"
    json.dump(facts, open(os.path.join(d, "k_residual_facts.json"), "w"), indent=1)
    print(json.dumps(facts, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
