#!/usr/bin/env python3
"""Aggregates the rocprofv3 --pmc passes of `tools/gpu_profile.sh <tag> pmc` (counter_collection.csv, one directory per
pass) into profiles/rN_pmc_walks.json: per kernel the mean of every counter over its launches and how many launches one
batch makes of it, plus the figures bench.py derives its issue-slot view and `roofline.traffic` from.  The schedule and the hash of the walk's source files are recorded so that
bench.py only uses the file for the build and schedule it was taken on.

    python tools/pmc_collect.py <dir with pass sub-directories> <bench json line of one pass> <out.json>
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import walk_source_hash  # noqa: E402


def tag(name):
    if "k_msm29" in name and "G1Acc29" in name:
        return "k_msm29<G1>"
    if "k_msm29" in name and ("G2Acc" in name):
        return "k_msm29<G2>"
    for k in ("k_witness29", "k_ntt_pass", "k_matvec", "k_recode", "k_sum_ranges", "k_fin_smul", "k_proof_values",
              "k_v29_to_fr", "k_hquot", "k_fin_affine", "k_fin_out", "k_glv_fold"):
        if k in name:
            return k
    import re
    m = re.search(r"\b(k_[a-z0-9_]+)", name)    # every other kernel of the library (staging, wipes, ...)
    return m.group(1) if m else None


def main():
    top, bench_line, out = sys.argv[1:4]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(top, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            t = tag(r["Kernel_Name"])
            if t:
                agg[t][r["Counter_Name"]].append(float(r["Counter_Value"]))
    line = json.loads(open(bench_line).read().strip().splitlines()[-1])
    cfg = line["config"]
    B = cfg["chunk"]
    kernels = {}
    for k, cs in agg.items():
        kernels[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        kernels[k]["launches_seen"] = max(len(v) for v in cs.values())
    # additions per launch: rows x table additions per row x proofs (bench line: madd_per_s x launch time would be
    # circular; the row counts are properties of the shipped zkey: 23 675 finite G1 rows, 3 847 G2 rows)
    g1 = kernels.get("k_msm29<G1>", {})
    g2 = kernels.get("k_msm29<G2>", {})
    if g1:
        g1["lane_additions_per_launch"] = 23675 * cfg["windows"] * B
        g1["valu_per_wave_addition"] = round(g1.get("SQ_INSTS_VALU", 0) / (g1["lane_additions_per_launch"] / 64), 1)
        if "FETCH_SIZE" in g1 and "WRITE_SIZE" in g1:   # KB; FETCH_SIZE doubled (guide: gfx950 wide reads count half)
            g1["traffic_bytes_per_launch"] = int(2 * g1["FETCH_SIZE"] * 1024 + g1["WRITE_SIZE"] * 1024)
    if g2:
        g2["lane_additions_per_launch"] = 3847 * cfg["windows_g2"] * B
        g2["valu_per_wave_addition"] = round(g2.get("SQ_INSTS_VALU", 0) / (g2["lane_additions_per_launch"] / 64), 1)
        if "FETCH_SIZE" in g2 and "WRITE_SIZE" in g2:
            g2["traffic_bytes_per_launch"] = int(2 * g2["FETCH_SIZE"] * 1024 + g2["WRITE_SIZE"] * 1024)
    # launches of a kernel per 1024-proof batch (the G1 walk runs once per batch): lets a reader sum a counter over every
    # kernel of one batch
    per = g1.get("launches_seen", 0)
    for k in kernels.values():
        k["launches_per_batch"] = round(k["launches_seen"] / per, 3) if per else None
    doc = {"source": "tools/gpu_profile.sh <tag> pmc: rocprofv3 --pmc passes (counters only, one block per pass) of "
                     "`bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs`; means over the launches seen",
           "walk_source_hash": walk_source_hash(),
           "schedule": {"window_bits": cfg["window_bits"], "windows": cfg["windows"],
                        "window_bits_g2": cfg["window_bits_g2"], "windows_g2": cfg["windows_g2"],
                        "glv": 1 if cfg["glv"] else 0, "batch": B},
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, "kernels:", sorted(kernels))


if __name__ == "__main__":
    main()
