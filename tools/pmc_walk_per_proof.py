#!/usr/bin/env python3
"""Per-proof counters of the two walks from rocprofv3 --pmc passes at different batch sizes (tools/ab_batch_size.sh):
   python tools/pmc_walk_per_proof.py <dir> <batch> [<dir> <batch> ...]
prints, for k_msm29<G1> / <G2>, every counter of the pass divided by the proofs of a launch."""
import collections
import csv
import glob
import json
import os
import sys

out = {}
args = sys.argv[1:]
for top, batch in zip(args[0::2], args[1::2]):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(top, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "k_msm29" not in n:
                continue
            t = "k_msm29<G1>" if "G1Acc29" in n else "k_msm29<G2>"
            agg[t][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[batch] = {k: dict({c: round(sum(v) / len(v) / int(batch), 1) for c, v in cs.items()}, launches=max(len(v) for v in cs.values()))
                  for k, cs in agg.items()}
print(json.dumps(out, indent=1))
