#!/usr/bin/env python3
"""Timeline of ONE single-proof run from a rocprofv3 kernel trace (kernel_trace.csv): every kernel of the last
k_witness_lanes .. k_fin_out window with its start offset, duration and the gap since the previous kernel ended.

    python tools/single_timeline.py <kernel_trace.csv> [first kernel of the window, default k_witness_lanes]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"rlnamd::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)", name)
    tag = m.group(1) if m else name
    if "k_msm29" in name:
        tag += "<G2>" if ("G2Acc" in name) else "<G1>"
    if "k_sum_tree" in name or "k_sum_ranges" in name:
        tag += "<Fq2>" if "Fp2" in name or "Fq2" in name else "<Fq>"
    return tag


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
first = sys.argv[2] if len(sys.argv) > 2 else "k_witness_lanes"
starts = [i for i, r in enumerate(rows) if r[2].startswith(first)]
if not starts:
    sys.exit("no %s launch in the trace" % first)
i0 = starts[-1]
# the run may begin a few kernels earlier (k_stage_in, k_proof_values)
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 200_000 and rows[i0 - 1][2] in ("k_stage_in", "k_proof_values"):
    i0 -= 1
t0 = rows[i0][0]
last_end = t0
print("%-22s %9s %9s %9s" % ("kernel", "start_ms", "dur_ms", "gap_ms"))
for s, e, n in rows[i0:]:
    print("%-22s %9.3f %9.3f %9.3f" % (n, (s - t0) / 1e6, (e - s) / 1e6, (s - last_end) / 1e6))
    last_end = max(last_end, e)
    if n == "k_fin_out" or (n.startswith("k_wipe") and first != "k_witness_lanes"):
        break
print("total %.3f ms" % ((last_end - t0) / 1e6))
