#!/usr/bin/env python3
"""Steady-state timeline of the throughput pipeline from a rocprofv3 --kernel-trace CSV of bench.py: for one window of
two steps, every kernel longer than 0.2 ms by stream (start -> end, ms relative to the window), and the walks' gaps.
    python tools/pipeline_timeline.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys


def tag(n):
    if "k_msm29" in n:
        return "G1walk" if "G1Acc29" in n else "G2walk"
    m = re.search(r"\b(k_[a-z0-9_]+)", n)
    return m.group(1) if m else n[:20]


rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag(r["Kernel_Name"]), r["Stream_Id"]) for r in rows]
g1 = [e for e in ev if e[2] == "G1walk"]
g2 = [e for e in ev if e[2] == "G2walk"]
for name, g in (("G1 walk", g1), ("G2 walk", g2)):
    prev, out = None, []
    for e in g[3:11]:
        out.append("%.1f (+%.1f idle)" % ((e[1] - e[0]) / 1e6, (e[0] - prev) / 1e6 if prev else 0.0))
        prev = e[1]
    print(name, "launch ms (stream idle before it):", ", ".join(out))
a, b = g1[6][0], g1[8][0]
print("window of two steps: %.2f ms" % ((b - a) / 1e6))
bys = collections.defaultdict(list)
for e in ev:
    if e[1] > a and e[0] < b:
        bys[e[3]].append(e)
for s, l in sorted(bys.items()):
    print("stream", s)
    for e in sorted(l):
        if e[1] - e[0] > 2e5:
            print("    %-16s %8.2f -> %8.2f  (%.2f ms)" % (e[2], (e[0] - a) / 1e6, (e[1] - a) / 1e6, (e[1] - e[0]) / 1e6))
