#!/usr/bin/env python3
"""Condenses rocprofv3 output (kernel_stats.csv from --kernel-trace --stats, counter_collection.csv from
--pmc passes) into the markdown summary committed under profiles/."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"rlnamd::", "", name)
    name = re.sub(r"Fp<FqParams>\s*", "Fq", name)
    name = re.sub(r"Fp<FrParams>\s*", "Fr", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_]+(?:<[^(]*>)?)\(", name)
    return (m.group(1) if m else name)[:48]


def stats(path):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | calls | avg ms | total ms | % |", "|---|---|---|---|---|"]
    for r in rows:
        out.append("| `%s` | %s | %.3f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6,
                                                       float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    return "\n".join(out)


def pmc(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    out = ["| kernel | launches | mean %s (KB) | mean GB |" % counter, "|---|---|---|---|"]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]))[:12]:
        m = sum(v) / len(v)
        out.append("| `%s` | %d | %.0f | %.3f |" % (k, len(v), m, m * 1024 / 1e9))
    return "\n".join(out)


if __name__ == "__main__":
    kind, path = sys.argv[1], sys.argv[2]
    print(stats(path) if kind == "stats" else pmc(path, sys.argv[3]))
