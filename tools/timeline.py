#!/usr/bin/env python3
"""Pipeline timeline from a rocprofv3 kernel trace (kernel_trace.csv of `rocprofv3 --kernel-trace --output-format csv`):
for every MSM launch prints when it started, how long it ran, and how long the MSM stream sat idle before it; then
the busy / idle totals of the MSM stream over the steady-state part of the run.

    python tools/timeline.py <kernel_trace.csv> [skip_first_n_batches]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"rlnamd::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)", name)
    tag = m.group(1) if m else name
    if "k_msm29" in name:
        tag += "<G2>" if "G2Acc29" in name else "<G1>"
    elif "k_sum_ranges" in name or "k_table" in name:
        tag += "<Fq2>" if "Fq2" in name or "Fp2" in name else "<Fq>"
    return tag


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                     r.get("Queue_Id", "?")))
    rows.sort()
    msm = [r for r in rows if r[2].startswith("k_msm29")]
    if not msm:
        print("no k_msm29 launches in the trace")
        return
    t0 = msm[0][0]
    print("# MSM stream: launch, start (ms since the first MSM), duration, idle gap before it")
    prev_end = None
    gaps, durs = [], []
    g1_starts = []
    for i, (s, e, name, q) in enumerate(msm):
        gap = (s - prev_end) / 1e6 if prev_end is not None else 0.0
        print("%-12s q%-3s start %9.3f  dur %7.3f  gap %6.3f" % (name, q, (s - t0) / 1e6, (e - s) / 1e6, gap))
        if name.endswith("<G1>"):
            g1_starts.append(s)
        if i >= 2 * skip:
            gaps.append(gap)
            durs.append((e - s) / 1e6)
        prev_end = e
    nb = len(durs) / 2
    if nb >= 1:
        print("\nsteady state over %.0f batches: busy %.2f ms/batch, idle %.2f ms/batch" %
              (nb, sum(durs) / nb, sum(gaps) / nb))
    if len(g1_starts) > skip + 1:
        per = [(b - a) / 1e6 for a, b in zip(g1_starts[skip:], g1_starts[skip + 1:])]
        print("G1 start-to-start period: mean %.2f ms (min %.2f, max %.2f)" % (sum(per) / len(per), min(per), max(per)))
    # what ran inside the idle gaps of the MSM stream
    print("\n# kernels overlapping the idle gaps (steady state)")
    prev_end = None
    for i, (s, e, name, q) in enumerate(msm):
        if prev_end is not None and i >= 2 * skip and s - prev_end > 300_000:
            inside = {}
            for (ks, ke, kn, kq) in rows:
                if kn.startswith("k_msm29"):
                    continue
                ov = min(ke, s) - max(ks, prev_end)
                if ov > 0:
                    inside[kn] = inside.get(kn, 0) + ov / 1e6
            top = sorted(inside.items(), key=lambda kv: -kv[1])[:6]
            print("gap %6.3f ms before %s @%9.3f: %s" % ((s - prev_end) / 1e6, name, (s - t0) / 1e6,
                                                      ", ".join("%s %.2f" % kv for kv in top)))
        prev_end = e


if __name__ == "__main__":
    main()
