#!/usr/bin/env python3
"""Single-leaf and scattered updates of the HBM-resident depth-20 tree (rlnamd_tree_set_leaves + rlnamd_tree_root):
wall time per call; run under `rocprofv3 --kernel-trace` to see the pass's kernels.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zerokit_amd.batch import PoseidonTree  # noqa: E402

t = PoseidonTree(20)
t.fill_sequential(0, 1 << 20, 1)
t.root()
out = {}
for k in (1, 4, 8, 10, 12, 14, 16, 64, 1000):
    ts = []
    for r in range(12):
        ups = [((r * 7919 + j * 104729) % (1 << 20), 5 + r + j) for j in range(k)]
        t0 = time.perf_counter()
        t.set_leaves(ups)
        t.root()
        ts.append((time.perf_counter() - t0) * 1e3)
    out["k=%d" % k] = round(sorted(ts)[len(ts) // 2], 3)
print(json.dumps(out))
