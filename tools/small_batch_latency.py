#!/usr/bin/env python3
"""submit + collect of n = 1 .. 8 proofs per call (median of 9) on the default tables; RLNAMD_HINTS decides up to which n
the witness graph is interpreted as segments behind hints.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from zerokit_amd import workload  # noqa: E402
from zerokit_amd.batch import BatchProver  # noqa: E402

p = BatchProver(max_batch=64)
out = {"hints": os.environ.get("RLNAMD_HINTS", "default")}
for n in (1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 48, 64):
    inp, rsb = workload.config2_packed(p.slots, p.inputs_size, 100, n)
    ts = []
    for i in range(11):
        if os.environ.get("COLD"):   # other members on every call: their chains of hints are hashed, not remembered
            inp, rsb = workload.config2_packed(p.slots, p.inputs_size, 1000 + 64 * i + 1000 * n, n)
        t0 = time.perf_counter()
        t, _ = p.submit(inp, rsb)
        p.collect_raw(t, n)
        ts.append((time.perf_counter() - t0) * 1e3)
    out["n=%d" % n] = round(sorted(ts[2:])[4], 3)
out["stats"] = p.hint_stats()
print(json.dumps(out))
p.close()
