#!/usr/bin/env python3
"""Latency of one batch of n proofs with the small-batch shapes (lanes = chunks walks, lanes = nodes interpreter, early
walks ...) against the lanes = proofs pipeline: where is the crossover?  Run with RLNAMD_LANECHUNK=<threshold>
(0 = never small).  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zerokit_amd import workload  # noqa: E402
from zerokit_amd.batch import BatchProver  # noqa: E402

p = BatchProver(max_batch=int(os.environ.get("MAXB", "512")), window_bits=int(os.environ.get("WBITS", "0")))   # WBITS=7150114: the throughput tables (228 GiB)
out = {"RLNAMD_LANECHUNK": os.environ.get("RLNAMD_LANECHUNK", "default"), "table_gib": round(p.info.table_bytes / 2**30, 1)}
for n in (1, 4, 8, 16, 32, 64, 128, 192, 256, 384, 512):
    inp, rsb = workload.config2_packed(p.slots, p.inputs_size, 0, n)
    t, _ = p.submit(inp, rsb)
    p.collect_raw(t, n)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        t, _ = p.submit(inp, rsb)
        p.collect_raw(t, n)
        ts.append((time.perf_counter() - t0) * 1e3)
    out[n] = round(min(ts), 2)
print(json.dumps(out))
p.close()
