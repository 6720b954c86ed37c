#!/usr/bin/env python3
"""Where a single proof's wall time goes below the FFI: upload / run / download of the resident path against
submit / collect, n = 1 (needs the GPU).  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zerokit_amd import workload  # noqa: E402
from zerokit_amd.batch import BatchProver  # noqa: E402

p = BatchProver(max_batch=int(os.environ.get("MAXB", "64")))
N = int(os.environ.get("N", "1"))   # proofs per batch
ITER = int(os.environ.get("ITER", "8"))
inp, rsb = workload.config2_packed(p.slots, p.inputs_size, 0, N)
_, rs = workload.config2_range(0, N)
out = {"upload": [], "run": [], "download": [], "submit_collect": [], "stage_ms": None}
for _ in range(8):
    t0 = time.perf_counter()
    p.upload(inp, rs)
    t1 = time.perf_counter()
    p.run(N)
    t2 = time.perf_counter()
    p.download(N)
    t3 = time.perf_counter()
    out["upload"].append(round((t1 - t0) * 1e3, 3))
    out["run"].append(round((t2 - t1) * 1e3, 3))
    out["download"].append(round((t3 - t2) * 1e3, 3))
out["stage_ms"] = {k: round(v, 3) for k, v in p.stage_ms().items()}
for _ in range(ITER):
    t0 = time.perf_counter()
    t, n = p.submit(inp, rsb)
    p.collect_raw(t, n)
    out["submit_collect"].append(round((time.perf_counter() - t0) * 1e3, 3))
sc = sorted(out["submit_collect"])
out["submit_collect_median"] = sc[len(sc) // 2]
out["submit_collect_min"] = sc[0]
print(json.dumps(out))
