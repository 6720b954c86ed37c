#!/usr/bin/env python3
"""One finish per call (finish_zk_proof_with_rs from a cached partial proof) on a prover of the given schedule: the
partial run keeps its known stored values on the device (rlnamd_prover_collect_partial_cached), every finish is one
rlnamd_prover_submit_finish + collect.  Prints one JSON line; run under `rocprofv3 --kernel-trace` for the timeline
(tools/single_timeline.py <csv> k_cone_restore).

    python tools/finish_latency.py [calls] [window_bits]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from zerokit_amd import workload  # noqa: E402
from zerokit_amd.batch import BatchProver  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
wb = int(sys.argv[2]) if len(sys.argv) > 2 else 120010
p = BatchProver(max_batch=64, window_bits=wb)
ws, rs = workload.config2_range(0, 1)
inp, rsb = p.pack_inputs(ws), p.pack_rs(rs)
t, _ = p.submit(p.pack_inputs([dict(ws[0], message_id=0, x=0, external_nullifier=0)]), bytes(64), 1)
pp, hs, _ = p.collect_partial_cached(t, 1)
full = p.prove(ws, rs)[0]["proof"]
out = {}
for label, h in (("whole_graph", [0]), ("cone", hs)):
    ts = []
    for i in range(calls):
        t0 = time.perf_counter()
        t, _ = p.submit_finish(inp, rsb, pp, h)
        pr, _, er = p.collect_raw(t, 1)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert pr[:128] == full and not any(er)
    ts = sorted(ts[4:])
    out[label] = {"ms_median": round(ts[len(ts) // 2], 3), "ms_min": round(ts[0], 3)}
ts = []
for i in range(calls):
    t0 = time.perf_counter()
    t, _ = p.submit(inp, rsb)
    p.collect_raw(t, 1)
    ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[4:])
out["full_proof"] = {"ms_median": round(ts[len(ts) // 2], 3), "ms_min": round(ts[0], 3)}
# the last call of the run is a cone finish: the one the timeline tool looks at
t, _ = p.submit_finish(inp, rsb, pp, hs)
p.collect_raw(t, 1)
out["cache"] = p.partial_cache_info()
out["window_bits"] = wb
print(json.dumps(out))
p.close()
