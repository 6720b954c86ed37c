#!/usr/bin/env python3
"""Two circuits resident on one device proving alternating 1 024-proof batches (bench.py --workload operating-points
measures the same beside the schedules; this is that leg by itself, for A/B runs): one JSON line."""
import collections
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from zerokit_amd import lib, workload  # noqa: E402
from zerokit_amd._native import check  # noqa: E402
from zerokit_amd.batch import BatchProver  # noqa: E402
import bench  # noqa: E402

B, K, nb = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 12, 6
a = BatchProver(max_batch=B, window_bits=int(os.environ.get("TWO_WB_A", "114")))
m = BatchProver(max_batch=B, window_bits=10, depth=20, multi=True)
packed = [workload.config2_packed(a.slots, a.inputs_size, B * k, B) for k in range(nb)]
named, rs = workload.circuit_range(0, 2 * B, 20, True)
mp = [(m.pack_named_inputs(named[k * B:(k + 1) * B]),
       b"".join(r.to_bytes(32, "little") + s_.to_bytes(32, "little") for r, s_ in rs[k * B:(k + 1) * B])) for k in range(2)]
bench._stream_proofs_per_s(a, packed, 2, B)
bench._stream_proofs_per_s(m, mp, 2, B)
alone_a, _ = bench._stream_proofs_per_s(a, packed, K, B)
alone_m, _ = bench._stream_proofs_per_s(m, mp, K, B)
out = {"alone_single": round(alone_a, 1), "alone_multi": round(alone_m, 1)}
for depth in (5, 3, 2, 1):
    qa, qm = collections.deque(), collections.deque()
    a.sync()
    m.sync()
    t1 = time.perf_counter()
    for j in range(K):
        if len(qa) == min(depth, a.n_slots()):
            t, kk = qa.popleft()
            a.collect_raw(t, B)
        qa.append((a.submit(*packed[j % nb])[0], j % nb))
        if len(qm) == min(depth, m.n_slots()):
            t, kk = qm.popleft()
            m.collect_raw(t, B)
        qm.append((m.submit(*mp[j % 2])[0], j % 2))
    while qa:
        a.collect_raw(qa.popleft()[0], B)
    while qm:
        m.collect_raw(qm.popleft()[0], B)
    a.sync()
    m.sync()
    out["alternating_in_flight_%d" % depth] = round(2 * K * B / (time.perf_counter() - t1), 1)
out["time_weighted_mix"] = round(2.0 / (1.0 / alone_a + 1.0 / alone_m), 1)
print(json.dumps(out))
a.close()
m.close()
