#!/usr/bin/env python3
"""Same-box A/B of the three ways to push K batches of 1024 proofs through one prover:
  resident   rlnamd_prover_upload once + K x run_async (round 2's bench: inputs pre-resident, everything pre-enqueued)
  submit     K x rlnamd_prover_submit / collect from Python (bench.py's timed region)
  stream     ONE rlnamd_prover_prove_stream call over K x 1024 proofs (the host loop in C)
Prints one JSON line."""
import json
import os
import sys
import time
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
if os.environ.get("RLN_AB_TORCH"):      # what bench.py does under the driver: torch (and its bundled HIP runtime) first
    import torch
    if os.environ["RLN_AB_TORCH"] == "1":
        torch.cuda.set_device(0)
from zerokit_amd import workload  # noqa: E402
from zerokit_amd.batch import BatchProver  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 1024
p = BatchProver(max_batch=B, window_bits=int(os.environ.get("RLNAMD_WINDOW_BITS", "7150114")))
batches = [workload.config2_packed(p.slots, p.inputs_size, B * k, B) for k in range(K)]
_, rs0 = workload.config2_range(0, B)
out = {}
for rep in range(2):
    p.upload(batches[0][0], rs0)
    p.run(B)
    t0 = time.perf_counter()
    for _ in range(K):
        p.run_async(B)
    p.sync()
    out["resident_ms_per_batch_%d" % rep] = round((time.perf_counter() - t0) / K * 1e3, 3)
    print("resident done", out, file=sys.stderr, flush=True)

    n = p.n_slots()
    q = collections.deque()
    t0 = time.perf_counter()
    for k in range(K):
        if len(q) == n:
            p.collect_raw(q.popleft(), B)
        q.append(p.submit(*batches[k])[0])
    while q:
        p.collect_raw(q.popleft(), B)
    out["submit_ms_per_batch_%d" % rep] = round((time.perf_counter() - t0) / K * 1e3, 3)
    print("submit done", file=sys.stderr, flush=True)

    big_in = b"".join(b[0] for b in batches)
    big_rs = b"".join(b[1] for b in batches)
    t0 = time.perf_counter()
    p.prove_stream_raw(big_in, big_rs)
    out["stream_ms_per_batch_%d" % rep] = round((time.perf_counter() - t0) / K * 1e3, 3)
print(json.dumps(out))
