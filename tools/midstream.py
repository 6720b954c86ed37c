#!/usr/bin/env python3
"""Throughput of a STREAM of mid-size batches: 4 096 proofs through rlnamd_prover_prove_stream in chunks of max_batch =
64 / 128 / 256 (c = 8 tables).  Run with RLNAMD_LANECHUNK=<threshold> to compare the small-batch shapes with the lanes =
proofs pipeline (profiles/r3_rocprof_summary.md section 10).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zerokit_amd import workload
from zerokit_amd.batch import BatchProver
out = {"RLNAMD_LANECHUNK": os.environ.get("RLNAMD_LANECHUNK")}
for mb in (64, 128, 256):
    p = BatchProver(max_batch=mb)
    inp, rsb = workload.config2_packed(p.slots, p.inputs_size, 0, 4096)
    p.prove_stream_raw(inp[:p.inputs_size*32*mb*2], rsb[:64*mb*2])
    t0 = time.perf_counter(); p.prove_stream_raw(inp, rsb); dt = time.perf_counter() - t0
    out["max_batch_%d_proofs_per_s" % mb] = round(4096 / dt, 1)
    p.close()
print(json.dumps(out))
