#!/usr/bin/env python3
"""ffi_generate_rln_proof from T threads on ONE object, one member: calls per second with the calls gathered into batches
(the default), with {"gather_calls": 0} (every call its own batch, one after the other: what the object did before) and
with {"auto_partial": 2} (a remembered member's proofs are finishes through the cone: its gathered calls are batches of finishes).
One JSON line."""
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zerokit_amd import hashers  # noqa: E402
from zerokit_amd.public import RLN, RLNWitnessInput  # noqa: E402

out = {}
for name, cfg in (("gathered", None), ("one_call_at_a_time", {"gather_calls": 0}), ("auto_partial", {"auto_partial": 2})):
    path = None
    if cfg:
        f = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump(cfg, f)
        f.close()
        path = f.name
    rln = RLN(20, tree_config=path) if path else RLN(20)
    secret = hashers.hash_to_field_le(b"thread-bench-member")
    rln.set_leaf(7, hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 100))
    e, b = rln.get_merkle_proof(7)
    for i in range(4):   # warm: the member is remembered
        rln.generate_rln_proof(RLNWitnessInput(secret, 100, i, e, b, 100 + i, 4242))
    res = {}
    for T in (1, 2, 4, 8, 16, 32, 64):
        calls = 150 if T <= 8 else 60

        def work(tid):
            for j in range(calls):
                rln.generate_rln_proof(RLNWitnessInput(secret, 100, (tid + j) % 100, e, b, 1 + tid * 1000 + j, 4242))
        ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        res[str(T)] = {"calls_per_s": round(T * calls / dt, 1), "ms_per_call_per_thread": round(dt / calls * 1e3, 3)}
    out[name] = res
    if name in ("gathered", "one_call_at_a_time"):   # the same with finishes of the member's partial proof
        from zerokit_amd.public import RLNPartialWitnessInput
        part = rln.generate_partial_zk_proof(RLNPartialWitnessInput(secret, 100, e, b))
        fres = {}
        for T in (1, 2, 4, 8, 16, 32, 64):
            calls = 150 if T <= 8 else 60

            def fwork(tid):
                for j in range(calls):
                    rln.finish_rln_proof(part, RLNWitnessInput(secret, 100, (tid + j) % 100, e, b, 1 + tid * 1000 + j, 4242))
            ths = [threading.Thread(target=fwork, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            dt = time.perf_counter() - t0
            fres[str(T)] = {"calls_per_s": round(T * calls / dt, 1), "ms_per_call_per_thread": round(dt / calls * 1e3, 3)}
        out[name + "_finishes"] = fres
    if name == "gathered":
        out["gather_stats"] = rln.gather_stats()
    if name == "auto_partial":
        out["memo"] = rln.memo_stats()
    rln.close()
print(json.dumps(out))
