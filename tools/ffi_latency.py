#!/usr/bin/env python3
"""What a caller of the zerokit C FFI (include/rln.h) sees on the GPU box.

  python tools/ffi_latency.py            one-at-a-time calls: single proof, verify, small batches, a leaf update
  python tools/ffi_latency.py --batch N  N proofs (default 8192) through ffi_generate_rln_proofs_batch on an object
                                         created with a config_path JSON carrying {"window_bits", "max_batch"} (the
                                         bench schedule), against rlnamd_prover_prove_stream on the same tables;
                                         element 0 is the golden (44, 77) case and must give the golden proof bytes
Prints JSON lines."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zerokit_amd import hashers, workload  # noqa: E402
from zerokit_amd.public import RLN, RLNWitnessInput  # noqa: E402


def one_at_a_time():
    rln = RLN(20)
    secret = 1234567
    rln.set_leaf(3, hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 100))
    elems, bits = rln.get_merkle_proof(3)
    ws = [RLNWitnessInput(secret, 100, i % 100, elems, bits, 1000 + i, 777) for i in range(64)]
    p = rln.generate_rln_proof(ws[0])
    out = {"single_proof_ms": []}
    for _ in range(5):
        t = time.perf_counter()
        p = rln.generate_rln_proof(ws[1])
        out["single_proof_ms"].append(round((time.perf_counter() - t) * 1e3, 2))
    t = time.perf_counter()
    ok = rln.verify_rln_proof(p, 1001)
    out["verify_ms"] = round((time.perf_counter() - t) * 1e3, 2)
    out["verify_ok"] = bool(ok)
    for n in (8, 64):
        t = time.perf_counter()
        rln.generate_rln_proofs_batch(ws[:n])
        dt = time.perf_counter() - t
        out["batch_%d_ms" % n] = round(dt * 1e3, 2)
    # the partial-proof split through the drop-in boundary (ffi_generate_partial_zk_proof / ffi_finish_rln_proof): the
    # partial proof object carries the prover's cache handle, a partial proof that came in as bytes does not
    from zerokit_amd.public import RLNPartialProof, RLNPartialWitnessInput
    pw = RLNPartialWitnessInput(secret, 100, elems, bits)
    pp = rln.generate_partial_zk_proof(pw)
    pp_wire = RLNPartialProof.from_bytes_le(pp.to_bytes_le())
    for label, obj in (("finish_ms", pp), ("finish_from_deserialised_partial_ms", pp_wire)):
        ts = []
        for i in range(9):
            t = time.perf_counter()
            q = rln.finish_rln_proof(obj, ws[2 + i])
            ts.append(round((time.perf_counter() - t) * 1e3, 3))
        out[label] = sorted(ts[2:])[len(ts[2:]) // 2]
        out[label.replace("_ms", "_verifies")] = bool(rln.verify_rln_proof(q, 1000 + 2 + 8))
    ts = []
    for i in range(5):
        t = time.perf_counter()
        rln.generate_partial_zk_proof(pw)
        ts.append(round((time.perf_counter() - t) * 1e3, 3))
    out["generate_partial_ms"] = sorted(ts)[len(ts) // 2]
    t = time.perf_counter()
    rln.set_leaf(5, 99)
    rln.get_root()
    out["set_leaf_plus_root_ms"] = round((time.perf_counter() - t) * 1e3, 3)
    # the same split behind the UNCHANGED ffi_generate_rln_proof: an object built with {"auto_partial": N} remembers its
    # members' partial proofs (first proof of a member at a root from scratch, later ones finishes)
    with tempfile.TemporaryDirectory() as d:
        cfg = os.path.join(d, "cfg.json")
        open(cfg, "w").write(json.dumps({"auto_partial": 4}))
        memo = RLN(20, tree_config=cfg)
    memo.set_leaf(3, hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 100))
    ts = []
    for i in range(12):
        t = time.perf_counter()
        q = memo.generate_rln_proof(ws[10 + i])
        ts.append(round((time.perf_counter() - t) * 1e3, 3))
    out["auto_partial_first_proof_of_a_member_ms"] = ts[0]
    out["auto_partial_repeat_member_ms"] = sorted(ts[2:])[len(ts[2:]) // 2]
    out["auto_partial_verifies"] = bool(memo.verify_rln_proof(q, 1000 + 10 + 11))
    out["auto_partial_stats"] = memo.memo_stats()
    del memo
    print(json.dumps(out))
    # host-side batch verification (rlnamd_verify_many_with_zkey: no GPU involved), golden proofs repeated
    from zerokit_amd.batch import verify_many_with_zkey
    z = open(os.path.join(ROOT, "zerokit_amd", "resources", "tree_depth_20", "rln_final.arkzkey"), "rb").read()
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]
    reps = 512 // len(cases)
    proofs = [bytes.fromhex(c["proof_compressed"]) for c in cases] * reps
    pubs = [[int(v) for v in c["public_inputs"]] for c in cases] * reps
    for threads in (1, 0):
        t = time.perf_counter()
        ok = verify_many_with_zkey(z, proofs, pubs, threads=threads)
        dt = time.perf_counter() - t
        print(json.dumps({"verify_many": len(proofs), "threads": threads or os.cpu_count(),
                          "verifications_per_s": round(len(proofs) / dt), "all_ok": all(ok)}))


def batch(n, window_bits, max_batch):
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]
    gold = next(c for c in cases if c["name"] == "survey_appendix_d")
    gw = gold["witness"]
    ws, rs = workload.config2_range(0, n)
    ws[0] = dict(identity_secret=int(gw["identity_secret"]), user_message_limit=int(gw["user_message_limit"]),
                 message_id=int(gw["message_id"]), path_elements=[int(t) for t in gw["path_elements"]],
                 identity_path_index=[int(t) for t in gw["identity_path_index"]], x=int(gw["x"]),
                 external_nullifier=int(gw["external_nullifier"]))
    rs[0] = (int(gold["r"]), int(gold["s"]))
    with tempfile.TemporaryDirectory() as d:
        cfg = os.path.join(d, "rln_config.json")
        open(cfg, "w").write(json.dumps({"window_bits": window_bits, "max_batch": max_batch}))
        t0 = time.time()
        rln = RLN(20, tree_config=cfg)
        init_s = time.time() - t0
    info = rln.prover_info()
    wi = [RLNWitnessInput(w["identity_secret"], w["user_message_limit"], w["message_id"], w["path_elements"],
                          w["identity_path_index"], w["x"], w["external_nullifier"]) for w in ws]
    rln.generate_rln_proofs_batch(wi[:2 * max_batch + 1], rs[:2 * max_batch + 1])      # warm-up
    times = []
    for _ in range(3):
        t = time.perf_counter()
        proofs = rln.generate_rln_proofs_batch(wi, rs)
        times.append(time.perf_counter() - t)
    ffi_s = min(times)
    raw0 = proofs[0].to_bytes_le()
    golden_ok = gold["proof_compressed"] in raw0.hex()
    del proofs, rln
    # the extension API on the same tables
    from zerokit_amd.batch import BatchProver
    p = BatchProver(max_batch=max_batch, window_bits=window_bits)
    inp, rsb = p.pack_inputs(ws), p.pack_rs(rs)
    p.prove_stream_raw(inp[:p.inputs_size * 32 * (2 * max_batch + 1)], rsb[:64 * (2 * max_batch + 1)])
    times = []
    for _ in range(3):
        t = time.perf_counter()
        pr, _, _ = p.prove_stream_raw(inp, rsb)
        times.append(time.perf_counter() - t)
    ext_s = min(times)
    same = pr[:128].hex() == gold["proof_compressed"]
    p.close()
    print(json.dumps({"n": n, "config": {"window_bits": window_bits, "max_batch": max_batch},
                      "prover_behind_ffi": {"capacity": int(info.capacity), "windows": int(info.windows),
                                            "windows_g2": int(info.windows_g2),
                                            "table_gib": round(info.table_bytes / 2**30, 2), "init_s": round(init_s, 2)},
                      "ffi_generate_rln_proofs_batch_proofs_per_s": round(n / ffi_s, 1),
                      "rlnamd_prover_prove_stream_proofs_per_s": round(n / ext_s, 1),
                      "ffi_over_extension": round(ext_s / ffi_s, 4),
                      "golden_44_77_bytes_through_ffi_batch": bool(golden_ok),
                      "golden_44_77_bytes_through_prove_stream": bool(same)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--batch":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
        batch(n, int(os.environ.get("RLNAMD_WINDOW_BITS", "7150114")), int(os.environ.get("RLNAMD_BENCH_BATCH", "1024")))
    else:
        one_at_a_time()
