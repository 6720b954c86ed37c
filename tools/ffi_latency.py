"""Latency of the zerokit FFI calls a nwaku-style caller makes one at a time (needs the GPU): single proof, verify,
small batches, one leaf update.  Run: python tools/ffi_latency.py"""
import time, sys
sys.path.insert(0,'/root/repo')
from zerokit_amd.public import RLN, RLNWitnessInput
from zerokit_amd import hashers
rln = RLN(20)
secret = 1234567
rln.set_leaf(3, hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 100))
elems, bits = rln.get_merkle_proof(3)
ws = [RLNWitnessInput(secret, 100, i % 100, elems, bits, 1000 + i, 777) for i in range(64)]
p = rln.generate_rln_proof(ws[0])
for n in (1, 1, 1):
    t = time.perf_counter(); p = rln.generate_rln_proof(ws[1]); dt = time.perf_counter() - t
    print("single proof latency ms", round(dt * 1e3, 2))
t = time.perf_counter(); ok = rln.verify_rln_proof(p, 1001); print("verify ms", round((time.perf_counter() - t) * 1e3, 2), ok)
for n in (8, 64):
    t = time.perf_counter(); ps = rln.generate_rln_proofs_batch(ws[:n]); dt = time.perf_counter() - t
    print("batch", n, "ms", round(dt * 1e3, 2), "per proof", round(dt * 1e3 / n, 3))
t = time.perf_counter(); rln.set_leaf(5, 99); r = rln.get_root(); print("set_leaf+root ms", round((time.perf_counter() - t) * 1e3, 3))
# host-side batch verification (rlnamd_verify_many_with_zkey: no GPU involved), golden proofs repeated
import json, os
from zerokit_amd.batch import verify_many_with_zkey
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = open(os.path.join(root, "zerokit_amd", "resources", "tree_depth_20", "rln_final.arkzkey"), "rb").read()
cases = json.load(open(os.path.join(root, "tests", "golden", "rln_h20_vectors.json")))["cases"]
reps = 512 // len(cases)
proofs = [bytes.fromhex(c["proof_compressed"]) for c in cases] * reps
pubs = [[int(v) for v in c["public_inputs"]] for c in cases] * reps
for threads in (1, 0):
    t = time.perf_counter(); ok = verify_many_with_zkey(z, proofs, pubs, threads=threads); dt = time.perf_counter() - t
    print("verify_many", len(proofs), "proofs, threads", threads or os.cpu_count(), "->", round(len(proofs) / dt), "verifications/s", all(ok))
