"""Pins the C oracle (oracle/c/rln_oracle.c -- the CPU baseline and the large-batch checker) against the
Python oracle and the committed golden vectors.  CPU only."""
import json
import os
import random

from oracle.c import binding as ob
from oracle.pyref import rln as o_rln
from oracle.pyref.bn254 import R
from oracle.pyref.poseidon import poseidon

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]


def _w(case):
    w = case["witness"]
    return dict(identity_secret=int(w["identity_secret"]), user_message_limit=int(w["user_message_limit"]),
                message_id=int(w["message_id"]), path_elements=[int(t) for t in w["path_elements"]],
                identity_path_index=[int(t) for t in w["identity_path_index"]], x=int(w["x"]),
                external_nullifier=int(w["external_nullifier"]))


def test_c_poseidon_vs_python_oracle():
    rnd = random.Random(5)
    for arity in (1, 2, 3):
        rows = [[rnd.randrange(R) for _ in range(arity)] for _ in range(40)] + [[0] * arity, [R - 1] * arity]
        assert ob.poseidon_batch(rows) == [poseidon(r) for r in rows]


def test_c_tree_root_vs_python_oracle():
    for depth, n in ((3, 8), (5, 7), (6, 0)):
        leaves = list(range(1, n + 1))
        t = o_rln.FullMerkleTree(depth)
        t.set_range(0, leaves)
        assert ob.tree_root(depth, leaves) == t.root()


def test_c_slots_match_graph_metadata(circuit20):
    _, g = circuit20
    m = g.input_mapping
    assert (m["x"][0], m["externalNullifier"][0], m["identitySecret"][0], m["userMessageLimit"][0],
            m["messageId"][0], m["pathElements"][0], m["identityPathIndex"][0]) == (1, 2, 3, 4, 5, 6, 26)


def test_c_prover_matches_goldens():
    c = ob.Circuit(20)
    assert (c.n_inputs, c.n_signals, c.domain) == (46, 5844, 8192)
    import hashlib
    for case in _cases():
        o = c.prove(_w(case), int(case["r"]), int(case["s"]), want_witness=True, want_h=True)
        assert [str(v) for v in o["public_inputs"]] == case["public_inputs"]
        assert hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in o["witness"])).hexdigest() == \
            case["witness_sha256"]
        assert hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in o["h"])).hexdigest() == case["h_sha256"]
        assert o["proof"].hex() == case["proof_compressed"], case["name"]
        assert [str(o["coords"][0]), str(o["coords"][1])] == case["a"]


def test_c_prove_many_threads_equal_single():
    from oracle.pyref import workload
    c = ob.Circuit(20)
    ws, rs = workload.config2_witnesses(4, seed=321)
    _, proofs, pub = c.prove_many(ws, rs, threads=4)
    for i in range(4):
        o = c.prove(ws[i], rs[i][0], rs[i][1])
        assert o["proof"] == proofs[i] and o["public_inputs"] == pub[i]
