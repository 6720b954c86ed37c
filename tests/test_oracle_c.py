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


def test_side_config_oracles_config3_and_config5():
    """the oracle legs of configs 3 and 5 against the Python oracle: the config-5 closed form (sum k_i s_i) G and the
    workload's points in C limbs == Python ints (all distribution variants), Pippenger over the materialised points == the
    closed form, the C FullMerkleTree == oracle/pyref's (roots, paths), and the product's tree-update stream generator ==
    the one oracle_tree_bench applies"""
    import random
    from oracle.c import binding as ob
    from oracle.pyref import workload as owl
    from oracle.pyref.bn254 import G1, G1_GEN
    from oracle.pyref.rln import FullMerkleTree
    from zerokit_amd import workload
    seed = 0xC0FFEE
    for mode in (0, 1, 2, 3):
        assert ob.msm_expected(seed, 7, 500, mode, threads=3) == owl.msm_expected(seed, 7, 500, mode)
        p, s = ob.msm_workload_item(seed, 11, mode)
        k, s2 = owl.msm_item(seed, 11, mode)
        assert p == G1.mul(G1_GEN, k) and s == s2
        assert ob.msm_pippenger(seed, 7, 500, mode, threads=2)[0] == ob.msm_expected(seed, 7, 500, mode)
    assert ob.msm_expected(seed, 0, 0) is None
    rnd = random.Random(5)
    t, o = ob.Tree(8), FullMerkleTree(8)
    assert t.root() == o.root()
    for _ in range(12):
        i, v = rnd.randrange(256), rnd.randrange(1 << 253)
        t.set(i, v)
        o.set(i, v)
    t.set_range(40, [1, 2, 3], threads=2)
    for k, v in enumerate([1, 2, 3]):
        o.set(40 + k, v)
    assert t.root() == o.root()
    pe, pb = o.proof(41)
    assert t.proof(41) == (list(pe), list(pb))
    t.close()
    assert workload.tree_update_stream(1 << 20, 40, 0x5CA7, 0x5CA7000000000000) == ob.scattered_updates(1 << 20, 40)


def test_c_oracle_on_the_other_shipped_circuits_vs_goldens_and_pyref():
    """oracle/c is generic over (arkzkey, graph): on the depth-10 single circuit and the depth-20 multi-message-id circuit
    it reproduces the pyref-generated goldens (tests/golden/rln_other_circuits.json: proof bytes, public inputs in the
    verifier's order) -- which pins it as the judge of the GPU's throughput shape on those circuits
    (tests/test_gpu_parity.py::test_other_circuits_throughput_shape_vs_oracle).  On items of the seeded workload the
    Poseidon-formula public values (witness.rs:759-802) equal the circuit's own public signals w[1..], and pyref's
    proof_values_multi agrees."""
    import json
    import os
    from oracle.c import binding as ob
    from oracle.pyref import rln as pr
    from zerokit_amd import workload
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(root, "tests", "golden", "rln_other_circuits.json")))["cases"]
    assert {(c["depth"], c["multi"]) for c in cases} == {(10, False), (20, True)}
    for c in cases:
        o = ob.Circuit(c["depth"], c["multi"])
        named = {k: [int(v) for v in vs] for k, vs in c["inputs"].items()}
        out = o.prove_packed(o.pack_named(named), int(c["r"]), int(c["s"]), want_witness=True)
        assert out["proof"].hex() == c["proof_compressed"], c["name"]
        assert [str(v) for v in out["public_inputs"]] == c["public"], c["name"]
        assert out["witness"][1:1 + o.n_public] == out["public_inputs"], c["name"]
        ws, rs = workload.circuit_range(40, 3, c["depth"], c["multi"])
        inp = b"".join(o.pack_named(w) for w in ws)
        rsb = b"".join(r.to_bytes(32, "little") + s.to_bytes(32, "little") for r, s in rs)
        _, proofs, pub = o.prove_many_packed(inp, rsb, threads=3)
        for i, w in enumerate(ws):
            one = o.prove_packed(o.pack_named(w), rs[i][0], rs[i][1], want_witness=True)
            assert one["proof"] == proofs[i] and one["public_inputs"] == pub[i]
            assert one["witness"][1:1 + o.n_public] == pub[i]
            if c["multi"]:
                assert pub[i] == pr.proof_values_multi(w["identitySecret"][0], w["userMessageLimit"][0], w["messageId"],
                                                       w["selectorUsed"], w["pathElements"], w["identityPathIndex"],
                                                       w["x"][0], w["externalNullifier"][0])
                assert any(w["selectorUsed"]) and len(set(w["messageId"])) == 4 and max(w["messageId"]) < 100
            else:
                wi = pr.WitnessInput(w["identitySecret"][0], w["userMessageLimit"][0], w["messageId"][0], w["pathElements"],
                                     w["identityPathIndex"], w["x"][0], w["externalNullifier"][0])
                assert pub[i] == pr.public_inputs(pr.proof_values_from_witness(wi))


def test_c_partial_and_finish_vs_pyref_fixture_and_goldens():
    """oracle/c's generate_partial_zk_proof / finish_zk_proof_with_rs (proof.rs:783-849, partial_proof.rs:108-274,
    graph.rs:274-312) against the pyref-generated fixture tests/golden/rln_h20_partial.json: the known-signal mask, the
    four partial points byte for byte, and finish(partial) == the golden FULL proof of the same (witness, r, s) -- the
    equality rln/tests/protocol.rs:222-248 asserts -- including the r = 0 branch (g1_b = 0, partial_proof.rs:242-248)."""
    import hashlib
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_partial.json")))
    c = ob.Circuit(20)
    mask = c.known_mask()
    assert (len(mask), sum(mask)) == (fx["mask_len"], fx["mask_known"]) and mask[0] and not mask[1] and mask[2]
    assert hashlib.sha256(bytes(int(b) for b in mask)).hexdigest() == fx["mask_sha256"]
    cases = {k["name"]: k for k in _cases()}
    for f in fx["cases"]:
        case = cases[f["name"]]
        packed = c.pack(_w(case))
        # the unknown slots are not read: garbage there gives the same partial proof
        junk = bytearray(packed)
        for slot in (1, 2, 5):
            junk[32 * slot:32 * slot + 32] = (12345678901234567890 + slot).to_bytes(32, "little")
        part = c.prove_partial_packed(packed)
        assert part.hex() == f["partial320"], f["name"]
        assert c.prove_partial_packed(bytes(junk)) == part
        got = c.finish_packed(packed, int(case["r"]), int(case["s"]), part)
        assert got.hex() == case["proof_compressed"], f["name"]
    # many at once on threads == one by one
    ws = [_w(cases[n]) for n in ("config2_0", "config2_1", "config2_2")]
    rs = [(int(cases[n]["r"]), int(cases[n]["s"])) for n in ("config2_0", "config2_1", "config2_2")]
    packed = [c.pack(w) for w in ws]
    parts = [c.prove_partial_packed(p) for p in packed]
    _, proofs = c.finish_many_packed(b"".join(packed), b"".join(ob._b(r) + ob._b(s) for r, s in rs), b"".join(parts), threads=3)
    assert [p.hex() for p in proofs] == [cases[n]["proof_compressed"] for n in ("config2_0", "config2_1", "config2_2")]


def test_c_field_product_forms_agree():
    """the mulx / adcx / adox product this build took (where the CPU has ADX + BMI2) against the portable CIOS product
    it replaced, both fields, random and edge operands"""
    assert ob.lib().oracle_selftest_mul(0xC0FFEE, 300000) == 1
    assert ob.lib().oracle_mul_kind() in (b"mulx/adcx/adox", b"portable (unsigned __int128)")
