"""Pins the Python oracle (oracle/pyref) against every known-answer vector the reference's own tests hold
for the hot path.  The vectors below are data copied from the reference's tests:
  utils/tests/poseidon_hash_test.rs:21-130, rln/tests/protocol.rs:14-87, rln/tests/public.rs:84-135.
"""
import pytest

from oracle.pyref import arkzkey, groth16, rln, wtns_graph
from oracle.pyref.bn254 import G1, G2, G1_GEN, G2_GEN, R, pairing, f12_pow, f12_one
from oracle.pyref.keccak import hash_to_field_le, keccak256
from oracle.pyref.poseidon import constants, poseidon

POSEIDON_1 = {  # utils/tests/poseidon_hash_test.rs:21-66
    0: 19014214495641488759237505126948346942972912379615652741039992445865937985820,
    1: 18586133768512220936620570745912940619677854269274689475585506675881198879027,
    255: 20026131459732984724454933360292530547665726761019872861025481903072111625788,
    0xFFFF: 12358868638722666642632413418981275677998688723398440898957566982787708451243,
    0xFFFFFFFFFFFFFFFF: 17449307747295017006142981453320720946812828330895590310359634430146721583189,
}


def test_poseidon_single_kats():
    for k, v in POSEIDON_1.items():
        assert poseidon([k]) == v


def test_poseidon_pair_kats_8_leaf_tree():  # poseidon_hash_test.rs:69-130
    l01 = 12583541437132735734108669866114103169564651237895298778035846191048104863326
    l23 = 17197790661637433027297685226742709599380837544520340689137581733613433332983
    l45 = 756592041685769348226045093946546956867261766023639881791475046640232555043
    l67 = 5558359459771725727593826278265342308584225092343962757289948761260561575479
    l03 = 3720616653028013822312861221679392249031832781774563366107458835261883914924
    l47 = 7960741062684589801276390367952372418815534638314682948141519164356522829957
    root = 11780650233517635876913804110234352847867393797952240856403268682492028497284
    assert poseidon([0, 1]) == l01 and poseidon([2, 3]) == l23
    assert poseidon([4, 5]) == l45 and poseidon([6, 7]) == l67
    assert poseidon([l01, l23]) == l03 and poseidon([l45, l67]) == l47
    assert poseidon([l03, l47]) == root
    t = rln.FullMerkleTree(3)
    t.set_range(0, list(range(8)))
    assert t.root() == root


def test_poseidon_first_ark_constant():  # utils/tests/poseidon_constants.rs:44 (t=2, first ARK entry)
    ark, mds, rf, rp = constants(2)
    assert len(ark) == 2 * 64 and (rf, rp) == (8, 56)
    assert ark[0] == 4417881134626180770308697923359573201005643519861877412381846989312604493735


def test_poseidon_all_round_constants_and_mds():
    """Every round constant and MDS entry for t = 2..9 that the reference's test hard-codes
    (utils/tests/poseidon_constants.rs:42-3490, test_bn254_constants_generation :3523): the digests in
    tests/golden/poseidon_constants_digest.json were taken from that file by gen_poseidon_constants_digest.py;
    the oracle derives the same values with its Grain LFSR."""
    import hashlib
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                    "poseidon_constants_digest.json")))
    assert [s["t"] for s in d["sets"]] == list(range(2, 10))
    for s in d["sets"]:
        ark, mds, rf, rp = constants(s["t"])
        assert (rf, rp, len(ark)) == (s["rf"], s["rp"], s["n_round_constants"])
        assert str(ark[0]) == s["first_round_constant"] and str(mds[0][0]) == s["mds_00"]
        flat = ark + [x for row in mds for x in row]
        assert hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in flat)).hexdigest() == s["sha256"], s["t"]


def test_keccak256_empty():
    assert keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"


PATH_KAT = [  # rln/tests/protocol.rs:53-74
    "0x0000000000000000000000000000000000000000000000000000000000000000",
    "0x2098f5fb9e239eab3ceac3f27b81e481dc3124d55ffed523a839ee8446b64864",
    "0x1069673dcdb12263df301a6ff584a7ec261a44cb9dc68df067a4774460b1f1e1",
    "0x18f43331537ee2af2e3d758d50f72106467c6eea50371dd528d57eb2b856d238",
    "0x07f9d837cb17b0d36320ffe93ba52345f1b728571a568265caac97559dbc952a",
    "0x2b94cf5e8746b3f5c9631f4c5df32907a699c58c94b2ad4d7b5cec1639183f55",
    "0x2dee93c5a666459646ea7d22cca9e1bcfed71e6951b953611d11dda32ea09d78",
    "0x078295e5a22b84e982cf601eb639597b8b0515a88cb5ac7fa8a4aabe3c87349d",
    "0x2fa5e5f18f6027a6501bec864564472a616b2e274a41211a444cbe3a99f3cc61",
    "0x0e884376d0d8fd21ecb780389e941f66e45e7acce3e228ab3e2156a614fcd747",
    "0x1b7201da72494f1e28717ad1a52eb469f95892f957713533de6175e5da190af2",
    "0x1f8d8822725e36385200c0b201249819a6e6e1e4650808b5bebc6bface7d7636",
    "0x2c5d82f66c914bafb9701589ba8cfcfb6162b0a12acf88a8d0879a0471b5f85a",
    "0x14c54148a0940bb820957f5adf3fa1134ef5c4aaa113f4646458f270e0bfbfd0",
    "0x190d33b12f986f961e10c0ee44d8b9af11be25588cad89d416118e4bf4ebe80c",
    "0x22f98aa9ce704152ac17354914ad73ed1167ae6596af510aa5b3649325e06c92",
    "0x2a7c7c9b6ce5880b9f6f228d72bf6a575a526f29c66ecceef8b753d38bba7323",
    "0x2e8186e558698ec1c67af9c14d463ffc470043c9c2988b954d75dd643f36b992",
    "0x0f57c5571e9a4eab49e2c8cf050dae948aef6ead647392273546249d1c1ff10f",
    "0x1830ee67b5fb554ad5f63d4388800e1cfe78e310697d46e43c9ce36134f72cca",
]
ROOT_LIMBS = [4939322235247991215, 5110804094006647505, 4427606543677101242, 910933464535675827]


def test_depth20_tree_kat():  # rln/tests/protocol.rs:14-87 == rln/tests/ffi.rs:325-423
    secret = hash_to_field_le(b"test-merkle-proof")
    rate_commitment = poseidon([poseidon([secret]), 100])
    tree = rln.FullMerkleTree(20)
    tree.set(3, rate_commitment)
    assert tree.root() == sum(l << (64 * i) for i, l in enumerate(ROOT_LIMBS))
    elems, bits = tree.proof(3)
    assert elems == [int(x, 16) for x in PATH_KAT]
    assert bits == [1, 1] + [0] * 18
    assert rln.compute_tree_root(secret, 100, elems, bits) == tree.root()


def test_pairing_bilinear():
    e1 = pairing(G1_GEN, G2_GEN)
    a, b = 0x1234567, 0x7654321
    assert pairing(G1.mul(G1_GEN, a), G2.mul(G2_GEN, b)) == f12_pow(e1, a * b)
    assert e1 != f12_one() and f12_pow(e1, R) == f12_one()


def test_arkzkey_consumed_and_points_on_curve(circuit20):
    zk, _ = circuit20
    assert zk.consumed == 3405784
    assert (zk.num_instance_variables, zk.num_witness_variables, zk.num_constraints) == (6, 5839, 5820)
    assert (zk.a_nnz, zk.b_nnz, zk.c_nnz) == (9658, 13282, 0)
    assert [len(q) for q in (zk.a_query, zk.b_g1_query, zk.b_g2_query, zk.h_query, zk.l_query)] == \
        [5844, 5844, 5844, 8192, 5838]
    assert sum(p is None for p in zk.a_query) == 47 and sum(p is None for p in zk.b_g2_query) == 1999
    for q in (zk.a_query[:200], zk.h_query[:200], zk.l_query[:200], zk.gamma_abc_g1):
        assert all(G1.on_curve(p) for p in q)
    assert all(G2.on_curve(p) for p in zk.b_g2_query[:200] + [zk.beta_g2, zk.gamma_g2, zk.delta_g2])


SNARKJS = dict(  # rln/tests/public.rs:84-135
    a=(606446415626469993821291758185575230335423926365686267140465300918089871829,
       14881534001609371078663128199084130129622943308489025453376548677995646280161),
    b=((18053812507994813734583839134426913715767914942522332114506614735770984570178,
        11219916332635123001710279198522635266707985651975761715977705052386984005181),
       (17371289494006920912949790045699521359436706797224428511776122168520286372970,
        14038575727257298083893642903204723310279435927688342924358714639926373603890)),
    c=(17701377127561410274754535747274973758826089226897242202671882899370780845888,
       12608543716397255084418384146504333522628400182843246910626782513289789807030),
    values=dict(
        root=8502402278351299594663821509741133196466235670407051417832304486953898514733,
        x=20645213238265527935869146898028115621427162613172918400241870500502509785943,
        external_nullifier=21074405743803627666274838159589343934394162804826017440941339048886754734203,
        y=16401008481486069296141645075505218976370369489687327284155463920202585288271,
        nullifier=9102791780887227194595604713537772536258726662792598131262022534710887343694),
)


def test_snarkjs_hardcoded_proof_verifies(circuit20):
    zk, _ = circuit20
    proof = (SNARKJS["a"], SNARKJS["b"], SNARKJS["c"])
    assert groth16.verify(zk, proof, rln.public_inputs(SNARKJS["values"]))
    # serialisation order (root, ext, x, y, nullifier) is NOT the verifier order: must fail
    v = SNARKJS["values"]
    wrong = [v["root"], v["external_nullifier"], v["x"], v["y"], v["nullifier"]]
    assert not groth16.verify(zk, proof, wrong)
    # compressed round trip
    assert arkzkey.proof_decompress(arkzkey.proof_compress(*proof)) == proof


def test_witness_graph_internal_oracle(circuit20):
    _, g = circuit20
    assert len(g.nodes) == 23414 and len(g.signals) == 5844 and g.tree_depth == 20 and g.max_out == 1
    w = rln.WitnessInput(12345, 100, 1, [7 * i + 1 for i in range(20)], [i & 1 for i in range(20)], 42, 100)
    full = wtns_graph.calc_witness(g, w.named_inputs())
    assert full[0] == 1
    assert full[1:6] == rln.public_inputs(rln.proof_values_from_witness(w))


def test_graph_op_semantics():  # iden3calc/graph.rs:485-707 (division, signed comparisons, shifts)
    ev = wtns_graph.eval_duo
    assert ev("Div", 2, 3) == 2 * pow(3, -1, R) % R and ev("Div", 5, 0) == 0
    assert ev("Idiv", 7, 2) == 3 and ev("Mod", 7, 2) == 1
    assert ev("Lt", R - 1, 1) == 1 and ev("Gt", R - 1, 1) == 0   # R-1 is "-1"
    assert ev("Leq", 3, 3) == 1 and ev("Geq", 2, 3) == 0
    assert ev("Shr", 0b1100, 2) == 0b11 and ev("Shr", 5, 254) == 0 and ev("Shl", 1, 254) == 0
    assert ev("Shr", 1 << 200, 130) == 1 << 70
    assert ev("Band", 0b1100, 0b1010) == 0b1000 and ev("Bxor", 0b1100, 0b1010) == 0b0110


@pytest.mark.slow
def test_prove_appendix_d_vector(circuit20):
    """SURVEY Appendix D cross-check vector (survey-derived, not a reference golden) + pairing check."""
    zk, g = circuit20
    w = rln.WitnessInput(12345, 100, 1, [0] * 20, [0] * 20, 42, 100)
    proof, full = rln.generate_zk_proof_with_rs(zk, g, w, 44, 77)
    h = groth16.witness_map(zk, full)
    assert h[0] == 6799102154578598571546694245595676695036245381228798351010643146227773903871
    assert h[8191] == 14763356662025135782995008099257030610107229462187136303552182093756151409330
    assert arkzkey.proof_compress(*proof).hex() == (
        "939c090ddaf2c439c6aa96f98ed1f6a5ee744fb198c5a0592afae8c81ae2c9067ab339d5a53820ed64cd383b70cb0748"
        "34966d01384d39783d9a793104998a150e4aa0c536dec8f60690b9e12c40598627c322f73f7324f33c569cd274f11aa1"
        "e9d8237c7b58b201c946e56ce4777ac0dc154929c5688c36f4ace4a2f2a126ab")
    assert groth16.verify(zk, proof, full[1:6])


def test_multi_message_id_circuit_internal_oracle():
    """depth-20 multi-message-id circuit (max_out 4): witness outputs equal the formulae of
    protocol/witness.rs:777-802 in the verifier order of proof.rs:870-885."""
    zk, g = rln.load_circuit(20, multi=True)
    assert (zk.num_instance_variables, zk.num_constraints, len(g.nodes), g.max_out) == (16, 7390, 29254, 4)
    sel = [0, 1, 1, 1]
    w = dict(identitySecret=[987654321], userMessageLimit=[10], messageId=[0, 4, 5, 9], selectorUsed=sel,
             pathElements=[3 * i + 2 for i in range(20)], identityPathIndex=[(i >> 1) & 1 for i in range(20)],
             x=[77], externalNullifier=[88])
    full = wtns_graph.calc_witness(g, w)
    assert full[1:16] == rln.proof_values_multi(987654321, 10, w["messageId"], sel, w["pathElements"],
                                                w["identityPathIndex"], 77, 88)


@pytest.mark.slow
def test_partial_then_finish_equals_full_proof(circuit20):
    """rln/tests/protocol.rs:222-248: partial + finish == full proof for r = 44, s = 77"""
    zk, g = circuit20
    mask = groth16.known_mask(g)
    assert sum(mask) == 5339 and mask[0] and not mask[1] and mask[2]      # root is known, y is not
    w = rln.WitnessInput(12345, 100, 1, [0] * 20, [0] * 20, 42, 100)
    full = wtns_graph.calc_witness(g, w.named_inputs())
    part = groth16.prove_partial(zk, full, mask)
    assert groth16.finish_partial(zk, part, full, mask, 44, 77) == groth16.prove(zk, full, 44, 77)


def test_seeded_keygen_kats():
    """rln/tests/protocol.rs:463-517 and rln/tests/ffi_utils.rs:8-66 (ChaCha20Rng + Fr::rand + Poseidon)"""
    from oracle.pyref import keygen
    s, c = keygen.seeded_keygen(b"A seed phrase example")
    assert s == 0x20df38f3f00496f19fe7c6535492543b21798ed7cb91aebe4af8012db884eda3
    assert c == 0x1223a78a5d66043a7f9863e14507dc80720a5602b2a894923e5b5147d5a9c325
    s, c = keygen.seeded_keygen(bytes(range(10)))
    assert s == 0x766ce6c7e7a01bdf5b3f257616f603918c30946fa23480f2859c597817e6716
    assert c == 0xbf16d2b5c0d6f9d9d561e05bfca16a81b4b873bb063508fae360d8c74cef51f
    t, n, s, c = keygen.extended_seeded_keygen(bytes(range(10)))
    assert t == 0x766ce6c7e7a01bdf5b3f257616f603918c30946fa23480f2859c597817e6716
    assert n == 0x1f18714c7bc83b5bca9e89d404cf6f2f585bc4c0f7ed8b53742b7e2b298f50b4
    assert s == 0x2aca62aaa7abaf3686fff2caf00f55ab9462dc12db5b5d4bcf3994e671f8e521
    assert c == 0x68b66aa0a8320d2e56842581553285393188714c48f9b17acd198b4f1734c5c


def test_sparse_tree_oracle_agrees_with_the_full_tree_oracle():
    """oracle/pyref SparseMerkleTree (the checker of the deep-tree GPU test) against FullMerkleTree, itself pinned to
    the reference's depth-20 tree KAT above: same root and paths after the same updates"""
    import random
    from oracle.pyref.rln import FullMerkleTree, SparseMerkleTree
    rnd = random.Random(5)
    d = 7
    full, sparse = FullMerkleTree(d), SparseMerkleTree(d)
    assert full.root() == sparse.root()
    for _ in range(40):
        i, v = rnd.randrange(1 << d), rnd.randrange(1, 1 << 200)
        full.set(i, v)
        sparse.set(i, v)
    assert full.root() == sparse.root()
    for i in (0, 1, 77, 127):
        assert tuple(full.proof(i)) == tuple(sparse.proof(i)) and full.get(i) == sparse.get(i)
