"""GPU parity tests: the HIP path, called through the C ABI, against the oracle and the committed golden
vectors.  Bit-exact everywhere (integer arithmetic): no tolerances."""
import hashlib
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _cases():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))


def _w(case):
    w = case["witness"]
    return dict(identity_secret=int(w["identity_secret"]), user_message_limit=int(w["user_message_limit"]),
                message_id=int(w["message_id"]), path_elements=[int(t) for t in w["path_elements"]],
                identity_path_index=[int(t) for t in w["identity_path_index"]], x=int(w["x"]),
                external_nullifier=int(w["external_nullifier"]))


def _digest(vals):
    return hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in vals)).hexdigest()


@pytest.fixture(scope="module")
def prover():
    from zerokit_amd.batch import BatchProver
    os.environ["RLNAMD_PARTIAL_CACHE"] = "160"      # (read once, when the prover is built: a 128-proof batch of cached partials fits)
    try:
        p = BatchProver(max_batch=128)
    finally:
        del os.environ["RLNAMD_PARTIAL_CACHE"]
    yield p
    p.close()


# ------------------------------------------------------------------------------------------ Poseidon
def test_poseidon_kats_and_random_vs_oracle():
    from oracle.pyref.poseidon import poseidon
    from zerokit_amd import hashers
    # utils/tests/poseidon_hash_test.rs:21-66
    assert hashers.poseidon_hash([0]) == 19014214495641488759237505126948346942972912379615652741039992445865937985820
    assert hashers.poseidon_hash([1]) == 18586133768512220936620570745912940619677854269274689475585506675881198879027
    assert hashers.poseidon_hash([0xFFFFFFFFFFFFFFFF]) == \
        17449307747295017006142981453320720946812828330895590310359634430146721583189
    # :69-130
    assert hashers.poseidon_hash_pair(0, 1) == \
        12583541437132735734108669866114103169564651237895298778035846191048104863326
    rnd = random.Random(7)
    for arity in range(1, 9):     # every width of rln/src/hashers.rs:14-23 (t = 2..9)
        rows = [[rnd.randrange(R) for _ in range(arity)] for _ in range(300 if arity <= 3 else 70)]
        rows[0] = [0] * arity
        rows[1] = [R - 1] * arity
        got = hashers.poseidon_hash_batch(rows)
        assert got == [poseidon(r) for r in rows]
    with pytest.raises(Exception, match="No parameters found for input length 9"):
        hashers.poseidon_hash(list(range(9)))     # utils/src/poseidon/error.rs:5
    with pytest.raises(Exception, match="Empty input provided"):
        hashers.poseidon_hash([])
    with pytest.raises(Exception, match="Non-canonical"):
        hashers.poseidon_hash([R])


# ------------------------------------------------------------------------------------------ tree
def test_depth20_tree_kat():
    """rln/tests/protocol.rs:14-87 == rln/tests/ffi.rs:325-423"""
    from tests.test_oracle_kats import PATH_KAT, ROOT_LIMBS
    from zerokit_amd import hashers
    from zerokit_amd.batch import PoseidonTree
    t = PoseidonTree(20)
    assert t.root() == 0x2134e76ac5d21aab186c2be1dd8f84ee880a1e46eaf712f9d371b6df22191f3e  # empty tree
    secret = hashers.hash_to_field_le(b"test-merkle-proof")
    rate_commitment = hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 100)
    t.set(3, rate_commitment)
    assert t.root() == sum(l << (64 * i) for i, l in enumerate(ROOT_LIMBS))
    elems, bits = t.proof(3)
    assert elems == [int(x, 16) for x in PATH_KAT]
    assert bits == [1, 1] + [0] * 18
    assert t.get(3) == rate_commitment and t.get(2) == 0


def test_tree_ragged_ranges_vs_oracle():
    from oracle.pyref.rln import FullMerkleTree
    from zerokit_amd.batch import PoseidonTree
    rnd = random.Random(11)
    for depth in (1, 3, 7):
        ref, dev = FullMerkleTree(depth), PoseidonTree(depth)
        assert dev.root() == ref.root()
        cap = 1 << depth
        for _ in range(6):
            start = rnd.randrange(cap)
            n = rnd.randrange(1, cap - start + 1)
            leaves = [rnd.randrange(R) for _ in range(n)]
            ref.set_range(start, leaves)
            dev.set_range(start, leaves)
            assert dev.root() == ref.root()
        for i in (0, cap - 1, rnd.randrange(cap)):
            assert dev.proof(i) == ref.proof(i)
            assert dev.get(i) == ref.get(i)
        allp = dev.proofs(0, cap)
        assert allp == [ref.proof(i) for i in range(cap)]
        with pytest.raises(Exception):
            dev.set_range(cap - 1, [1, 2])      # TooManySet
        with pytest.raises(Exception):
            dev.proof(cap)                       # InvalidLeaf
    d0 = PoseidonTree(0)
    d0.set(0, 5)
    assert d0.root() == 5 and d0.proof(0) == ([], [])


def test_tree_config3_small_and_checksum():
    """BASELINE config 3 at 2^12 leaves: leaves i+1, all paths recompute the root on the device; root and
    sampled paths equal the oracle."""
    from oracle.pyref.rln import FullMerkleTree
    from zerokit_amd.batch import PoseidonTree
    depth = 12
    dev = PoseidonTree(depth)
    res = dev.bench(1 << depth, first_value=1, verify=True)
    assert res["bad"] == 0
    ref = FullMerkleTree(depth)
    ref.set_range(0, list(range(1, (1 << depth) + 1)))
    assert dev.root() == ref.root()
    for i in (0, 1, 2047, 4095):
        assert dev.proof(i) == ref.proof(i)


def test_tree_config3_full_size_properties():
    """BASELINE config 3 at full size (2^20 leaves i+1, 2^20 paths): every emitted path recomputes the root on
    the device (compute_root_from, full_merkle_tree.rs:441-446); root equals the C oracle's; sampled paths
    equal a host recomputation from the oracle's Poseidon; an update of one leaf moves exactly its path."""
    from oracle.c import binding as ob
    from zerokit_amd.batch import PoseidonTree
    depth = 20
    t = PoseidonTree(depth)
    res = t.bench(1 << depth, first_value=1, verify=True)
    assert res["bad"] == 0
    root = t.root()
    assert root == ob.tree_root(depth, list(range(1, (1 << depth) + 1)))
    for leaf in (0, 1, 524287, 524288, (1 << depth) - 1):
        elems, bits = t.proof(leaf)
        assert bits == [(leaf >> k) & 1 for k in range(depth)]
        h = leaf + 1
        for e, b in zip(elems, bits):
            h = ob.poseidon_batch([[e, h] if b else [h, e]])[0]
        assert h == root
    before = t.proof(12345)
    t.set(777777, 42)
    assert t.root() != root and t.proof(12345)[0][:5] == before[0][:5] and t.get(777777) == 42


# ------------------------------------------------------------------------------------------ prover
def test_witness_and_h_vs_golden(prover):
    cases = _cases()["cases"]
    ws = [_w(c) for c in cases]
    rs = [(int(c["r"]), int(c["s"])) for c in cases]
    out = prover.prove(ws, rs)
    for i, c in enumerate(cases):
        assert out[i]["error"] == 0
        full = prover.fetch_witness(i)
        assert full[0] == 1 and [str(v) for v in full[1:6]] == c["public_inputs"]
        assert _digest(full) == c["witness_sha256"], c["name"]
        h = prover.fetch_h(i)
        assert [str(h[0]), str(h[1])] == c["h_first"] and str(h[-1]) == c["h_last"]
        assert _digest(h) == c["h_sha256"], c["name"]


def test_proofs_bit_exact_vs_golden_and_verify(prover):
    cases = _cases()["cases"]
    out = prover.prove([_w(c) for c in cases], [(int(c["r"]), int(c["s"])) for c in cases])
    for o, c in zip(out, cases):
        assert [str(v) for v in o["public_inputs"]] == c["public_inputs"], c["name"]
        assert [str(o["a"][0]), str(o["a"][1])] == c["a"], c["name"]
        assert [[str(o["b"][0][0]), str(o["b"][0][1])], [str(o["b"][1][0]), str(o["b"][1][1])]] == c["b"], c["name"]
        assert [str(o["c"][0]), str(o["c"][1])] == c["c"], c["name"]
        assert o["proof"].hex() == c["proof_compressed"], c["name"]
        assert prover.verify(o["proof"], o["public_inputs"])
        bad = list(o["public_inputs"])
        bad[2] = (bad[2] + 1) % R
        assert not prover.verify(o["proof"], bad)


def test_batch_vs_python_oracle_random(prover):
    """config-2 generator, 70 proofs (crosses a 64-lane boundary): every proof verifies; 2 sampled proofs
    are recomputed by the Python oracle and must be bit-identical."""
    from oracle.pyref import arkzkey, groth16, rln, workload, wtns_graph
    ws, rs = workload.config2_witnesses(70, seed=0xABCDEF)
    out = prover.prove(ws, rs)
    assert all(o["error"] == 0 for o in out)
    for o in out:
        assert prover.verify(o["proof"], o["public_inputs"])
    zk, g = rln.load_circuit(20)
    for i in (5, 69):
        w = ws[i]
        wi = rln.WitnessInput(w["identity_secret"], w["user_message_limit"], w["message_id"], w["path_elements"],
                              w["identity_path_index"], w["x"], w["external_nullifier"])
        proof, full = rln.generate_zk_proof_with_rs(zk, g, wi, rs[i][0], rs[i][1])
        assert arkzkey.proof_compress(*proof) == out[i]["proof"]
        assert rln.public_inputs(rln.proof_values_from_witness(wi)) == out[i]["public_inputs"]


def test_full_size_batch_1024_bit_exact_vs_c_oracle():
    """BASELINE config 2 at full size: all 1024 proofs of the bench workload are bit-identical to the C oracle
    (proof bytes and public values), pipelined twice to cover both workspace slots, and a sample verifies."""
    from conftest import oracle_config2
    from oracle.pyref import workload
    from zerokit_amd.batch import BatchProver
    n = 1024
    ws, rs, ref_proofs, ref_pub = oracle_config2(0, n)
    w3, r3 = workload.config2_witnesses(3)       # the Python oracle's own generator of the same stream
    assert (ws[:3], rs[:3]) == (w3, r3)
    p = BatchProver(max_batch=n)
    p.upload(p.pack_inputs(ws), rs)
    p.run_async(n)
    p.run_async(n)          # second slot; must reproduce the same bytes
    p.run_async(n)
    out = p.download(n)
    assert all(o["error"] == 0 for o in out)
    assert [o["proof"] for o in out] == ref_proofs
    assert [o["public_inputs"] for o in out] == ref_pub
    assert p.download_public(n) == ref_pub
    for i in (0, 63, 64, 511, 1023):
        assert p.verify(out[i]["proof"], out[i]["public_inputs"])
    p.close()


def test_small_batch_sizes_across_the_shape_boundaries_vs_c_oracle():
    """Every size at which the small-batch path changes shape (tiny plans with a lane pair per G2 point up to 5, lanes =
    chunks up to 48 for a lone batch, lanes = proofs over the short chunks up to 128, the throughput shape above; compact
    digit rows below, capacity-strided above), on the DEFAULT tables and through submit / collect: bit-identical to oracle/c"""
    from oracle.c import binding as ob
    from oracle.pyref import workload
    from zerokit_amd.batch import BatchProver
    nmax = 130
    ws, rs = workload.config2_witnesses(nmax, seed=4242)
    _, ref_proofs, ref_pub = ob.Circuit(20).prove_many(ws, rs)
    p = BatchProver(max_batch=192)
    try:
        for n in (1, 2, 3, 4, 5, 6, 15, 16, 17, 47, 48, 49, 64, 65, 96, 97, 128, 129, 130, 1):
            inp, rsb = p.pack_inputs(ws[:n]), p.pack_rs(rs[:n])
            t, _ = p.submit(inp, rsb)
            out = p.collect(t, n)
            assert all(o["error"] == 0 for o in out), n
            assert [o["proof"] for o in out] == ref_proofs[:n], n
            assert [o["public_inputs"] for o in out] == ref_pub[:n], n
        # not alone on the device: three sizes enqueued back to back
        ts = [p.submit(p.pack_inputs(ws[:n]), p.pack_rs(rs[:n]))[0] for n in (3, 40, 100)]
        for t, n in zip(ts, (3, 40, 100)):
            assert [o["proof"] for o in p.collect(t, n)] == ref_proofs[:n], n
    finally:
        p.close()


def test_edge_case_witnesses_vs_c_oracle(prover):
    """boundary values of every input: zero / r-1 field elements, message_id = limit - 1, all-ones path index,
    r = 0 (g1_b = 0 branch), s = 0, r = s = r_mod - 1"""
    from oracle.c import binding as ob
    base = dict(identity_secret=R - 1, user_message_limit=R - 1, message_id=R - 2, path_elements=[R - 1] * 20,
                identity_path_index=[1] * 20, x=R - 1, external_nullifier=R - 1)
    zero = dict(identity_secret=0, user_message_limit=1, message_id=0, path_elements=[0] * 20,
                identity_path_index=[0] * 20, x=0, external_nullifier=0)
    mixed = dict(identity_secret=1, user_message_limit=2, message_id=1, path_elements=[i for i in range(20)],
                 identity_path_index=[i & 1 for i in range(20)], x=1 << 253, external_nullifier=(1 << 128) - 1)
    ws = [base, zero, mixed, base, zero]
    rs = [(1, 1), (0, 5), (7, 0), (R - 1, R - 1), (0, 0)]
    out = prover.prove(ws, rs)
    c = ob.Circuit(20)
    for i, (o, w, (r, s)) in enumerate(zip(out, ws, rs)):
        ref = c.prove(w, r, s)
        assert o["error"] == 0
        assert o["proof"] == ref["proof"] and o["public_inputs"] == ref["public_inputs"]
        # `base` violates the circuit's message-id range check (limit and id near r): the prover still returns
        # the same (unsatisfying) proof as the CPU path, and it must NOT verify
        assert prover.verify(o["proof"], o["public_inputs"]) == (w is not base), i


def test_batch_is_order_and_size_independent(prover):
    """the same witness gives the same proof alone, in a batch, and at a different lane"""
    from oracle.pyref import workload
    ws, rs = workload.config2_witnesses(9, seed=99)
    a = prover.prove(ws, rs)
    b = prover.prove(ws[::-1], rs[::-1])[::-1]
    c = prover.prove([ws[4]], [rs[4]])
    assert [o["proof"] for o in a] == [o["proof"] for o in b]
    assert c[0]["proof"] == a[4]["proof"]


# ------------------------------------------------------------------------------------------ zerokit FFI
def test_ffi_round_trip_like_reference_tests():
    """rln/tests/protocol.rs:182-198 (prove -> verify) and :222-248 (fixed r=44, s=77 is deterministic),
    rln/tests/public.rs:349-427 (tree ops equivalence), through the zerokit C ABI."""
    from zerokit_amd import hashers
    from zerokit_amd.public import RLN, RLNError, RLNWitnessInput
    c = next(x for x in _cases()["cases"] if x["name"] == "config1_bench_witness")
    rln = RLN(20)
    secret = hashers.hash_to_field_le(b"test-merkle-proof")
    rate_commitment = hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 100)
    rln.set_leaf(3, rate_commitment)
    assert str(rln.get_root()) == _cases()["tree_root_config1"]
    elems, bits = rln.get_merkle_proof(3)
    w = _w(c)
    assert elems == w["path_elements"] and bits == w["identity_path_index"]
    wi = RLNWitnessInput(secret, 100, 1, elems, bits, w["x"], w["external_nullifier"])
    p1 = rln.generate_rln_proof_with_rs(wi, 44, 77)
    assert p1.to_bytes_le().hex() == c["rln_proof_le"]
    assert rln.verify_rln_proof(p1, w["x"])
    p2 = rln.generate_rln_proof(wi)                       # random blinding: different bytes, still valid
    assert p2.to_bytes_le() != p1.to_bytes_le()
    assert rln.verify_rln_proof(p2, w["x"])
    assert rln.verify_with_roots(p2, w["x"], []) and rln.verify_with_roots(p2, w["x"], [5, rln.get_root()])
    with pytest.raises(RLNError, match="Signal value does not match"):
        rln.verify_rln_proof(p2, w["x"] + 1)
    with pytest.raises(RLNError, match="Expected one of the provided roots"):
        rln.verify_with_roots(p2, w["x"], [1, 2])
    rln.set_leaf(4, 7)                                    # root moves: proof no longer matches the tree
    with pytest.raises(RLNError, match="Expected one of the provided roots"):
        rln.verify_rln_proof(p2, w["x"])
    # batch extension == single calls
    many = rln.generate_rln_proofs_batch([wi, wi], rs=[(44, 77), (1, 2)])
    assert many[0].to_bytes_le() == p1.to_bytes_le()
    # wrong path length is rejected before any kernel runs (proof.rs:644-662)
    with pytest.raises(RLNError, match="path_elements"):
        rln.generate_rln_proof(RLNWitnessInput(secret, 100, 1, elems[:19], bits[:19], 1, 2))
    # tree ops: single / next / batch insertion agree (public.rs tests :349-427)
    leaves = [hashers.poseidon_hash([i + 1]) for i in range(9)]
    a, b, d = RLN(20), RLN(20), RLN(20)
    for i, l in enumerate(leaves):
        a.set_leaf(i, l)
        b.set_next_leaf(l)
    d.init_tree_with_leaves(leaves)
    assert a.get_root() == b.get_root() == d.get_root()
    assert a.leaves_set() == b.leaves_set() == d.leaves_set() == 9
    d.set_leaves_from(9, leaves[:3])
    a.atomic_operation(9, leaves[:3], [])
    assert a.get_root() == d.get_root() and a.leaves_set() == 12
    a.delete_leaf(10)
    assert a.get_leaf(10) == 0 and a.leaves_set() == 12
    a.delete_leaf(500)                                    # >= next_index: no-op
    a.set_tree(20)
    assert a.get_root() == RLN(20).get_root() and a.leaves_set() == 0


_SCATTERED = {}


def _scattered_plan_and_oracle():
    """the update stream of test_tree_scattered_updates_one_pass_vs_oracle and what oracle/c's FullMerkleTree answers at
    every step -- the same for each of the four host-chain thresholds, so the oracle walks it ONCE per session (30 000
    set() calls of 20 hashes on one core)"""
    if _SCATTERED:
        return _SCATTERED
    from oracle.c import binding as ob
    rnd = random.Random(77)
    small = []
    for depth in (0, 1, 3):
        o = ob.Tree(depth)
        ups = [(rnd.randrange(1 << depth), rnd.randrange(1, R)) for _ in range(5)]
        for i, v in ups:
            o.set(i, v)
        small.append((depth, ups, o.root()))
        o.close()
    depth = 20
    o = ob.Tree(depth)
    o.set_range(0, list(range(1, (1 << 14) + 1)), threads=4)
    root0 = o.root()
    steps = []
    for k in (1, 1, 2, 5, 6, 7, 8, 9, 21, 84, 85, 86, 1000, 30000):
        ups = [(rnd.randrange(1 << depth), rnd.randrange(1, R)) for _ in range(k)]
        if k == 5:
            ups += [(ups[0][0] ^ 1, 17), (ups[1][0] ^ 2, 18)]      # a sibling and a cousin of dirty leaves
        if k >= 21:
            ups += [(ups[0][0], 5), (ups[1][0], 0), (ups[0][0], 6)]      # rewrites: the last one wins
        last = dict(ups)
        for i, v in last.items():
            o.set(i, v)
        probe = list(last)[:3] + [0, (1 << depth) - 1]
        steps.append((k, ups, last, o.root(), [(i, o.proof(i)) for i in probe]))
    # neighbours: both children of one parent dirty, whole aligned blocks dirty
    ups = [(i, 1000 + i) for i in range(4096, 4096 + 257)] + [(1 << 19, 1), ((1 << 19) + 1, 2)]
    for i, v in ups:
        o.set(i, v)
    _SCATTERED.update(small=small, root0=root0, steps=steps, last_ups=ups, last_root=o.root())
    o.close()
    return _SCATTERED


@pytest.mark.parametrize("host_max", ["0", None, "8", "4096"])
def test_tree_scattered_updates_one_pass_vs_oracle(host_max, monkeypatch):
    """(host_max: the largest pass that runs its dependent chain on a host core -- MerkleTreeDev::set_few: one gather of
    the clean siblings, the path hashes with the library's host Poseidon, one scatter -- forced to never / the default /
    8 / 4096, so that every k below goes through both forms.)
    rlnamd_tree_set_leaves: k single-leaf writes + ONE bottom-up pass over the union of their paths equals k set()
    calls of FullMerkleTree (full_merkle_tree.rs:141-147,336-399) -- for k = 1 (the whole path in one launch), 21 / 84 /
    85 / 86 (around the capacity of the single-workgroup tail), 1 000, 30 000 (more dirty parents than the three-lane
    kernel takes) and on small depths (0, 1, 3); duplicates keep the last write; root, leaves and paths vs oracle/c"""
    from zerokit_amd.batch import PoseidonTree
    if host_max is None:
        monkeypatch.delenv("RLNAMD_TREE_HOST_MAX", raising=False)
    else:
        monkeypatch.setenv("RLNAMD_TREE_HOST_MAX", host_max)
    plan = _scattered_plan_and_oracle()
    for depth, ups, root in plan["small"]:
        t = PoseidonTree(depth)
        t.set_leaves(ups)
        assert t.root() == root
        t.close()
    depth = 20
    t = PoseidonTree(depth)
    t.fill_sequential(0, 1 << 14, 1)
    assert t.root() == plan["root0"]
    for k, ups, last, root, proofs in plan["steps"]:
        t.set_leaves(ups)
        assert t.root() == root, k
        for i, proof in proofs:
            assert t.get(i) == last.get(i, t.get(i))
            assert t.proof(i) == proof
    t.set_leaves(plan["last_ups"])
    assert t.root() == plan["last_root"]
    with pytest.raises(Exception):
        t.set_leaves([(1 << depth, 1)])
    t.close()


# ------------------------------------------------------------------------------------------ variable-base MSM
def test_msm_g1_vs_oracle_small_and_edge_cases():
    """msm_bigint semantics on small inputs: random points/scalars vs the Python oracle, plus the edge cases
    (empty, zero scalars, infinity bases, repeated and opposite points, maximal scalar)."""
    from oracle.pyref.bn254 import G1, G1_GEN
    from zerokit_amd.batch import MsmG1
    rnd = random.Random(21)
    m = MsmG1(4096)
    pts = [G1.mul(G1_GEN, rnd.randrange(1, R)) for _ in range(40)]
    sc = [rnd.randrange(R) for _ in range(40)]
    assert m.msm(pts, sc) == G1.msm_naive(pts, sc)
    assert m.msm([], []) is None
    assert m.msm(pts[:3], [0, 0, 0]) is None
    assert m.msm([None, pts[0]], [5, 7]) == G1.mul(pts[0], 7)
    assert m.msm([pts[0]] * 5, [1, 2, 3, 4, 5]) == G1.mul(pts[0], 15)
    assert m.msm([pts[0], G1.neg(pts[0])], [9, 9]) is None
    assert m.msm([pts[1]], [R - 1]) == G1.neg(pts[1])
    assert m.msm([pts[2], pts[3]], [0x8000, 0xFFFF8000FFFF]) == G1.msm_naive([pts[2], pts[3]], [0x8000, 0xFFFF8000FFFF])
    # the same base with the same scalar lands in the same bucket of every window: the accumulator meets its own point
    # (doubling) and its negation (cancellation) inside the 9 x 29-bit group law
    assert m.msm([pts[4]] * 300, [7] * 300) == G1.mul(pts[4], 2100)
    assert m.msm([pts[5], G1.neg(pts[5]), pts[5], pts[6]] * 2, [R - 3] * 8) == \
        G1.msm_naive([pts[5], pts[6]], [2 * (R - 3) % R, 2 * (R - 3) % R])
    big = [pts[i % 40] for i in range(3000)]
    bsc = [rnd.randrange(R) for _ in range(3000)]
    assert m.msm(big, bsc) == G1.msm(big, bsc, c=8)


def test_msm_g1_generated_closed_form_and_split():
    """config-5 workload at 2^16 points: the generated points / scalars are the oracle's workload (spot check), the
    result equals the ORACLE's closed form (sum k_i s_i) G (oracle/c and, independently, oracle/pyref), and splitting the
    points into 4 slices (the per-GPU shards) and combining their window sums gives the same point (linearity), as the
    8-GPU all-gather does."""
    from oracle.c import binding as ob
    from oracle.pyref import workload as owl
    from zerokit_amd.batch import MsmG1
    n, seed = 1 << 16, 0xC0FFEE
    m = MsmG1(n)
    m.generate(seed, 0, n)
    for i in (0, 1, 63, 64, 4097, n - 1):
        assert m.fetch(i, 1)[0] == ob.msm_workload_item(seed, i)
    blob, ms = m.run_windows()
    want = ob.msm_expected(seed, 0, n)
    assert want == owl.msm_expected(seed, 0, n)          # the two oracles agree (Python ints vs C limbs)
    assert m.combine([blob]) == want
    blobs = []
    q = n // 4
    for r in range(4):
        m.generate(seed, r * q, q)
        blobs.append(m.run_windows()[0])
    assert m.combine(blobs) == want
    m.close()


def test_msm_g1_adversarial_distributions_vs_oracle():
    """the size-dependent paths of the sort and of the bucket walk under skewed inputs, each against the oracle's closed
    form AND (at 2^14) against the oracle's own Pippenger over the materialised points: every scalar equal (ONE bucket
    per window holds all points -- a partition far above the LDS staging capacity, a bucket cut into thousands of slices
    joined by the fix-up kernel), four distinct bases (a bucket keeps meeting a point it already holds), and both"""
    from oracle.c import binding as ob
    from zerokit_amd.batch import MsmG1
    seed = 0xC0FFEE
    n = 1 << 14
    m = MsmG1(1 << 18)
    for mode in (1, 2, 3):
        m.generate(seed, 5, n, mode)
        assert m.fetch(n - 1, 1)[0] == ob.msm_workload_item(seed, 5 + n - 1, mode)
        got = m.combine([m.run_windows()[0]])
        assert got == ob.msm_expected(seed, 5, n, mode), mode
        assert got == ob.msm_pippenger(seed, 5, n, mode)[0], mode
    n = 1 << 18
    for mode in (1, 2, 3):
        m.generate(seed, 0, n, mode)
        assert m.combine([m.run_windows()[0]]) == ob.msm_expected(seed, 0, n, mode), mode
    m.close()


def test_msm_fold_on_the_host_and_on_the_device_agree_with_the_oracle(monkeypatch):
    """the last step of config 5 -- adding the window sums of the contributors and 240 dependent doublings of one point --
    runs on a host core by default (0.1 ms; a lone GPU lane needs 1.75 ms for the same chain) and on the device with
    RLNAMD_MSM_FOLD=device: both equal the oracle's closed form, for one contributor and for a 3-way split, uniform and
    skewed scalars; a window sum at infinity (all scalars below 2^16 leave the upper windows empty) included"""
    from oracle.c import binding as ob
    from zerokit_amd.batch import MsmG1
    seed, n = 0xFEED, 3 * (1 << 14)
    m = MsmG1(n)
    try:
        want = {mode: ob.msm_expected(seed, 0, n, mode) for mode in (0, 1)}
        for fold in ("host", "device"):
            monkeypatch.setenv("RLNAMD_MSM_FOLD", fold)
            for mode in (0, 1):
                m.generate(seed, 0, n, mode)
                assert m.combine([m.run_windows()[0]]) == want[mode], (fold, mode)
                blobs = []
                for r in range(3):
                    m.generate(seed, r * (n // 3), n // 3, mode)
                    blobs.append(m.run_windows()[0])
                assert m.combine(blobs) == want[mode], (fold, mode, "split")
        # small scalars: set_host with scalars < 2^16 -> only window 0 holds anything
        pts = [p for p, _ in m.fetch(0, 64)]
        sc = [(7 * i + 1) for i in range(64)]
        from oracle.pyref.bn254 import G1
        acc = G1.msm_naive(pts, sc)
        for fold in ("host", "device"):
            monkeypatch.setenv("RLNAMD_MSM_FOLD", fold)
            assert m.msm(pts, sc) == acc, fold
    finally:
        m.close()


def test_msm_g2_vs_oracle_small_generated_split_and_skewed(monkeypatch):
    """north_star: "windowed Pippenger MSM on G1/G2" (`partial_proof.rs:98-104` is generic over the group).  The same
    kernels on the twist (rlnamd_msm_new_g2): msm_bigint semantics on small inputs against the Python oracle with the edge
    cases (empty, zero scalars, infinity bases, repeated and opposite points, maximal scalar: doubling and cancellation
    inside the G2 law); the generated workload P_i = k_i G2 at 2^16 / 2^18 against oracle/c's closed form
    (sum k_i s_i) G2 -- whole, as a 3-way split combined, with every scalar equal (one bucket per window: the big-bucket
    join) and four distinct bases -- with the fold on the host and on the device; generated points spot-checked"""
    from oracle.c import binding as ob
    from oracle.pyref.bn254 import G2, G2_GEN
    from zerokit_amd.batch import MsmG2
    rnd = random.Random(22)
    m = MsmG2(1 << 18)
    try:
        pts = [G2.mul(G2_GEN, rnd.randrange(1, R)) for _ in range(24)]
        sc = [rnd.randrange(R) for _ in range(24)]
        assert m.msm(pts, sc) == G2.msm_naive(pts, sc)
        assert m.msm([], []) is None
        assert m.msm(pts[:3], [0, 0, 0]) is None
        assert m.msm([None, pts[0]], [5, 7]) == G2.mul(pts[0], 7)
        assert m.msm([pts[0]] * 5, [1, 2, 3, 4, 5]) == G2.mul(pts[0], 15)
        assert m.msm([pts[0], G2.neg(pts[0])], [9, 9]) is None
        assert m.msm([pts[1]], [R - 1]) == G2.neg(pts[1])
        assert m.msm([pts[4]] * 200, [7] * 200) == G2.mul(pts[4], 1400)
        seed = 0x62
        for fold in ("host", "device"):
            monkeypatch.setenv("RLNAMD_MSM_FOLD", fold)
            n = 1 << 16
            m.generate(seed, 0, n)
            assert m.fetch(n - 1, 1)[0] == ob.msm_workload_item_g2(seed, n - 1)
            want = ob.msm_expected_g2(seed, 0, n)
            assert m.combine([m.run_windows()[0]]) == want, fold
            blobs = []
            for r in range(3):
                lo, hi = n * r // 3, n * (r + 1) // 3
                m.generate(seed, lo, hi - lo)
                blobs.append(m.run_windows()[0])
            assert m.combine(blobs) == want, (fold, "split")
        monkeypatch.delenv("RLNAMD_MSM_FOLD")
        n = 1 << 18
        for mode in (0, 1, 2, 3):
            m.generate(seed, 7, n, mode)
            assert m.combine([m.run_windows()[0]]) == ob.msm_expected_g2(seed, 7, n, mode), mode
    finally:
        m.close()


def test_msm_g1_config5_full_size_2_24_vs_oracle():
    """BASELINE config 5 at FULL size: 2^24 generated points on one device against the oracle's closed form; the same
    points as EIGHT 2^21 slices (the shards of the 8-way split of BASELINE.json) each against the oracle by itself and
    recombined from their window sums; and the all-equal-scalars distribution at full size (16.7 M points in one bucket
    per window: the oversized-bucket join of k_slice_fix_big)."""
    from oracle.c import binding as ob
    from zerokit_amd.batch import MsmG1
    n, seed = 1 << 24, 0xC0FFEE
    want = ob.msm_expected(seed, 0, n)
    m = MsmG1(n)
    m.generate(seed, 0, n)
    assert m.fetch(n - 1, 1)[0] == ob.msm_workload_item(seed, n - 1)
    blob, ms = m.run_windows()
    assert m.combine([blob]) == want
    blobs, q = [], n // 8
    for r in range(8):
        m.generate(seed, r * q, q)
        blobs.append(m.run_windows()[0])
        assert m.combine(blobs[-1:]) == ob.msm_expected(seed, r * q, q)      # every shard by itself
    assert m.combine(blobs) == want
    m.generate(seed, 0, n, MsmG1.EQUAL_SCALARS)
    assert m.combine([m.run_windows()[0]]) == ob.msm_expected(seed, 0, n, ob.MSM_EQUAL_SCALARS)
    m.close()


# ------------------------------------------------------------------------------------------ other circuits
def test_other_circuits_depth10_and_multi_message_id():
    """The prover is circuit-generic: the shipped depth-10 single circuit and the depth-20 multi-message-id
    (max_out 4) circuit (extra graph ops Neq / Div / TernCond / Neg; 16 public inputs in the order of
    proof.rs:870-885) reproduce the oracle's witness digest, public signals and proof bytes, and verify."""
    from zerokit_amd.batch import BatchProver
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_other_circuits.json")))["cases"]
    for c in cases:
        p = BatchProver(max_batch=64, depth=c["depth"], multi=c["multi"])
        named = {k: [int(v) for v in vs] for k, vs in c["inputs"].items()}
        n = p.upload(p.pack_named_inputs([named, named]), [(int(c["r"]), int(c["s"])), (1, 2)])
        p.run(n)
        out = p.download(n)
        assert out[0]["error"] == 0
        pub = p.download_public(n)
        assert [str(v) for v in pub[0]] == c["public"], c["name"]
        assert _digest(p.fetch_witness(0)) == c["witness_sha256"], c["name"]
        assert out[0]["proof"].hex() == c["proof_compressed"], c["name"]
        assert p.verify_public(out[0]["proof"], pub[0]) and p.verify_public(out[1]["proof"], pub[1])
        assert out[1]["proof"] != out[0]["proof"] and pub[1] == pub[0]
        bad = list(pub[0])
        bad[-1] ^= 1
        assert not p.verify_public(out[0]["proof"], bad)
        if not c["multi"]:
            assert out[0]["public_inputs"] == pub[0]          # Poseidon-formula values == witness outputs
        # round 6: one proof per call through the segments behind hints, on these circuits too (the multi-message-id
        # circuit: an a1 hint per message slot): the golden bytes, every hint checked, none failed
        st = p.hint_stats()
        assert st["segments"] > c["depth"] and st["hints"] == c["depth"] + 1 + (4 if c["multi"] else 1), st
        t, k = p.submit(p.pack_named_inputs([named]), p.pack_rs([(int(c["r"]), int(c["s"]))]))
        got = p.collect_raw(t, k)
        assert got[0].hex() == c["proof_compressed"] and not any(got[2]), c["name"]
        st = p.hint_stats()
        assert st["hinted_batches"] == 1 and st["fallbacks"] == 0, st
        p.close()


@pytest.mark.parametrize("depth,multi", [(10, False), (20, True)])
def test_other_circuits_throughput_shape_vs_oracle(depth, multi):
    """the path that makes the headline number (more than 128 proofs per batch: lanes = proofs walks,
    pair chunks over the rows that share a witness scalar, GLV halves, the lanes = proofs interpreter and NTT passes) on
    the OTHER shipped circuits -- depth 10 single, depth 20 multi-message-id (max_out 4: 15 public inputs in the order of
    proof.rs:870-885, witness.rs:117-180) -- on the default tables and on a wide schedule (G1 14 + 9 x 13 bits, G2
    3 x 16 + 6 x 15 bits: digits at both ends of the int16 range).  n = 129 (one proof past two waves) and 200, resident
    and streamed: every proof's bytes and every public input equal oracle/c's (itself pinned to the pyref goldens of
    these circuits in tests/test_oracle_c.py), every proof verifies on the host against those public inputs, and the
    <= 64-proof latency shape of the same witnesses gives the same bytes."""
    from oracle.c import binding as ob
    from zerokit_amd import workload
    from zerokit_amd.batch import BatchProver
    o = ob.Circuit(depth, multi)
    N = 200
    named, rs = workload.circuit_range(1000, N, depth, multi)
    rsb = b"".join(r.to_bytes(32, "little") + s.to_bytes(32, "little") for r, s in rs)
    _, ref_proofs, ref_pub = o.prove_many_packed(b"".join(o.pack_named(w) for w in named), rsb)
    assert len(set(ref_proofs)) == N
    for wb in (0, 3150113):
        p = BatchProver(max_batch=256, window_bits=wb, depth=depth, multi=multi)
        try:
            assert p.slots == o.slots and p.num_public == o.n_public
            inp = p.pack_named_inputs(named)
            assert inp == b"".join(o.pack_named(w) for w in named)
            per = p.inputs_size * 32
            for n in (129, 200, 64, 3):
                k = p.upload(inp[:n * per], rs[:n])
                p.run(k)
                out, pub = p.download(k), p.download_public(k)
                assert all(x["error"] == 0 for x in out), (wb, n)
                assert [x["proof"] for x in out] == ref_proofs[:n], (wb, n)
                assert pub == ref_pub[:n], (wb, n)
                if n == 200:
                    assert all(p.verify_public(out[i]["proof"], pub[i]) for i in range(n)), wb
                    bad = list(pub[7])
                    bad[0] ^= 1
                    assert not p.verify_public(out[7]["proof"], bad)
            # streamed (submit / collect), the public signals taken before the collect wipes the witness
            import ctypes as C
            from zerokit_amd import lib
            from zerokit_amd._native import check
            for n in (200, 129):
                t, k = p.submit(inp[:n * per], rsb[:64 * n])
                buf = C.create_string_buffer(32 * p.num_public * n)
                check(lib().rlnamd_prover_collect_public(p._h, t, n, buf))
                got = p.collect(t, k)
                assert [g["proof"] for g in got] == ref_proofs[:n], (wb, n)
                q = p.num_public
                assert [[int.from_bytes(buf.raw[32 * (i * q + j):32 * (i * q + j + 1)], "little") for j in range(q)]
                        for i in range(n)] == ref_pub[:n], (wb, n)
        finally:
            p.close()


def test_both_graph_interpreters_give_the_golden_witness(monkeypatch):
    """The graph interpreter runs in the 9 x 29-bit limb form by default (k_witness29: static value bounds and G_RED
    reductions chosen on the host, stored signals converted by k_v29_to_fr); RLNAMD_WIT29=0 keeps the 8 x 32 one.  Both
    must reproduce the oracle's witness digests -- the depth-20 goldens, and the multi message-id circuit whose graph
    carries the slow operations (Shr / Band / Neq / Div on canonical integers), TernCond and Neg."""
    from zerokit_amd.batch import BatchProver
    cases = _cases()["cases"]
    other = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "rln_other_circuits.json")))["cases"]
             if c["multi"]]
    for flag in ("1", "0"):
        monkeypatch.setenv("RLNAMD_WIT29", flag)
        p = BatchProver(max_batch=64)
        try:
            out = p.prove([_w(c) for c in cases], [(int(c["r"]), int(c["s"])) for c in cases])
            for i, c in enumerate(cases):
                assert out[i]["error"] == 0 and out[i]["proof"].hex() == c["proof_compressed"], (flag, c["name"])
                assert _digest(p.fetch_witness(i)) == c["witness_sha256"], (flag, c["name"])
        finally:
            p.close()
        for c in other:
            p = BatchProver(max_batch=64, depth=c["depth"], multi=True)
            try:
                named = {k: [int(v) for v in vs] for k, vs in c["inputs"].items()}
                n = p.upload(p.pack_named_inputs([named]), [(int(c["r"]), int(c["s"]))])
                p.run(n)
                out = p.download(n)
                assert out[0]["error"] == 0 and out[0]["proof"].hex() == c["proof_compressed"], (flag, c["name"])
                assert _digest(p.fetch_witness(0)) == c["witness_sha256"], (flag, c["name"])
            finally:
                p.close()


def test_lane_chunk_walk_and_clock_tap(monkeypatch):
    """Batches of at most RLNAMD_LANECHUNK (128) proofs take the small-batch shapes (lanes = chunks walks, walk29.h; a
    wave per proof in the interpreter; early walks), larger ones the lanes = proofs pipeline.  The same witnesses must give the same proof bytes either way -- alone, inside a batch that is
    walked the other way, and with the threshold forced to 0 -- and the clock tap of the walk kernels must report a
    plausible shader clock after a lanes = proofs run."""
    from zerokit_amd.batch import BatchProver
    cases = _cases()["cases"]
    ws, rs = [_w(c) for c in cases], [(int(c["r"]), int(c["s"])) for c in cases]
    p = BatchProver(max_batch=192)
    try:
        small = p.prove(ws[:2], rs[:2])                       # lanes = chunks
        reps = 160 // len(ws) + 1
        mid = p.prove((ws * reps)[:100], (rs * reps)[:100])   # still the small-batch shapes, two waves of proofs
        big = p.prove((ws * reps)[:160], (rs * reps)[:160])   # lanes = proofs
        clk = p.walk_clock_mhz()
        assert 500.0 < clk["g1_walk"] < 3000.0 and 500.0 < clk["g2_walk"] < 3000.0, clk
        for i in range(2):
            assert small[i]["proof"].hex() == cases[i]["proof_compressed"]
        for i in range(100):
            assert mid[i]["proof"].hex() == cases[i % len(ws)]["proof_compressed"], i
        for i in range(160):
            assert big[i]["proof"].hex() == cases[i % len(ws)]["proof_compressed"], i
    finally:
        p.close()
    monkeypatch.setenv("RLNAMD_LANECHUNK", "0")
    p = BatchProver(max_batch=64)
    try:
        out = p.prove(ws[:2], rs[:2])
        assert [o["proof"].hex() for o in out] == [c["proof_compressed"] for c in cases[:2]]
    finally:
        p.close()


@pytest.mark.parametrize("env", [
    {"RLNAMD_LONE": "0"},             # as inside a stream of batches: plain small plan, back end on its own stream
    {"RLNAMD_LONE": "1"},
    {"RLNAMD_VALUES_WITNESS": "0"},   # proof values by the Poseidon chain instead of the circuit's public signals
    {"RLNAMD_WL_REASSOC": "0"},       # the interpreter's schedule with the circuit's sums in source order
    {"RLNAMD_WITROWS": "0"},          # lane-form products
    {"RLNAMD_EARLY_FIN": "0", "RLNAMD_FUSED_SMUL": "0"},
    {"RLNAMD_LANECHUNK_WALK": "0"},   # the short-chunk plans walked with lanes = proofs (the shape of 49..128 proofs)
    {"RLNAMD_TINY": "0"},             # no one-entry chunks / two-stage sums: the small-batch plan for a single proof too
    {"RLNAMD_TINY": "8"},             # ... and that shape for the batch of five as well
], ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_small_batch_shape_variants_give_the_golden_bytes(monkeypatch, env):
    """Every latency shape of the single-proof path has a switch that restores the shape it replaced (DESIGN section 4,
    "Small batches"); each combination must give the golden proof bytes, public inputs and witness digest -- for one
    proof and for a batch of five, also when the batch is NOT alone on the device (two batches back to back)."""
    import hashlib
    from zerokit_amd.batch import BatchProver
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cases = _cases()["cases"]
    ws, rs = [_w(c) for c in cases], [(int(c["r"]), int(c["s"])) for c in cases]
    p = BatchProver(max_batch=64)
    try:
        # the switches are read once, when the prover is built, and rlnamd_prover_describe reports the values in force
        names = {"RLNAMD_LONE": "lone", "RLNAMD_VALUES_WITNESS": "values_from_witness",
                 "RLNAMD_EARLY_FIN": "early_fin", "RLNAMD_FUSED_SMUL": "fused_smul", "RLNAMD_LANECHUNK_WALK": "lanechunk_walk",
                 "RLNAMD_TINY": "tiny"}
        desc = p.describe().split()
        for k, v in env.items():
            if k in names:
                assert "%s=%s" % (names[k], v) in desc, (k, desc)
        one = p.prove(ws[:1], rs[:1])[0]
        assert one["proof"].hex() == cases[0]["proof_compressed"]
        assert [str(v) for v in one["public_inputs"]] == cases[0]["public_inputs"]
        full = p.fetch_witness(0)
        assert hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in full)).hexdigest() == cases[0]["witness_sha256"]
        five = p.prove(ws[:5], rs[:5])
        for i in range(5):
            assert five[i]["proof"].hex() == cases[i]["proof_compressed"], i
            assert [str(v) for v in five[i]["public_inputs"]] == cases[i]["public_inputs"], i
        # two batches in flight: the second is enqueued while the first still runs
        inp, rsb = p.pack_inputs(ws[:3]), p.pack_rs(rs[:3])
        t1, _ = p.submit(inp, rsb)
        t2, _ = p.submit(inp, rsb)
        a, b = p.collect(t1, 3), p.collect(t2, 3)
        for i in range(3):
            assert a[i]["proof"].hex() == cases[i]["proof_compressed"] == b[i]["proof"].hex(), i
    finally:
        p.close()


# ------------------------------------------------------------------------------------------ partial proofs
def test_partial_proof_then_finish_equals_full(prover):
    """generate_partial_zk_proof + finish_zk_proof_with_rs == generate_zk_proof_with_rs
    (rln/tests/protocol.rs:222-248, here for a batch and against the oracle):
    the mask equals the oracle's evaluate_partial knownness, the four partial points equal the oracle's,
    and finishing with (r, s) reproduces the full proof bytes and the golden vector."""
    from oracle.pyref import groth16, rln, wtns_graph
    cases = _cases()["cases"][:3]
    ws = [_w(c) for c in cases]
    rs = [(int(c["r"]), int(c["s"])) for c in cases]
    zk, g = rln.load_circuit(20)
    mask = groth16.known_mask(g)
    assert prover.known_mask() == [1 if m else 0 for m in mask]
    pws = [dict(identity_secret=w["identity_secret"], user_message_limit=w["user_message_limit"],
                path_elements=w["path_elements"], identity_path_index=w["identity_path_index"]) for w in ws]
    partials = prover.prove_partial(pws)
    # oracle partial for the first witness
    w0 = ws[0]
    full = wtns_graph.calc_witness(g, rln.WitnessInput(w0["identity_secret"], w0["user_message_limit"],
                                                       w0["message_id"], w0["path_elements"],
                                                       w0["identity_path_index"], w0["x"],
                                                       w0["external_nullifier"]).named_inputs())
    pa, rho, pb, pc = groth16.prove_partial(zk, full, mask)
    want = b"".join(v.to_bytes(32, "little") for v in (pa[0], pa[1], rho[0], rho[1], pb[0][0], pb[0][1], pb[1][0],
                                                        pb[1][1], pc[0], pc[1]))
    assert partials[0] == want
    out = prover.finish(ws, rs, partials)
    full_out = prover.prove(ws, rs)
    for o, f, c in zip(out, full_out, cases):
        assert o["proof"] == f["proof"] == bytes.fromhex(c["proof_compressed"])
        assert o["public_inputs"] == f["public_inputs"]
    # the partial proof is message-independent: reuse it for a different message of the same member
    w_new = dict(ws[1], x=ws[1]["x"] ^ 1, message_id=2, external_nullifier=7)
    o = prover.finish([w_new], [(5, 6)], [partials[1]])[0]
    assert o["proof"] == prover.prove([w_new], [(5, 6)])[0]["proof"] and prover.verify(o["proof"], o["public_inputs"])


def test_finish_from_a_cached_partial_interprets_only_the_unknown_cone(prover, monkeypatch):
    """Round 6.  rlnamd_prover_collect_partial_cached keeps the stored values the partial witness fixes
    on the device and hands back a handle; rlnamd_prover_submit_finish with live handles restores them and interprets only
    the cone evaluate_partial (graph.rs:274-312) leaves unknown.  Checked: the partial points equal the pyref fixture and
    oracle/c's; a finish through the cone, a finish with no handle and a finish with a stale handle all give the golden
    FULL proof bytes (and oracle/c's finish); the cone counter moves only for live handles; the partial proof is
    message-independent through the cone too; a batch whose handles are not all live falls back; a batch of 37 through
    the cone equals oracle/c's finish proof by proof; released entries are wiped (residue 0) and reused."""
    from oracle.c import binding as ob
    from zerokit_amd import workload
    from zerokit_amd.batch import BatchProver
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_partial.json")))
    cases = {c["name"]: c for c in _cases()["cases"]}
    oc = ob.Circuit(20)
    p = prover
    info0 = p.partial_cache_info()
    assert info0["capacity"] >= 64 and info0["in_use"] == 0 and info0["cone_nodes"] > 1934 and info0["cone_steps"] * 8 < info0["full_steps"]
    assert info0["entry_bytes"] > 5000 * 48

    def partial_inputs(ws):
        return p.pack_inputs([dict(w, message_id=0, x=0, external_nullifier=0) for w in ws])

    def run_partial(ws):
        t, n = p.submit(partial_inputs(ws), bytes(64 * len(ws)), 1)
        return p.collect_partial_cached(t, n)

    def run_finish(ws, rs, parts, handles):
        t, n = p.submit_finish(p.pack_inputs(ws), p.pack_rs(rs), parts, handles)
        return p.collect(t, n)
    for f in fx["cases"]:
        c = cases[f["name"]]
        w, rs = _w(c), (int(c["r"]), int(c["s"]))
        parts, hs, errs = run_partial([w])
        assert hs[0] != 0 and errs == [0] and parts[0].hex() == f["partial320"] == oc.prove_partial_packed(oc.pack(w)).hex()
        before = p.partial_cache_info()["cone_batches"]
        golden = bytes.fromhex(c["proof_compressed"])
        o = run_finish([w], [rs], parts, hs)[0]
        assert o["proof"] == golden == oc.finish_packed(oc.pack(w), rs[0], rs[1], parts[0]) and o["error"] == 0
        assert [str(v) for v in o["public_inputs"]] == c["public_inputs"]
        assert p.partial_cache_info()["cone_batches"] == before + 1
        assert run_finish([w], [rs], parts, [0])[0]["proof"] == golden          # no handle: the whole graph
        assert p.partial_cache_info()["cone_batches"] == before + 1
        # the partial proof is message-independent, through the cone too
        w2 = dict(w, x=w["x"] ^ 5, message_id=7, external_nullifier=99)
        o2 = run_finish([w2], [(5, 6)], parts, hs)[0]
        assert o2["proof"] == p.prove([w2], [(5, 6)])[0]["proof"] and p.verify(o2["proof"], o2["public_inputs"])
        assert p.partial_cache_info()["cone_batches"] == before + 2
        p.release_partial(hs)
        assert run_finish([w], [rs], parts, hs)[0]["proof"] == golden           # stale handle: the whole graph
        info = p.partial_cache_info()
        assert info["cone_batches"] == before + 2 and info["in_use"] == 0 and info["residue_in_free_entries"] == 0
    # a batch through the cone against oracle/c's finish, every proof; then the same with one dead handle (falls back)
    n = 37
    ws, rs = workload.config2_range(5000, n)
    parts, hs, errs = run_partial(ws)
    assert all(hs) and not any(errs) and len(set(hs)) == n and p.partial_cache_info()["in_use"] == n
    packed = [oc.pack(w) for w in ws]
    _, want = oc.finish_many_packed(b"".join(packed), p.pack_rs(rs), b"".join(oc.prove_partial_packed(q) for q in packed))
    before = p.partial_cache_info()["cone_batches"]
    out = run_finish(ws, rs, parts, hs)
    assert [o["proof"] for o in out] == want and not any(o["error"] for o in out)
    assert [o["public_inputs"] for o in out] == [oc.public_values(q) for q in packed]
    assert p.partial_cache_info()["cone_batches"] == before + 1
    dead = list(hs)
    p.release_partial(dead[3:4])
    assert [o["proof"] for o in run_finish(ws, rs, parts, dead)] == want
    assert p.partial_cache_info()["cone_batches"] == before + 1
    p.release_partial(hs)
    info = p.partial_cache_info()
    assert info["in_use"] == 0 and info["residue_in_free_entries"] == 0
    # every shape boundary of the finish path (tiny <= 5, fused <= 96, the lanes = proofs walks above 48, the cone up to the
    # small-batch threshold 128): through the cone, byte-identical to the FULL proofs of the same prover
    ws, rs = workload.config2_range(7000, 128)
    full = p.prove(ws, rs)
    for n in (2, 5, 6, 16, 48, 49, 96, 97, 128):
        parts_n, hs_n, _ = run_partial(ws[:n])
        assert all(hs_n)
        before = p.partial_cache_info()["cone_batches"]
        out = run_finish(ws[:n], rs[:n], parts_n, hs_n)
        assert [o["proof"] for o in out] == [f["proof"] for f in full[:n]], n
        assert [o["public_inputs"] for o in out] == [f["public_inputs"] for f in full[:n]]
        assert p.partial_cache_info()["cone_batches"] == before + 1
        p.release_partial(hs_n)
    ws, rs = workload.config2_range(5000, 37)
    # a cache smaller than the batch: a prefix gets handles, the batch finishes through the whole graph, same bytes
    monkeypatch.setenv("RLNAMD_PARTIAL_CACHE", "3")
    q = BatchProver(max_batch=64, window_bits=8)
    try:
        t, k = q.submit(q.pack_inputs([dict(w, message_id=0, x=0, external_nullifier=0) for w in ws[:5]]), bytes(64 * 5), 1)
        parts5, hs5, _ = q.collect_partial_cached(t, k)
        assert [bool(h) for h in hs5] == [True, True, True, False, False] and parts5 == parts[:5]
        t, k = q.submit_finish(q.pack_inputs(ws[:5]), q.pack_rs(rs[:5]), parts5, hs5)
        assert [o["proof"] for o in q.collect(t, k)] == want[:5] and q.partial_cache_info()["cone_batches"] == 0
        t, k = q.submit_finish(q.pack_inputs(ws[:3]), q.pack_rs(rs[:3]), parts5[:3], hs5[:3])
        assert [o["proof"] for o in q.collect(t, k)] == want[:3] and q.partial_cache_info()["cone_batches"] == 1
        # another prover's handles mean nothing here, even where index and generation coincide (both caches just started):
        # the whole graph is walked, the bytes are the same
        parts3, hs3, _ = run_partial(ws[:3])
        assert all(hs3) and set(hs3).isdisjoint(hs5[:3])
        before = p.partial_cache_info()["cone_batches"]
        assert [o["proof"] for o in run_finish(ws[:3], rs[:3], parts3, hs5[:3])] == want[:3]
        assert p.partial_cache_info()["cone_batches"] == before
        p.release_partial(hs3)
    finally:
        q.close()


def test_a_few_proofs_per_call_interpret_the_graph_as_segments_behind_hints(monkeypatch):
    """Round 6.  A lone batch of one or two proofs: the calling thread computes the values between the circuit's 22 chained
    hashes (rlnamd_prover_hint_stats; the library's host Poseidon), the device interprets the 23 segments those values
    separate at once and compares every cut node's own value with its hint.  Checked: the golden proof bytes and public
    inputs (all golden cases, r = 0 among them), one, two, eight, nine and twenty-four proofs per call; twenty-five per call
    keep the whole-graph interpreter; a partial proof through the segments equals the pyref fixture; an input >= r is still an
    error; with a
    corrupted hint (test hook) the batch is run again over the whole graph and the caller sees the golden bytes; with
    RLNAMD_HINTS=0 nothing is hinted and the bytes are the same; a member proving again at the same root finds the chain
    part of its hints remembered (hint_stats()["chains_remembered"]) and gets the same bytes; 25 and 64 proofs per call take
    the segments when the members' chains are remembered and the whole graph when they are not; hints computed ahead of
    the call (hints_for / submit_hinted) give the golden bytes, and so do hints that belong to other inputs (one rerun)."""
    from zerokit_amd.batch import BatchProver
    cases = _cases()["cases"]
    fx = {c["name"]: c["partial320"] for c in json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_partial.json")))["cases"]}
    monkeypatch.setenv("RLNAMD_HINTS_WARM", "0")            # (first without the rule for batches above 24: see the end)

    def run(p, idx):
        ws = [_w(cases[i]) for i in idx]
        rs = [(int(cases[i]["r"]), int(cases[i]["s"])) for i in idx]
        t, n = p.submit(p.pack_inputs(ws), p.pack_rs(rs))
        out = p.collect(t, n)
        for o, i in zip(out, idx):
            assert o["proof"].hex() == cases[i]["proof_compressed"] and o["error"] == 0, cases[i]["name"]
            assert [str(v) for v in o["public_inputs"]] == cases[i]["public_inputs"]
    p = BatchProver(max_batch=64, window_bits=8)
    try:
        st = p.hint_stats()
        assert st["segments"] >= 22 and st["hints"] == 22 and st["longest_segment_steps"] * 8 < st["full_steps"], st
        for i in range(len(cases)):
            run(p, [i])
        assert p.hint_stats()["hinted_batches"] == len(cases) and p.hint_stats()["fallbacks"] == 0
        seen = p.hint_stats()["chains_remembered"]          # members the golden cases share count here already
        run(p, [0, 1])
        run(p, [2, 3])
        assert p.hint_stats()["hinted_batches"] == len(cases) + 2
        # the same members at the same root again: their chains of hints are remembered (two host hashes each), same bytes
        assert p.hint_stats()["chains_remembered"] == seen + 4, p.hint_stats()
        run(p, [0, 1, 2, 3, 4, 5, 0, 1])                   # eight per call: still segments (a host thread per proof hashes its hints)
        assert p.hint_stats()["hinted_batches"] == len(cases) + 3
        run(p, [0, 1, 2, 3, 4, 5, 0, 1, 2])                # nine per call: segments as well since the chains are shared out
        assert p.hint_stats()["hinted_batches"] == len(cases) + 4      # over eight host threads (RLNAMD_HINTS = 24)
        run(p, [i % len(cases) for i in range(24)])
        assert p.hint_stats()["hinted_batches"] == len(cases) + 5
        run(p, [i % len(cases) for i in range(25)])        # twenty-five per call: the whole graph
        assert p.hint_stats()["hinted_batches"] == len(cases) + 5
        for c in cases:                                    # a partial proof through the segments
            if c["name"] in fx:
                t, n = p.submit(p.pack_inputs([dict(_w(c), message_id=0, x=0, external_nullifier=0)]), bytes(64), 1)
                assert p.collect_partial(t, 1)[0].hex() == fx[c["name"]]
        assert p.hint_stats()["hinted_batches"] == len(cases) + 5 + len(fx) and p.hint_stats()["fallbacks"] == 0
        bad = dict(_w(cases[0]), x=R)                      # x = r: not canonical (graph.rs:42-45)
        inp = bytearray(p.pack_inputs([dict(bad, x=0)]))
        off = p.slots["x"][0]
        inp[32 * off:32 * off + 32] = R.to_bytes(32, "little")
        t, n = p.submit(bytes(inp), p.pack_rs([(1, 2)]))
        assert p.collect(t, n)[0]["error"] != 0
    finally:
        p.close()
    # the hints computed ahead of the call (rlnamd_prover_hints_for / _submit_hinted): 40 proofs in one batch take the
    # segments although their chains were never seen; hints that belong to other inputs cost a rerun, not a wrong proof
    p = BatchProver(max_batch=64, window_bits=8)
    try:
        idx = [i % len(cases) for i in range(40)]
        inp = p.pack_inputs([_w(cases[i]) for i in idx])
        rsb = p.pack_rs([(int(cases[i]["r"]), int(cases[i]["s"])) for i in idx])
        hints = p.hints_for(inp)
        assert hints is not None and len(hints) == 40 and len(hints[0]) == 22 * 8
        t, n = p.submit_hinted(inp, rsb, hints)
        for o, i in zip(p.collect(t, n), idx):
            assert o["proof"].hex() == cases[i]["proof_compressed"] and o["error"] == 0
        st = p.hint_stats()
        assert st["hinted_batches"] == 1 and st["fallbacks"] == 0, st
        t, n = p.submit_hinted(inp, rsb, hints[1:] + hints[:1])     # every proof with its neighbour's hints
        for o, i in zip(p.collect(t, n), idx):
            assert o["proof"].hex() == cases[i]["proof_compressed"] and o["error"] == 0
        st = p.hint_stats()
        assert st["hinted_batches"] == 2 and st["fallbacks"] == 1, st
    finally:
        p.close()
    monkeypatch.setenv("RLNAMD_HINT_FAULT", "9")
    p = BatchProver(max_batch=64, window_bits=8)
    try:
        run(p, [0])
        run(p, [1, 2])
        assert p.hint_stats()["hinted_batches"] == 2 and p.hint_stats()["fallbacks"] == 2
    finally:
        p.close()
    monkeypatch.delenv("RLNAMD_HINT_FAULT")
    monkeypatch.setenv("RLNAMD_HINT_CHAINS", "0")           # nothing remembered between calls: every chain hashed, same bytes
    p = BatchProver(max_batch=64, window_bits=8)
    try:
        run(p, [0])
        run(p, [0])
        assert p.hint_stats()["hinted_batches"] == 2 and p.hint_stats()["chains_remembered"] == 0
    finally:
        p.close()
    monkeypatch.delenv("RLNAMD_HINT_CHAINS")
    # above 24 proofs per call (up to 64) the segments are taken when the members' chains are remembered: a batch of
    # members never seen keeps the whole graph, the same 25 after their chains were hashed once do not -- same bytes
    monkeypatch.delenv("RLNAMD_HINTS_WARM")
    p = BatchProver(max_batch=64, window_bits=8)
    try:
        run(p, [i % len(cases) for i in range(25)])
        assert p.hint_stats()["hinted_batches"] == 0
        run(p, list(range(len(cases))))
        run(p, [i % len(cases) for i in range(25)])
        run(p, [i % len(cases) for i in range(64)])
        st = p.hint_stats()
        assert st["hinted_batches"] == 3 and st["fallbacks"] == 0 and st["chains_remembered"] >= 25 + 64, st
    finally:
        p.close()
    monkeypatch.setenv("RLNAMD_HINTS", "0")
    p = BatchProver(max_batch=64, window_bits=8)
    try:
        run(p, [0])
        assert p.hint_stats()["hinted_batches"] == 0 and p.hint_stats()["segments"] == 0
    finally:
        p.close()


def test_small_batches_in_flight_behind_each_other_keep_the_latency_shapes_and_their_bytes():
    """Round 6.  Batches of at most 48 proofs take the lone (latency) shapes -- segments behind hints, fused / tiny plans,
    lanes = chunks walks -- even while earlier batches are still in flight (RLNAMD_LONE_SMALL; a stream of them was 1.2 - 2 x
    slower in the throughput shapes).  Three batches in flight at a time, sizes 1 ... 48, full proofs and finishes of cached
    partial proofs mixed: every proof equals the golden bytes / the proof of a call that ran alone, every public input too;
    the hinted batches were hinted although they were not alone."""
    from zerokit_amd.batch import BatchProver
    cases = _cases()["cases"]
    p = BatchProver(max_batch=64, window_bits=8)
    try:
        def pack(idx):
            ws = [_w(cases[i % len(cases)]) for i in idx]
            rs = [(int(cases[i % len(cases)]["r"]), int(cases[i % len(cases)]["s"])) for i in idx]
            return p.pack_inputs(ws), p.pack_rs(rs)

        def check(out, idx):
            for o, i in zip(out, idx):
                c = cases[i % len(cases)]
                assert o["proof"].hex() == c["proof_compressed"] and o["error"] == 0, (c["name"], len(idx))
                assert [str(v) for v in o["public_inputs"]] == c["public_inputs"]
        before = p.hint_stats()["hinted_batches"]
        sizes = [1, 5, 8, 2, 16, 24, 3, 33, 48, 1, 12, 7]
        flight = []
        for k, n in enumerate(sizes):
            idx = [(k + j) for j in range(n)]
            if len(flight) == 3:
                t, m, ix = flight.pop(0)
                check(p.collect(t, m), ix)
            inp, rsb = pack(idx)
            t, m = p.submit(inp, rsb)
            flight.append((t, m, idx))
        while flight:
            t, m, ix = flight.pop(0)
            check(p.collect(t, m), ix)
        st = p.hint_stats()
        assert st["hinted_batches"] - before >= sum(1 for n in sizes if n <= 24) and st["fallbacks"] == 0, st
    finally:
        p.close()


def test_partial_cache_lifetimes_under_a_random_sequence_of_calls(monkeypatch):
    """The cache's lifetimes (entries written on the wipe stream, read on the front-end stream and by k_pp_smul, wiped on
    release, indices reused with a new generation) under 120 pseudo-random calls on a prover with EIGHT entries: partial
    batches of 1 - 6 members (a prefix gets handles when the cache runs out), finishes of random members with whatever
    handle they hold (live, released, never given), releases, back to back without a sync between them.  Every finished
    proof must equal the member's FULL proof for the same (r, s); at the end everything is released and the free entries
    are zero."""
    from zerokit_amd import workload
    from zerokit_amd.batch import BatchProver
    monkeypatch.setenv("RLNAMD_PARTIAL_CACHE", "8")
    p = BatchProver(max_batch=64, window_bits=8)
    try:
        M = 12
        ws, rs = workload.config2_range(31000, M)
        full = [o["proof"] for o in p.prove(ws, rs)]
        parts, handles = [None] * M, [0] * M
        rnd = random.Random(20261003)
        cone_before = p.partial_cache_info()["cone_batches"]
        finishes = 0
        for step in range(120):
            op = rnd.random()
            if op < 0.3:                                   # a partial batch of a few members
                ids = rnd.sample(range(M), rnd.randint(1, 6))
                p.release_partial([handles[i] for i in ids if handles[i]])
                for i in ids:
                    handles[i] = 0
                t, n = p.submit(p.pack_inputs([dict(ws[i], message_id=0, x=0, external_nullifier=0) for i in ids]), bytes(64 * len(ids)), 1)
                pp, hs, errs = p.collect_partial_cached(t, n)
                assert not any(errs)
                seen_zero = False
                for i, q, h in zip(ids, pp, hs):
                    parts[i], handles[i] = q, h
                    assert not (seen_zero and h), "handles must be a prefix of the batch"
                    seen_zero = seen_zero or h == 0
            elif op < 0.85:                                # a finish of members that have a partial proof
                have = [i for i in range(M) if parts[i] is not None]
                if not have:
                    continue
                ids = rnd.sample(have, min(len(have), rnd.randint(1, 5)))
                hs = [handles[i] if rnd.random() < 0.9 else 0 for i in ids]
                t, n = p.submit_finish(p.pack_inputs([ws[i] for i in ids]), p.pack_rs([rs[i] for i in ids]), [parts[i] for i in ids], hs)
                if rnd.random() < 0.3:                     # released while the finish that reads them is in flight
                    p.release_partial([h for h in hs if h])
                    for i, h in zip(ids, hs):
                        if h:
                            handles[i] = 0
                out = p.collect(t, n)
                assert [o["proof"] for o in out] == [full[i] for i in ids], (step, ids, hs)
                finishes += 1
            else:                                          # release a few, some of them twice
                ids = rnd.sample(range(M), rnd.randint(1, 4))
                p.release_partial([handles[i] for i in ids] + [handles[ids[0]]])
                for i in ids:
                    handles[i] = 0                         # (the partial proof itself stays usable: the whole graph)
            info = p.partial_cache_info() if step % 20 == 19 else None
            if info:
                assert info["in_use"] == sum(1 for h in handles if h) <= 8
        info = p.partial_cache_info()
        assert finishes > 40 and 0 < info["cone_batches"] - cone_before < finishes      # both paths were taken
        p.release_partial([h for h in handles if h])
        info = p.partial_cache_info()
        assert info["in_use"] == 0 and info["residue_in_free_entries"] == 0
    finally:
        p.close()


def test_fq29_group_law_matches_the_8x32_group_law_on_device():
    """csrc/fq29.h (9 x 29-bit unsaturated limbs, the form both fixed-base walks and the Pippenger buckets use)
    against curve.h on 16 384 pseudo-random walks of 96 signed additions each, with repeated points (doubling),
    point / negation pairs (cancellation) and restarts from infinity: bit-identical affine results, G1 and G2"""
    import ctypes as C
    from oracle.pyref.bn254 import G2_GEN
    from zerokit_amd._native import lib, check
    bad = C.c_uint32(123)
    check(lib().rlnamd_selftest_fq29(1, 16384, 96, None, C.byref(bad)))
    assert bad.value == 0
    g2 = b"".join(int(v).to_bytes(32, "little") for v in (G2_GEN[0][0], G2_GEN[0][1], G2_GEN[1][0], G2_GEN[1][1]))
    bad = C.c_uint32(123)
    check(lib().rlnamd_selftest_fq29(2, 4096, 48, g2, C.byref(bad)))
    assert bad.value == 0
    bad = C.c_uint32(123)
    check(lib().rlnamd_selftest_fq29(3, 4096, 48, g2, C.byref(bad)))   # a lane pair per point (Fq2PairOps), incl. its loads / stores
    assert bad.value == 0
    bad = C.c_uint32(123)
    check(lib().rlnamd_selftest_fq29(4, 8192, 0, None, C.byref(bad)))   # G1 general additions by lane pairs (G1AccPair29)
    assert bad.value == 0


def _golden_batch(p, cases, ws, rs, tag):
    out = p.prove(ws, rs)
    for o, c in zip(out, cases):
        assert o["proof"].hex() == c["proof_compressed"], (tag, c["name"])
        assert p.verify(o["proof"], o["public_inputs"])


def test_mixed_window_schedules_give_the_same_proofs(monkeypatch):
    """The comb tables may use (c + 1)-bit windows for the first few windows and different schedules for G1 and G2
    (window_bits = g1 + 10000 * g2, spec = c + 100 * wide), over the 127-bit halves of the GLV split or -- RLNAMD_GLV=0
    -- over the whole 254-bit scalar.  Group elements are canonical, so every schedule must return the golden proof
    bytes."""
    from zerokit_amd.batch import BatchProver
    cases = _cases()["cases"]
    ws, rs = [_w(c) for c in cases], [(int(c["r"]), int(c["s"])) for c in cases]
    # GLV: 7x9+8x8 = 127 bits -> 2 x 15 additions; G1 3x10+11x9 = 129, G2 3x11+10x10 = 133; 10x11+1x10 = 120+..: 12 windows
    for wb, w1, w2 in ((708, 30, 30), (3100309, 28, 26), (1010, 24, 24)):
        p = BatchProver(max_batch=64, window_bits=wb)
        try:
            assert int(p.info.glv) == 1 and int(p.info.windows) == w1 and int(p.info.windows_g2) == w2
            _golden_batch(p, cases, ws, rs, wb)
        finally:
            p.close()
    monkeypatch.setenv("RLNAMD_GLV", "0")
    for wb, w1 in ((708, 31), (1010, 25)):   # 7x9+24x8 = 255, 10x11+15x10 = 260 bits
        p = BatchProver(max_batch=64, window_bits=wb)
        try:
            assert int(p.info.glv) == 0 and int(p.info.windows) == w1 and int(p.info.windows_g2) == w1
            _golden_batch(p, cases, ws, rs, wb)
        finally:
            p.close()


def test_bench_schedule_gives_the_golden_proofs():
    """bench.py's comb schedule (G1: 15 + 8 x 14 bits, G2: 7 x 16 + 15 bits over the GLV halves; 228 GiB of tables):
    the 16-bit windows produce digits at both ends of the int16 range.  Golden proof bytes, and partial + finish
    (which walk subsets of the same tables) must give the full proof."""
    from zerokit_amd.batch import BatchProver
    cases = _cases()["cases"]
    ws, rs = [_w(c) for c in cases], [(int(c["r"]), int(c["s"])) for c in cases]
    try:
        p = BatchProver(max_batch=64, window_bits=7150114)
    except Exception as e:  # noqa: BLE001
        pytest.skip("not enough free HBM for the bench tables: %s" % e)
    try:
        assert int(p.info.windows) == 18 and int(p.info.windows_g2) == 16
        _golden_batch(p, cases, ws, rs, "bench")
        partial = p.prove_partial([{k: w[k] for k in ("identity_secret", "user_message_limit", "path_elements",
                                                       "identity_path_index")} for w in ws])
        fin = p.finish(ws, rs, partial)
        for o, c in zip(fin, cases):
            assert o["proof"].hex() == c["proof_compressed"], c["name"]
    finally:
        p.close()
