"""The reference's own FFI tests (rln/tests/ffi.rs) replayed through the zerokit C ABI of this backend
(include/rln.h via zerokit_amd.public).  Needs the GPU: every tree update and proof runs HIP kernels."""
import random

import pytest

pytestmark = pytest.mark.gpu

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
NO_OF_LEAVES = 64  # the reference uses 256; smaller keeps one-by-one insertion quick


def _leaves(seed=1):
    rnd = random.Random(seed)
    return [rnd.randrange(R) for _ in range(NO_OF_LEAVES)]


@pytest.fixture(scope="module")
def rln():
    from zerokit_amd.public import RLN
    return RLN(20)


def test_merkle_operations_ffi(rln):
    """rln/tests/ffi.rs:119-196: single / next / batch insertion agree; deleting everything restores the
    empty root"""
    leaves = _leaves()
    rln.set_tree(20)
    root_empty = rln.get_root()
    for i, l in enumerate(leaves):
        rln.set_leaf(i, l)
    root_single = rln.get_root()
    rln.set_tree(20)
    for l in leaves:
        rln.set_next_leaf(l)
    assert rln.get_root() == root_single
    rln.set_tree(20)
    rln.init_tree_with_leaves(leaves)
    assert rln.get_root() == root_single
    for i in range(NO_OF_LEAVES):
        rln.delete_leaf(i)
    assert rln.get_root() == root_empty
    assert rln.leaves_set() == NO_OF_LEAVES          # delete never lowers next_index


def test_leaf_setting_with_index_ffi(rln):
    """ffi.rs:199-262"""
    leaves = _leaves(2)
    rln.set_tree(20)
    assert rln.leaves_set() == 0
    set_index = 23
    rln.init_tree_with_leaves(leaves)
    root_init = rln.get_root()
    rln.init_tree_with_leaves(leaves[:set_index])
    rln.set_leaves_from(set_index, leaves[set_index:])
    assert rln.get_root() == root_init
    rln.set_tree(20)
    for l in leaves:
        rln.set_next_leaf(l)
    assert rln.get_root() == root_init


def test_atomic_operation_ffi(rln):
    """ffi.rs:265-295: deleting and re-setting the same last leaf is a no-op"""
    leaves = _leaves(3)
    rln.init_tree_with_leaves(leaves)
    root = rln.get_root()
    last = NO_OF_LEAVES - 1
    rln.atomic_operation(last, [leaves[-1]], [last])
    assert rln.get_root() == root
    # Mixed delete + set with delete indices BEFORE `start`: the reference writes its merged buffer
    # [default x (start - min_index) | new leaves] AT `start`, not at min_index (SURVEY Appendix C.1,
    # full_merkle_tree.rs:251-268 / pm_tree_adapter.rs:460-473); the old leaves are only flagged empty.
    rln.atomic_operation(last + 1, [7, 8], [last - 1, last])
    assert [rln.get_leaf(last + k) for k in (1, 2, 3, 4)] == [0, 0, 7, 8]
    assert rln.get_leaf(last) == leaves[-1] and rln.get_leaf(last - 1) == leaves[-2]
    assert rln.leaves_set() == last + 5


def test_set_leaves_bad_index_ffi(rln):
    """ffi.rs:298-322: a range that does not fit leaves the tree untouched"""
    from zerokit_amd.public import RLNError
    rln.set_tree(20)
    root_empty = rln.get_root()
    with pytest.raises(RLNError, match="too many leaves"):
        rln.set_leaves_from((1 << 20) - 5, _leaves(4))
    assert rln.get_root() == root_empty


def test_rln_out_of_bounds_ffi(rln):
    """ffi.rs:1070-1173"""
    from zerokit_amd.public import RLNError
    rln.set_tree(20)
    cap = 1 << 20
    for idx in (cap + 10, cap):
        with pytest.raises(RLNError):
            rln.set_leaf(idx, 123)
        with pytest.raises(RLNError):
            rln.get_merkle_proof(idx)
        with pytest.raises(RLNError):
            rln.get_leaf(idx)
        with pytest.raises(RLNError):
            rln.delete_leaf(idx)
    with pytest.raises(RLNError):
        rln.set_leaves_from(cap + 10, [1, 2])
    with pytest.raises(RLNError):
        rln.atomic_operation(cap + 10, [1], [cap + 10])
    with pytest.raises(RLNError):
        rln.atomic_operation(0, [1], [cap + 10])
    with pytest.raises(RLNError):
        rln.atomic_operation(0, [1, 2], [1])             # delete index after `start`
    with pytest.raises(RLNError):
        rln.atomic_operation((1 << 64) - 1, [1], [0])     # start + len overflows
    rln.set_tree(4)
    with pytest.raises(RLNError):
        rln.init_tree_with_leaves(list(range(1, 18)))     # 17 leaves into 16 slots
    for _ in range(16):
        rln.set_next_leaf(3)
    with pytest.raises(RLNError):
        rln.set_next_leaf(3)                              # tree full
    rln.set_tree(20)


def test_get_leaf_and_metadata_ffi(rln):
    """ffi.rs:763-860"""
    rln.set_tree(20)
    idx = 0xABCDE
    rln.set_leaf(idx, 424242)
    assert rln.get_leaf(idx) == 424242
    assert rln.get_metadata() == b""
    rln.set_metadata(bytes(range(10)))
    assert rln.get_metadata() == bytes(range(10))


def test_rln_invalid_witness_input_ffi():
    """ffi.rs:986-1067: constructor validation errors surface as CResult.err strings"""
    from zerokit_amd.public import RLNError, RLNWitnessInput
    with pytest.raises(RLNError, match="cannot be zero"):
        RLNWitnessInput(1, 0, 0, [0] * 20, [0] * 20, 1, 1)
    with pytest.raises(RLNError, match="length mismatch"):
        RLNWitnessInput(1, 5, 0, [0] * 20, [0] * 19, 1, 1)
    with pytest.raises(RLNError, match="not within user_message_limit"):
        RLNWitnessInput(1, 5, 5, [0] * 20, [0] * 20, 1, 1)


def test_depth10_rln_object_with_params():
    """ffi_rln_new_with_params (ffi.rs:425-470 pattern) with the shipped depth-10 resources"""
    from zerokit_amd import hashers
    from zerokit_amd.batch import resource_paths
    from zerokit_amd.public import RLN, RLNError, RLNWitnessInput
    zp, gp = resource_paths(10)
    z, g = open(zp, "rb").read(), open(gp, "rb").read()
    rln = RLN.new_with_params(10, z, g)
    with pytest.raises(RLNError, match="depth"):
        RLN.new_with_params(20, z, g)                     # graph_from_raw expected-depth check
    with pytest.raises(RLNError):
        RLN.new_with_params(10, b"", g)
    secret = 99
    rc = hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 10)
    rln.set_leaf(5, rc)
    elems, bits = rln.get_merkle_proof(5)
    assert len(elems) == 10 and bits == [1, 0, 1, 0, 0, 0, 0, 0, 0, 0]
    w = RLNWitnessInput(secret, 10, 3, elems, bits, 1234, 5678)
    p = rln.generate_rln_proof(w)
    assert rln.verify_rln_proof(p, 1234)
    assert p.values.root == rln.get_root()


def test_partial_and_finish_proof_ffi():
    """rln/tests/ffi.rs:1176-1291 + :1294-1393: partial proof, finish, serialisation round trip (6 011 bytes + version byte,
    SURVEY Appendix B), and partial + finish == full for fixed (r, s) (protocol.rs:222-248)"""
    from zerokit_amd import hashers
    from zerokit_amd.public import RLN, RLNError, RLNPartialProof, RLNPartialWitnessInput, RLNWitnessInput
    rln = RLN(20)
    secret = hashers.hash_to_field_le(b"partial-member")
    rc = hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 100)
    rln.set_leaf(7, rc)
    elems, bits = rln.get_merkle_proof(7)
    pw = RLNPartialWitnessInput(secret, 100, elems, bits)
    pp = rln.generate_partial_zk_proof(pw)
    raw = pp.to_bytes_le()
    assert len(raw) == 6012 and raw[0] == 0 and raw[1:9] == (5843).to_bytes(8, "little")
    pp2 = RLNPartialProof.from_bytes_le(raw)
    assert pp2.to_bytes_le() == raw
    with pytest.raises(RLNError):
        RLNPartialProof.from_bytes_le(raw[:-1])
    for msg_id, x in ((1, 111), (2, 222)):                 # one partial proof, many messages
        w = RLNWitnessInput(secret, 100, msg_id, elems, bits, x, 999)
        p_fin = rln.finish_rln_proof_with_rs(pp2, w, 44, 77)      # pp2 came in as bytes: no cache handle, the whole graph
        p_full = rln.generate_rln_proof_with_rs(w, 44, 77)
        assert p_fin.to_bytes_le() == p_full.to_bytes_le()
        # pp carries the prover's cache handle (round 6): the cone of the witness graph only -- the same bytes
        assert rln.finish_rln_proof_with_rs(pp, w, 44, 77).to_bytes_le() == p_full.to_bytes_le()
        assert rln.verify_rln_proof(rln.finish_rln_proof(pp, w), x)
    # a partial proof outlives nothing it should not: freeing it releases (wipes) its cache entry; another object's handle
    # means nothing to a second RLN object (there the whole graph is walked: same bytes)
    other = RLN(20)
    other.set_leaf(7, rc)
    assert other.finish_rln_proof_with_rs(pp, w, 44, 77).to_bytes_le() == p_full.to_bytes_le()
    del other
    pw2 = RLNPartialWitnessInput.from_witness(w)
    assert rln.generate_partial_zk_proof(pw2).to_bytes_le() == raw
    with pytest.raises(RLNError, match="cannot be zero"):
        RLNPartialWitnessInput(secret, 0, elems, bits)
    with pytest.raises(RLNError, match="path_elements"):
        rln.generate_partial_zk_proof(RLNPartialWitnessInput(secret, 100, elems[:10], bits[:10]))


def test_c_program_links_and_proves(tmp_path):
    """A plain C program compiled against include/rln.h and linked with -lrln (the drop-in situation of
    rln/ffi_c_examples/): creates the object, registers a member, proves, serialises, verifies."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ffi_smoke")
    libdir = os.path.join(root, "zerokit_amd", "lib")
    subprocess.check_call(["gcc", "-Wall", "-std=c11", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "host", "ffi_smoke.c"), "-L", libdir, "-lrln",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C harness: OK" in out.stdout


def _multi_rln():
    from zerokit_amd.batch import resource_paths
    from zerokit_amd.public import RLN
    zp, gp = resource_paths(20, multi=True)
    return RLN.new_with_params(20, open(zp, "rb").read(), open(gp, "rb").read())


def test_multi_message_rln_proof_ffi():
    """rln/tests/public.rs:1672-1737 through the C ABI, plus the committed golden of the max_out = 4 circuit:
    same inputs and (r, s) -> the oracle's compressed proof and public signals"""
    import json
    import os
    from zerokit_amd.public import RLNError, RLNProof, RLNWitnessInput
    rln = _multi_rln()
    assert rln.max_out() == 4
    case = [c for c in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rln_other_circuits.json")))
            ["cases"] if c["multi"]][0]
    i = {k: [int(t) for t in v] for k, v in case["inputs"].items()}
    w = RLNWitnessInput.new_multi(i["identitySecret"][0], i["userMessageLimit"][0], i["messageId"], i["pathElements"],
                                  i["identityPathIndex"], i["x"][0], i["externalNullifier"][0],
                                  [bool(b) for b in i["selectorUsed"]])
    p = rln.generate_rln_proof_with_rs(w, int(case["r"]), int(case["s"]))
    raw = p.to_bytes_le()
    assert raw[0] == 1 and raw[1:129].hex() == case["proof_compressed"]
    v = p.values
    pub = [int(t) for t in case["public"]]
    assert v.ys + [v.root] + v.nullifiers + [v.x, v.external_nullifier] + [int(b) for b in v.selector_used] == pub
    assert v.ys[1] == 0 and v.ys[3] == 0 and v.nullifiers[1] == 0 and v.ys[0] != 0 and v.nullifiers[2] != 0
    assert rln.verify_with_roots(p, i["x"][0], [])
    # partial + finish on the multi-message-id circuit (its cone: 7 773 unknown nodes of 29 254): with the object's cache
    # handle and from the wire form, the full proof's bytes and public signals
    from zerokit_amd.public import RLNPartialProof, RLNPartialWitnessInput
    pp = rln.generate_partial_zk_proof(RLNPartialWitnessInput(i["identitySecret"][0], i["userMessageLimit"][0], i["pathElements"],
                                                              i["identityPathIndex"]))
    for partial in (pp, RLNPartialProof.from_bytes_le(pp.to_bytes_le())):
        f = rln.finish_rln_proof_with_rs(partial, w, int(case["r"]), int(case["s"]))
        assert f.to_bytes_le() == raw
    q = RLNProof.from_bytes_le(raw)
    assert q.to_bytes_le() == raw and RLNProof.from_bytes_be(p.to_bytes_be()).to_bytes_le() == raw
    assert rln.verify_with_roots(q, i["x"][0], [v.root])
    # a tampered output no longer verifies; a single-message witness is refused by the multi circuit and back
    bad = bytearray(raw)
    bad[129 + 1 + 96 + 8] ^= 1                      # first byte of ys[0]
    with pytest.raises(RLNError, match="Invalid proof"):
        rln.verify_with_roots(RLNProof.from_bytes_le(bytes(bad)), i["x"][0], [])
    single = RLNWitnessInput(i["identitySecret"][0], 100, 3, i["pathElements"], i["identityPathIndex"], 5, 6)
    with pytest.raises(RLNError):
        rln.generate_rln_proof(single)
    from zerokit_amd.public import RLN
    with pytest.raises(RLNError):
        RLN(20).generate_rln_proof(w)
    with pytest.raises(RLNError, match="max_out|length"):
        rln.generate_rln_proof(RLNWitnessInput.new_multi(1, 100, [0, 1], i["pathElements"], i["identityPathIndex"],
                                                         5, 6, [True, True]))


def test_multi_message_recover_id_secret_ffi():
    """rln/tests/public.rs:1739-1828"""
    import random
    from zerokit_amd import hashers
    from zerokit_amd.public import RLNError, RLNWitnessInput, recover_id_secret
    rln = _multi_rln()
    rnd = random.Random(5)
    pe, pi = [rnd.randrange(R) for _ in range(20)], [rnd.randrange(2) for _ in range(20)]
    ext = hashers.poseidon_hash_pair(hashers.hash_to_field_le(b"test-epoch"), hashers.hash_to_field_le(b"test-rln-identifier"))
    secret, other = rnd.randrange(R), rnd.randrange(R)
    mk = lambda s, x, sel: RLNWitnessInput.new_multi(s, 10, [0, 1, 2, 3], pe, pi, x, ext, sel)
    sel = [True, True, False, False]
    proofs = rln.generate_rln_proofs_batch([mk(secret, 111, sel), mk(secret, 222, sel), mk(other, 333, sel)])
    v1, v2, v3 = (p.values for p in proofs)
    assert recover_id_secret(v1, v2) == secret
    with pytest.raises(RLNError, match="No matching nullifier"):
        recover_id_secret(v1, v3)
    # single-message slashing (rln/tests/ffi.rs:640-760 pattern) on the default circuit
    from zerokit_amd.public import RLN
    r1 = RLN(20)
    ws = [RLNWitnessInput(secret, 10, 1, pe, pi, x, ext) for x in (111, 222)]
    a, b = (p.values for p in r1.generate_rln_proofs_batch(ws))
    assert recover_id_secret(a, b) == secret
    with pytest.raises(RLNError, match="No matching nullifier"):
        recover_id_secret(a, v1)                  # cross-mode pairs are not matched (slashing.rs:97-98)


def test_generate_rln_proof_with_witness_ffi():
    """ffi_generate_rln_proof_with_witness (ffi_rln.rs:875-920): the proof is built from an externally calculated
    witness (here: this backend's own witness tap, as decimal strings with one entry sent as its negative)"""
    from zerokit_amd import hashers
    from zerokit_amd.batch import BatchProver
    from zerokit_amd.public import RLN, RLNError, RLNWitnessInput
    rln = RLN(20)
    secret = 4242
    rln.set_leaf(3, hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 50))
    elems, bits = rln.get_merkle_proof(3)
    w = RLNWitnessInput(secret, 50, 7, elems, bits, 1357, 2468)
    full = rln.generate_rln_proof(w)
    bp = BatchProver(max_batch=64, window_bits=8)
    bp.prove([dict(identity_secret=secret, user_message_limit=50, message_id=7, path_elements=elems,
                   identity_path_index=bits, x=1357, external_nullifier=2468)], [(1, 2)])
    calc = bp.fetch_witness(0)
    assert calc[0] == 1 and len(calc) == 5844
    strs = [str(v) for v in calc]
    strs[10] = "-" + str(R - calc[10]) if calc[10] else "0"      # negative entries are reduced mod r
    p = rln.generate_rln_proof_with_witness(strs, w)
    assert rln.verify_rln_proof(p, 1357)
    assert p.to_bytes_le()[129:] == full.to_bytes_le()[129:]       # same proof values
    wrong = list(calc)
    wrong[100] = (wrong[100] + 1) % R
    with pytest.raises(RLNError, match="Invalid proof"):
        rln.verify_rln_proof(rln.generate_rln_proof_with_witness(wrong, w), 1357)
    with pytest.raises(RLNError, match="Failed to parse witness"):
        rln.generate_rln_proof_with_witness(["12x"] + strs[1:], w)
    with pytest.raises(RLNError, match="malformed"):
        rln.generate_rln_proof_with_witness(strs[:-1], w)


def test_seeded_keygen_ffi():
    """rln/tests/ffi_utils.rs:8-66 and rln/tests/protocol.rs:463-540"""
    from zerokit_amd import hashers
    from zerokit_amd.public import extended_keygen, extended_seeded_keygen, seeded_keygen
    s, c = seeded_keygen(bytes(range(10)))
    assert s == 0x766ce6c7e7a01bdf5b3f257616f603918c30946fa23480f2859c597817e6716
    assert c == 0xbf16d2b5c0d6f9d9d561e05bfca16a81b4b873bb063508fae360d8c74cef51f
    s, c = seeded_keygen(b"A seed phrase example")
    assert s == 0x20df38f3f00496f19fe7c6535492543b21798ed7cb91aebe4af8012db884eda3
    assert c == 0x1223a78a5d66043a7f9863e14507dc80720a5602b2a894923e5b5147d5a9c325
    t, n, s, c = extended_seeded_keygen(bytes(range(10)))
    assert t == 0x766ce6c7e7a01bdf5b3f257616f603918c30946fa23480f2859c597817e6716
    assert n == 0x1f18714c7bc83b5bca9e89d404cf6f2f585bc4c0f7ed8b53742b7e2b298f50b4
    assert s == 0x2aca62aaa7abaf3686fff2caf00f55ab9462dc12db5b5d4bcf3994e671f8e521
    assert c == 0x68b66aa0a8320d2e56842581553285393188714c48f9b17acd198b4f1734c5c
    assert extended_seeded_keygen(b"seed") == extended_seeded_keygen(b"seed") != extended_seeded_keygen(b"seed2")
    t, n, s, c = extended_keygen()
    assert s == hashers.poseidon_hash_pair(t, n) and c == hashers.poseidon_hash([s])


def test_concurrent_proving_on_one_object():
    """generate_rln_proof takes &self in the reference (public.rs:624) and may be called from several threads: four
    threads share one RLN object; every proof must verify and carry its own x"""
    import threading
    from zerokit_amd import hashers
    from zerokit_amd.public import RLN, RLNWitnessInput
    rln = RLN(20)
    secret = 271828
    rln.set_leaf(9, hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 50))
    elems, bits = rln.get_merkle_proof(9)
    results, errors = {}, []

    def work(tid):
        try:
            for k in range(3):
                x = 1000 * tid + k + 1
                p = rln.generate_rln_proof(RLNWitnessInput(secret, 50, (tid * 3 + k) % 50, elems, bits, x, 4242))
                results[(tid, k)] = (x, p)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(results) == 12
    for x, p in results.values():
        assert p.values.x == x and rln.verify_rln_proof(p, x)


def test_persistent_tree_config(tmp_path):
    """PmTreeConfig (pm_tree_adapter.rs:71-176) through ffi_rln_new's config file: a non-temporary tree with a path
    comes back after the object is dropped (leaves, next index, root, metadata); flush writes without dropping;
    set_tree replaces it by a temporary tree; the config errors carry the reference's texts.  Same flows as
    rln/tests/pm_tree.rs:58-83 (config from JSON + reopen), :110-130 (persistence incl. metadata), :133-149 (depth
    mismatch on reload), :152-170 (deleted leaf stays empty after reload), :435-468 (multiple reopen)."""
    import json
    from zerokit_amd.public import RLN
    store = tmp_path / "tree_db"
    cfg = tmp_path / "cfg.json"
    cfg.write_text(json.dumps({"profile": "small",      # (the tree is the subject: the smallest tables build fastest)
                               "path": str(store), "temporary": False, "cache_capacity": 1 << 20,
                               "flush_every_ms": 500, "mode": "HighThroughput", "use_compression": False}))
    leaves = _leaves(11)
    r = RLN(20, str(cfg))
    assert r.leaves_set() == 0
    r.set_leaves_from(0, leaves)
    r.delete_leaf(5)
    r.set_metadata(b"block 1234567")
    root, n = r.get_root(), r.leaves_set()
    r.flush()
    assert (store / "rlnamd_tree.bin").exists()
    r.set_leaf(NO_OF_LEAVES + 3, 77)          # after the flush: written when the object is dropped
    root2 = r.get_root()
    assert root2 != root and r.leaves_set() == NO_OF_LEAVES + 4
    r.close()

    r = RLN(20, str(cfg))
    assert r.get_root() == root2 and r.leaves_set() == NO_OF_LEAVES + 4
    assert r.get_metadata() == b"block 1234567"
    assert r.get_leaf(4) == leaves[4] and r.get_leaf(5) == 0 and r.get_leaf(NO_OF_LEAVES + 3) == 77
    elems, bits = r.get_merkle_proof(4)
    assert len(elems) == 20 and bits[:3] == [0, 0, 1]
    r.set_next_leaf(9)                         # continues at the stored next index
    assert r.get_leaf(NO_OF_LEAVES + 4) == 9
    r.set_tree(20)                             # PoseidonTree::default: the stored tree is left as it was flushed
    r.set_leaf(0, 1)
    r.close()
    r = RLN(20, str(cfg))
    assert r.leaves_set() == NO_OF_LEAVES + 5 and r.get_leaf(NO_OF_LEAVES + 4) == 9 and r.get_leaf(0) == leaves[0]
    r.close()

    with pytest.raises(Exception, match="Tree depth"):          # stored depth 20 != requested depth
        RLN(10, str(cfg))
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps({"temporary": False}))
    with pytest.raises(Exception, match="Configuration error: Error while creating pmtree config: missing path"):
        RLN(20, str(bad))
    bad.write_text(json.dumps({"temporary": True, "path": str(store)}))
    with pytest.raises(Exception, match="path already exists"):
        RLN(20, str(bad))
    bad.write_text("{not json")
    with pytest.raises(Exception, match="Error while reading pmtree config"):
        RLN(20, str(bad))
    bad.write_text(json.dumps({"path": str(tmp_path / "t2"), "temporary": False, "tree_depth": 16}))
    with pytest.raises(Exception, match="Tree depth"):          # config depth != requested depth
        RLN(20, str(bad))
    # a missing config file is the default (temporary) tree, as `.unwrap_or_default()` in ffi_rln.rs:45
    r = RLN(20, str(tmp_path / "does_not_exist.json"))
    assert r.leaves_set() == 0


def test_profile_key_names_an_operating_point(tmp_path):
    """{"profile": "small"} = window_bits 8 / max_batch 64; explicit keys override what the profile implies; no key = the
    defaults (G1 c = 10, G2 c = 12, 256 proofs)"""
    import json
    from zerokit_amd.public import RLN
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({"profile": "small"}))
    info = RLN(20, tree_config=str(cfgp)).prover_info()
    assert (int(info.capacity), int(info.window_bits), int(info.window_bits_g2)) == (64, 8, 8)
    cfgp.write_text(json.dumps({"profile": "small", "max_batch": 128, "window_bits": 90009}))
    info = RLN(20, tree_config=str(cfgp)).prover_info()
    assert (int(info.capacity), int(info.window_bits), int(info.window_bits_g2)) == (128, 9, 9)
    cfgp.write_text(json.dumps({"profile": "latency"}))
    info = RLN(20, tree_config=str(cfgp)).prover_info()
    assert (int(info.capacity), int(info.window_bits), int(info.window_bits_g2)) == (256, 10, 12)
    info = RLN(20).prover_info()
    assert (int(info.capacity), int(info.window_bits), int(info.window_bits_g2)) == (256, 10, 12)


def test_auto_partial_member_memo_behind_generate_rln_proof(tmp_path):
    """{"auto_partial": N} (round 6): ffi_generate_rln_proof remembers the partial proofs of up to N members.  The first proof
    of a member at a root is made from scratch and its partial proof follows behind the call; later proofs of that member
    are finishes through the cone -- byte-identical to what an object without the memo makes for the same (r, s); a tree
    change (new path) is a new key; the least recently used member is evicted when N are held; the counters say which
    path each call took; with the key absent nothing is remembered."""
    import json
    from zerokit_amd import hashers
    from zerokit_amd.public import RLN, RLNWitnessInput
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({"auto_partial": 2}))
    rln, plain = RLN(20, tree_config=str(cfgp)), RLN(20)
    assert plain.memo_stats() == dict(members=0, finishes=0, from_scratch=0, pending=0)
    secrets = [hashers.hash_to_field_le(b"memo-member-%d" % k) for k in range(3)]
    for obj in (rln, plain):
        for k, sec in enumerate(secrets):
            obj.set_leaf(10 + k, hashers.poseidon_hash_pair(hashers.poseidon_hash([sec]), 100))
    paths = [rln.get_merkle_proof(10 + k) for k in range(3)]

    def wit(k, msg, x, path=None):
        e, b = path or paths[k]
        return RLNWitnessInput(secrets[k], 100, msg, e, b, x, 4242)

    def same(k, msg, x, path=None):
        a = rln.generate_rln_proof_with_rs(wit(k, msg, x, path), 44 + msg, 77 + x)
        b = plain.generate_rln_proof_with_rs(wit(k, msg, x, path), 44 + msg, 77 + x)
        assert a.to_bytes_le() == b.to_bytes_le()
        assert rln.verify_rln_proof(a, x)
    same(0, 1, 1000)                                   # member 0: from scratch, its partial proof enqueued
    st = rln.memo_stats()
    assert (st["from_scratch"], st["finishes"], st["pending"]) == (1, 0, 1)
    same(0, 2, 1001)                                   # adopted at the start of this call: a finish
    same(0, 3, 1002)
    st = rln.memo_stats()
    assert (st["members"], st["from_scratch"], st["finishes"], st["pending"]) == (1, 1, 2, 0)
    same(1, 1, 2000)                                   # a second member
    same(1, 2, 2001)
    same(0, 4, 1003)                                   # both remembered
    assert rln.memo_stats()["members"] == 2 and rln.memo_stats()["finishes"] == 4
    same(2, 1, 3000)                                   # a third member: the least recently used one (member 1) goes
    same(2, 2, 3001)
    same(0, 5, 1004)                                   # member 0 is still there
    st = rln.memo_stats()
    assert (st["members"], st["from_scratch"], st["finishes"]) == (2, 3, 6)
    same(1, 3, 2002)                                   # member 1 again: from scratch
    assert rln.memo_stats()["from_scratch"] == 4
    # the tree moves on: member 0's path changes, the old partial proof is not used for the new root
    for obj in (rln, plain):
        obj.set_leaf(500, 12345)
    new_path = rln.get_merkle_proof(10)
    assert new_path[0] != paths[0][0]
    same(0, 6, 1005, new_path)
    assert rln.memo_stats()["from_scratch"] == 5
    same(0, 7, 1006, new_path)
    assert rln.memo_stats()["finishes"] == 7
    # random (r, s): the memo's finish verifies like any proof
    p = rln.generate_rln_proof(wit(0, 8, 1007, new_path))
    assert rln.verify_rln_proof(p, 1007) and rln.memo_stats()["finishes"] == 8


def test_concurrent_single_proof_calls_are_gathered_into_batches(tmp_path):
    """generate_rln_proof takes &self (public.rs:624).  Calls that arrive from other threads while a proof is on the device
    go out together as one batch when it returns (ffi.cpp: prove_one).  Eight threads, ten calls each, on one object:
    every proof equals, byte for byte, what an object with {"gather_calls": 0} makes alone for that witness and (r, s),
    and verifies; batches of more than one call were led; a call whose witness cannot be evaluated (x = r, not canonical:
    graph.rs:42-45 -- refused before the device sees it -- and a path of the wrong length) gets ITS error text while
    the calls gathered with it get their proofs; the V3 entry point is gathered too."""
    import json
    import threading
    from zerokit_amd import hashers
    from zerokit_amd.public import RLN, RLNError, RLNWitnessInput
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({"gather_calls": 0}))
    rln, alone = RLN(20), RLN(20, tree_config=str(cfgp))
    assert alone.gather_stats()["cap"] == 0 and rln.gather_stats()["cap"] > 1
    secrets = [hashers.hash_to_field_le(b"gathered-member-%d" % k) for k in range(4)]
    for obj in (rln, alone):
        for k, sec in enumerate(secrets):
            obj.set_leaf(30 + k, hashers.poseidon_hash_pair(hashers.poseidon_hash([sec]), 100))
    paths = [rln.get_merkle_proof(30 + k) for k in range(4)]
    made, errors, refused = {}, [], {}

    def work(tid):
        try:
            for j in range(10):
                k = (tid + j) % 4
                msg, x = (tid * 10 + j) % 100, 9000 + 100 * tid + j
                if tid == 3 and j % 5 == 2:      # a witness the circuit cannot take, in the middle of the others
                    try:
                        rln.generate_rln_proof_with_rs(RLNWitnessInput(secrets[k], 100, msg, paths[k][0][:-1], paths[k][1][:-1], x, 4242), 1, 2)
                        refused[(tid, j)] = None
                    except RLNError as e:
                        refused[(tid, j)] = str(e)
                    continue
                w = RLNWitnessInput(secrets[k], 100, msg, paths[k][0], paths[k][1], x, 4242)
                made[(tid, j)] = (k, msg, x, rln.generate_rln_proof_with_rs(w, 9 + x, 3 + msg).to_bytes_le())
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(made) + len(refused) == 80 and len(refused) == 2, errors
    assert all(m and "path_elements has length 19" in m for m in refused.values()), refused
    for k, msg, x, got in made.values():
        w = RLNWitnessInput(secrets[k], 100, msg, paths[k][0], paths[k][1], x, 4242)
        ref = alone.generate_rln_proof_with_rs(w, 9 + x, 3 + msg)
        assert got == ref.to_bytes_le() and alone.verify_rln_proof(ref, x)
    st, sa = rln.gather_stats(), alone.gather_stats()
    assert st["calls"] == 80 and st["largest"] > 1 and st["batches"] < 80, st
    assert sa["batches"] == 0, sa
    # random (r, s) through the same door: each call its own blinding, every proof verifies
    outs = {}

    def rnd(tid):
        w = RLNWitnessInput(secrets[tid % 4], 100, 10 + tid, paths[tid % 4][0], paths[tid % 4][1], 400 + tid, 4242)
        outs[tid] = [rln.generate_rln_proof(w) for _ in range(3)]
    threads = [threading.Thread(target=rnd, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert len(outs) == 4
    for tid, ps in outs.items():
        assert all(rln.verify_rln_proof(p, 400 + tid) for p in ps)
        assert len({p.to_bytes_le() for p in ps}) == 3


def test_concurrent_finishes_of_partial_proofs_are_gathered_into_batches(tmp_path):
    """The reference's partial proofs exist to be finished quickly per message (rln/README.md:360-375); finishes from
    several threads on one object are gathered like full proofs (ffi.cpp: finish_one).  Two members' partial proofs (one
    with its cache handle, one deserialised from bytes: no handle), eight threads finishing for both: every proof equals,
    byte for byte, the full proof an object that gathers nothing makes for the same witness and (r, s), and verifies;
    batches of more than one finish were led; a finish whose witness does not fit the circuit gets its own error."""
    import json
    import threading
    from zerokit_amd import hashers
    from zerokit_amd.public import RLN, RLNError, RLNPartialProof, RLNPartialWitnessInput, RLNWitnessInput
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({"gather_calls": 0}))
    rln, alone = RLN(20), RLN(20, tree_config=str(cfgp))
    secrets = [hashers.hash_to_field_le(b"finish-member-%d" % k) for k in range(2)]
    for obj in (rln, alone):
        for k, sec in enumerate(secrets):
            obj.set_leaf(40 + k, hashers.poseidon_hash_pair(hashers.poseidon_hash([sec]), 100))
    paths = [rln.get_merkle_proof(40 + k) for k in range(2)]
    partials = [rln.generate_partial_zk_proof(RLNPartialWitnessInput(secrets[k], 100, paths[k][0], paths[k][1])) for k in range(2)]
    partials[1] = RLNPartialProof.from_bytes_le(partials[1].to_bytes_le())      # no cache handle: finished over the whole graph
    made, errors, refused = {}, [], []

    def work(tid):
        try:
            for j in range(10):
                k = (tid + j) % 2
                msg, x = (tid * 10 + j) % 100, 3000 + 100 * tid + j
                if tid == 5 and j == 4:
                    try:
                        rln.finish_rln_proof_with_rs(partials[k], RLNWitnessInput(secrets[k], 100, msg, paths[k][0][:-1], paths[k][1][:-1], x, 4242), 1, 2)
                        refused.append(None)
                    except RLNError as e:
                        refused.append(str(e))
                    continue
                w = RLNWitnessInput(secrets[k], 100, msg, paths[k][0], paths[k][1], x, 4242)
                made[(tid, j)] = (k, msg, x, rln.finish_rln_proof_with_rs(partials[k], w, 9 + x, 3 + msg).to_bytes_le())
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(made) == 79, errors
    assert refused and refused[0] and "path_elements has length 19" in refused[0], refused
    for k, msg, x, got in made.values():
        w = RLNWitnessInput(secrets[k], 100, msg, paths[k][0], paths[k][1], x, 4242)
        ref = alone.generate_rln_proof_with_rs(w, 9 + x, 3 + msg)
        assert got == ref.to_bytes_le() and alone.verify_rln_proof(ref, x)
    st = rln.gather_stats()
    assert st["finish_calls"] == 80 and st["finish_batches"] < 80, st


def test_finish_rln_proofs_batch_ext():
    """EXT ffi_finish_rln_proofs_batch: one member's partial proof finished for many messages in ONE call (what partial
    proofs are for, rln/README.md:360-375).  300 finishes over two members' partial proofs (more than the object's 256-proof
    workspace: two chunks) equal, byte for byte, the full proofs ffi_generate_rln_proofs_batch makes for the same witnesses
    and (r, s); random blinding verifies; a witness that does not fit fails the call with the reference's text."""
    from zerokit_amd import hashers
    from zerokit_amd.public import RLN, RLNError, RLNPartialWitnessInput, RLNWitnessInput
    rln = RLN(20)
    secrets = [hashers.hash_to_field_le(b"batch-finish-%d" % k) for k in range(2)]
    for k, sec in enumerate(secrets):
        rln.set_leaf(50 + k, hashers.poseidon_hash_pair(hashers.poseidon_hash([sec]), 1000))
    paths = [rln.get_merkle_proof(50 + k) for k in range(2)]
    parts = [rln.generate_partial_zk_proof(RLNPartialWitnessInput(secrets[k], 1000, paths[k][0], paths[k][1])) for k in range(2)]
    n = 300
    ws = [RLNWitnessInput(secrets[i % 2], 1000, i, paths[i % 2][0], paths[i % 2][1], 7000 + i, 4242) for i in range(n)]
    rs = [(11 + i, 5 + 3 * i) for i in range(n)]
    fin = rln.finish_rln_proofs_batch([parts[i % 2] for i in range(n)], ws, rs)
    full = rln.generate_rln_proofs_batch(ws, rs)
    assert len(fin) == n
    for i in range(n):
        assert fin[i].to_bytes_le() == full[i].to_bytes_le(), i
    assert all(rln.verify_rln_proof(fin[i], 7000 + i) for i in (0, 1, 255, 256, 299))
    rnd = rln.finish_rln_proofs_batch([parts[0]] * 5, [ws[2 * i] for i in range(5)])
    assert all(rln.verify_rln_proof(rnd[i], 7000 + 2 * i) for i in range(5))
    assert rnd[0].to_bytes_le() != fin[0].to_bytes_le()
    bad = RLNWitnessInput(secrets[0], 1000, 1, paths[0][0][:-1], paths[0][1][:-1], 1, 4242)
    with pytest.raises(RLNError, match="path_elements has length 19"):
        rln.finish_rln_proofs_batch([parts[0]] * 3, [ws[0], bad, ws[2]], rs[:3])


def test_concurrent_proving_with_the_member_memo(tmp_path):
    """generate_rln_proof takes &self (public.rs:624): four threads prove for three members on ONE object whose memo holds
    two ({"auto_partial": 2}) -- adoptions of pending partial proofs, finishes through the cone, evictions and proofs from
    scratch interleave as the scheduler likes.  Every proof equals, byte for byte, the proof an object without the memo
    makes for the same witness and (r, s), and verifies; the counters add up to the calls; afterwards the memo still turns
    a member's repeated proofs into finishes."""
    import json
    import threading
    from zerokit_amd import hashers
    from zerokit_amd.public import RLN, RLNWitnessInput
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({"auto_partial": 2}))
    rln, plain = RLN(20, tree_config=str(cfgp)), RLN(20)
    secrets = [hashers.hash_to_field_le(b"memo-thread-member-%d" % k) for k in range(3)]
    for obj in (rln, plain):
        for k, sec in enumerate(secrets):
            obj.set_leaf(20 + k, hashers.poseidon_hash_pair(hashers.poseidon_hash([sec]), 100))
    paths = [rln.get_merkle_proof(20 + k) for k in range(3)]
    made, errors = {}, []

    def work(tid):
        try:
            for j in range(6):
                k = (tid + j) % 3
                msg, x = (tid * 6 + j) % 100, 5000 + 100 * tid + j
                w = RLNWitnessInput(secrets[k], 100, msg, paths[k][0], paths[k][1], x, 4242)
                made[(tid, j)] = (k, msg, x, rln.generate_rln_proof_with_rs(w, 9 + x, 3 + msg).to_bytes_le())
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(made) == 24, errors
    for k, msg, x, got in made.values():
        w = RLNWitnessInput(secrets[k], 100, msg, paths[k][0], paths[k][1], x, 4242)
        ref = plain.generate_rln_proof_with_rs(w, 9 + x, 3 + msg)
        assert got == ref.to_bytes_le() and plain.verify_rln_proof(ref, x)
    st = rln.memo_stats()
    # (how many of the 24 were finishes is the scheduler's choice: three members taking turns through a memo of two can
    # miss every time)
    assert st["finishes"] + st["from_scratch"] == 24 and st["members"] <= 2, st
    for j in range(3):          # and the memo still works after all that: the same member three times in a row
        w = RLNWitnessInput(secrets[0], 100, 90 + j, paths[0][0], paths[0][1], 7000 + j, 4242)
        a, b = rln.generate_rln_proof_with_rs(w, 5 + j, 6 + j), plain.generate_rln_proof_with_rs(w, 5 + j, 6 + j)
        assert a.to_bytes_le() == b.to_bytes_le()
    assert rln.memo_stats()["finishes"] >= st["finishes"] + 2, rln.memo_stats()


def test_default_object_falls_back_to_the_small_point_when_the_device_is_nearly_full(monkeypatch, tmp_path):
    """every ffi_rln_new* without a sizing key allocates ~23 GiB.  With less than 26 GiB free the object is
    built at the "small" point instead (same proofs), below 10 GiB the error names the `profile` key; an explicit
    choice is never second-guessed.  (The free-memory reading is overridden by a test hook.)"""
    import json
    from zerokit_amd.public import RLN
    monkeypatch.setenv("RLNAMD_ASSUME_FREE_GIB", "20")
    info = RLN(20).prover_info()
    assert (int(info.capacity), int(info.window_bits), int(info.window_bits_g2)) == (64, 8, 8)
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({"profile": "latency"}))
    info = RLN(20, tree_config=str(cfgp)).prover_info()
    assert (int(info.capacity), int(info.window_bits), int(info.window_bits_g2)) == (256, 10, 12)
    monkeypatch.setenv("RLNAMD_ASSUME_FREE_GIB", "5")
    with pytest.raises(Exception, match="profile"):
        RLN(20)


def test_config_path_sizes_the_prover_and_batch_streams_past_max_batch(tmp_path):
    """the `window_bits` / `max_batch` keys of the config_path JSON (beside the PmTreeConfig keys of
    pm_tree_adapter.rs:139-174, which stay honoured) size the prover behind ffi_rln_new, and
    ffi_generate_rln_proofs_batch streams n > max_batch proofs through the workspace slots: 300 proofs with
    max_batch 64 (4 full chunks + 44, more chunks than slots) equal the extension API's bytes and the golden (44, 77)
    proof in position 0; the multi message-id circuit takes the same path"""
    import json
    import os
    from zerokit_amd import workload
    from zerokit_amd.batch import BatchProver, resource_paths
    from zerokit_amd.public import RLN, RLNWitnessInput
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({"window_bits": 10, "max_batch": 64, "temporary": True, "cache_capacity": 1073741824}))
    rln = RLN(20, tree_config=str(cfgp))
    info = rln.prover_info()
    assert int(info.capacity) == 64 and int(info.window_bits) == 10
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]
    gold = next(c for c in cases if c["name"] == "survey_appendix_d")
    gw = gold["witness"]
    n = 300
    ws, rs = workload.config2_range(40000, n)
    ws[0] = dict(identity_secret=int(gw["identity_secret"]), user_message_limit=int(gw["user_message_limit"]),
                 message_id=int(gw["message_id"]), path_elements=[int(t) for t in gw["path_elements"]],
                 identity_path_index=[int(t) for t in gw["identity_path_index"]], x=int(gw["x"]),
                 external_nullifier=int(gw["external_nullifier"]))
    rs[0] = (int(gold["r"]), int(gold["s"]))
    wi = [RLNWitnessInput(w["identity_secret"], w["user_message_limit"], w["message_id"], w["path_elements"],
                          w["identity_path_index"], w["x"], w["external_nullifier"]) for w in ws]
    proofs = rln.generate_rln_proofs_batch(wi, rs)
    p = BatchProver(max_batch=128)
    ref = p.prove_stream(ws, rs)
    p.close()
    got = [pr.to_bytes_le() for pr in proofs]
    assert all(r["proof"] in g for r, g in zip(ref, got))
    assert bytes.fromhex(gold["proof_compressed"]) in got[0]
    for i in (0, 63, 64, 299):
        v = proofs[i].values
        assert v.y == ref[i]["values"]["y"] and v.nullifier == ref[i]["values"]["nullifier"]
        assert rln.verify_with_roots(proofs[i], ws[i]["x"], [])   # no roots given: the zk proof and x only
    del proofs, rln
    # multi message-id circuit through the same streaming path
    zp, gp = resource_paths(20, multi=True)
    z, g = open(zp, "rb").read(), open(gp, "rb").read()
    cfgp.write_text(json.dumps({"max_batch": 64}))
    m = RLN.new_with_params(20, z, g, tree_config=str(cfgp))
    assert int(m.prover_info().capacity) == 64
    mws = []
    for i in range(150):
        w = ws[i + 1]
        mws.append(RLNWitnessInput.new_multi(w["identity_secret"], 100, [10 + i % 80, 3, 5, 7], w["path_elements"],
                                             w["identity_path_index"], w["x"], w["external_nullifier"],
                                             [True, i % 2 == 0, False, True]))
    a = m.generate_rln_proofs_batch(mws, rs[:150])           # 64 + 64 + 22: streamed
    b = [m.generate_rln_proofs_batch([mws[i]], [rs[i]])[0] for i in (0, 64, 149)]   # one at a time: resident path
    for k, i in enumerate((0, 64, 149)):
        assert a[i].to_bytes_le() == b[k].to_bytes_le()
    # ... and both judged by oracle/c on the same circuit (the library was compared with itself here): proof
    # bytes and the values of witness.rs:777-802 for all 150
    from oracle.c import binding as ob
    o = ob.Circuit(20, multi=True)
    named = []
    for i in range(150):
        w = ws[i + 1]
        named.append({"identitySecret": [w["identity_secret"]], "userMessageLimit": [100],
                      "messageId": [10 + i % 80, 3, 5, 7], "selectorUsed": [1, int(i % 2 == 0), 0, 1],
                      "pathElements": w["path_elements"], "identityPathIndex": w["identity_path_index"], "x": [w["x"]],
                      "externalNullifier": [w["external_nullifier"]]})
    rsb = b"".join(r.to_bytes(32, "little") + s_.to_bytes(32, "little") for r, s_ in rs[:150])
    _, oproofs, opub = o.prove_many_packed(b"".join(o.pack_named(w) for w in named), rsb)
    for i in range(150):
        assert oproofs[i] in a[i].to_bytes_le(), i
        v = a[i].values
        assert list(v.ys) + [v.root] + list(v.nullifiers) + [v.x, v.external_nullifier] + \
            [int(bool(t)) for t in v.selector_used] == opub[i], i
    del a, b, m
    # the throughput shape behind the FFI on this circuit: one 200-proof batch in a 256-proof workspace (lanes = proofs,
    # pair chunks), witnesses of the seeded multi-circuit workload
    cfgp.write_text(json.dumps({"max_batch": 256}))
    m = RLN.new_with_params(20, z, g, tree_config=str(cfgp))
    named, rs2 = workload.circuit_range(7000, 200, 20, True)
    mws = [RLNWitnessInput.new_multi(w["identitySecret"][0], w["userMessageLimit"][0], w["messageId"], w["pathElements"],
                                     w["identityPathIndex"], w["x"][0], w["externalNullifier"][0],
                                     [bool(t) for t in w["selectorUsed"]]) for w in named]
    a = m.generate_rln_proofs_batch(mws, rs2)
    rsb = b"".join(r.to_bytes(32, "little") + s_.to_bytes(32, "little") for r, s_ in rs2)
    _, oproofs, opub = o.prove_many_packed(b"".join(o.pack_named(w) for w in named), rsb)
    for i in range(200):
        assert oproofs[i] in a[i].to_bytes_le(), i
        v = a[i].values
        assert list(v.ys) + [v.root] + list(v.nullifiers) + [v.x, v.external_nullifier] + \
            [int(bool(t)) for t in v.selector_used] == opub[i], i
    assert m.verify_with_roots(a[0], named[0]["x"][0], []) and m.verify_with_roots(a[199], named[199]["x"][0], [])


def test_config_devices_puts_a_pool_behind_the_ffi_object(tmp_path):
    """`"devices": [0, 0]` in the config_path JSON: two prover replicas (here sharing device 0) behind one FFI_RLN;
    ffi_generate_rln_proofs_batch shards 300 proofs over them by index (128 + 172, chunks of 64) and returns the bytes
    a single prover gives; single proofs, the tree and verification keep working on the object"""
    import json
    from zerokit_amd import hashers, workload
    from zerokit_amd.batch import BatchProver
    from zerokit_amd.public import RLN, RLNWitnessInput
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({"devices": [0, 0], "max_batch": 64}))
    rln = RLN(20, tree_config=str(cfgp))
    assert int(rln.prover_info().capacity) == 64
    n = 300
    ws, rs = workload.config2_range(50000, n)
    wi = [RLNWitnessInput(w["identity_secret"], w["user_message_limit"], w["message_id"], w["path_elements"],
                          w["identity_path_index"], w["x"], w["external_nullifier"]) for w in ws]
    proofs = rln.generate_rln_proofs_batch(wi, rs)
    p = BatchProver(max_batch=128)
    ref = p.prove_stream(ws, rs)
    p.close()
    got = [pr.to_bytes_le() for pr in proofs]
    assert all(r["proof"] in g for r, g in zip(ref, got))
    for i in (0, 127, 128, 299):
        assert proofs[i].values.y == ref[i]["values"]["y"]
        assert rln.verify_with_roots(proofs[i], ws[i]["x"], [])
    # the ordinary calls on the same object
    secret = 424242
    rln.set_leaf(5, hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 100))
    elems, bits = rln.get_merkle_proof(5)
    one = rln.generate_rln_proof(RLNWitnessInput(secret, 100, 7, elems, bits, 99, 1234))
    assert rln.verify_rln_proof(one, 99)
    # the pool's assignment and failover switches as keys of the same JSON: the same bytes out
    del rln
    cfgp.write_text(json.dumps({"devices": [0, 0, 0], "max_batch": 64, "dynamic_shards": True, "failover": 1}))
    rln = RLN(20, tree_config=str(cfgp))
    got = [pr.to_bytes_le() for pr in rln.generate_rln_proofs_batch(wi, rs)]
    assert all(r["proof"] in g for r, g in zip(ref, got))
    del rln
    cfgp.write_text(json.dumps({"devices": [0, 0], "failover": 65}))
    with pytest.raises(Exception, match="failover"):
        RLN(20, tree_config=str(cfgp))
    # a device that does not exist is a configuration error, not a crash
    cfgp.write_text(json.dumps({"devices": [0, 57]}))
    with pytest.raises(Exception, match="does not exist"):
        RLN(20, tree_config=str(cfgp))


@pytest.mark.parametrize("host_max", ["0", None, "8", "4096"])
def test_deferred_tree_updates_with_reads_at_random_points_vs_oracle(host_max, monkeypatch):
    """(host_max: how many dirty leaves a pass may hold and still run its dependent chain on a host core --
    MerkleTreeDev::set_few -- forced to never / the default 11 / 8 / always-below-4096.)
    ffi_set_leaf / ffi_set_next_leaf / ffi_delete_leaf only record the write; the first reader hashes the union of the
    dirty paths in one pass (TreeAny::set_leaf).  A random stream of the mutation calls of rln/src/ffi/ffi_tree.rs with
    reads (root, leaf, Merkle proof, leaves_set, a proof generated over the tree) at random points gives what
    FullMerkleTree gives call by call (utils/src/merkle_tree/full_merkle_tree.rs:197-223,271-285,336-399), on the dense
    depth-20 tree; rewrites of one index between two reads keep the last value; a burst of 3 000 writes between two reads
    (every list kernel of the pass) and a range write over pending single writes included."""
    from oracle.c import binding as ob
    from zerokit_amd.public import RLN
    if host_max is None:
        monkeypatch.delenv("RLNAMD_TREE_HOST_MAX", raising=False)
    else:
        monkeypatch.setenv("RLNAMD_TREE_HOST_MAX", host_max)
    rnd = random.Random(2024)
    rln, o = RLN(20), ob.Tree(20)
    nxt, written = 0, {}

    def check_reads():
        assert rln.get_root() == o.root()
        i = rnd.choice(list(written) or [0])
        assert rln.get_leaf(i) == written.get(i, 0)
        elems, bits = rln.get_merkle_proof(i)
        oe, obits = o.proof(i)
        assert elems == oe and list(bits) == obits
        assert rln.leaves_set() == nxt

    assert rln.get_root() == o.root()
    for step in range(400):
        op = rnd.random()
        if op < 0.45:
            i, v = rnd.randrange(1 << 20) if rnd.random() < 0.7 else rnd.choice(list(written) or [3]), rnd.randrange(1, R)
            rln.set_leaf(i, v)
            o.set(i, v)
            written[i] = v
            nxt = max(nxt, i + 1)
        elif op < 0.6:
            v = rnd.randrange(1, R)
            if nxt < (1 << 20):
                rln.set_next_leaf(v)
                o.set(nxt, v)
                written[nxt] = v
                nxt += 1
        elif op < 0.7 and written:
            i = rnd.choice(list(written))
            rln.delete_leaf(i)
            o.set(i, 0)
            written[i] = 0
        elif op < 0.75:
            start, vals = rnd.randrange(1 << 19), [rnd.randrange(1, R) for _ in range(rnd.choice([2, 9, 64, 65, 300]))]
            rln.set_leaves_from(start, vals)
            o.set_range(start, vals)
            for k, v in enumerate(vals):
                written[start + k] = v
            nxt = max(nxt, start + len(vals))
        else:
            check_reads()
    check_reads()
    # a burst: 3 000 scattered writes, two of them rewritten, then one read
    burst = [(rnd.randrange(1 << 20), rnd.randrange(1, R)) for _ in range(3000)]
    burst += [(burst[5][0], 777), (burst[6][0], 0)]
    for i, v in burst:
        rln.set_leaf(i, v)
        o.set(i, v)
        written[i] = v
        nxt = max(nxt, i + 1)
    check_reads()
    # the witness of a proof is read from the tree after pending writes: the proof verifies against the oracle's root
    from zerokit_amd import hashers
    from zerokit_amd.public import RLNWitnessInput
    secret = 987654321
    rc = hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 100)
    rln.set_leaf(4242, rc)
    o.set(4242, rc)
    elems, bits = rln.get_merkle_proof(4242)
    pr = rln.generate_rln_proof(RLNWitnessInput(secret, 100, 3, elems, bits, 11, 22))
    assert pr.values.root == o.root() and rln.verify_rln_proof(pr, 11)
    o.close()


def test_current_device_is_unchanged_across_object_lifetime_and_devices_list_must_start_with_it(tmp_path):
    """the calling thread's device is the same after ffi_rln_new / ffi_rln_free of an object with a device pool behind it
    (~rlnamd_pool switches to each replica's device to free it, then back); a one-entry devices list is honoured; a list
    that does not start with the current device is a configuration error"""
    import ctypes as C
    import json
    from zerokit_amd import lib
    from zerokit_amd._native import check
    from zerokit_amd.public import RLN

    def cur():
        d = C.c_int(-1)
        check(lib().rlnamd_get_device(C.byref(d)))
        return d.value

    before = cur()
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({"devices": [before, before], "max_batch": 64}))
    r = RLN(20, tree_config=str(cfgp))
    assert cur() == before
    r.get_root()
    del r
    import gc
    gc.collect()
    assert cur() == before
    cfgp.write_text(json.dumps({"devices": [before]}))
    r = RLN(20, tree_config=str(cfgp))
    assert r.get_root() == RLN(20).get_root()
    del r
    cfgp.write_text(json.dumps({"devices": [before + 1, before]}))
    with pytest.raises(Exception, match="current device"):
        RLN(20, tree_config=str(cfgp))
    cfgp.write_text(json.dumps({"devices": []}))
    with pytest.raises(Exception, match="empty list"):
        RLN(20, tree_config=str(cfgp))
    assert cur() == before


def test_deep_tree_depth_40_sparse():
    """depths the dense HBM tree cannot hold (31 .. 63; the reference's OptimalMerkleTree allows < 64,
    utils/src/merkle_tree/optimal_merkle_tree.rs:15-41) take the sparse host-indexed tree with device hashing: a depth-40
    tree with leaves set far apart gives the oracle's root, leaves and paths; depth 64 stays an error"""
    from oracle.pyref.rln import SparseMerkleTree
    from zerokit_amd.public import RLN
    d = 40
    rln = RLN(20)
    rln.set_tree(d)
    assert rln.tree_depth() == d
    o = SparseMerkleTree(d)
    assert rln.get_root() == o.root()
    rnd = random.Random(11)
    idx = [0, 1, 2, (1 << 31) + 5, (1 << 39) + 123456789, (1 << 40) - 1, 77, (1 << 31) + 4]
    for i in idx:
        v = rnd.randrange(1, R)
        rln.set_leaf(i, v)
        o.set(i, v)
        assert rln.get_root() == o.root()
    rln.set_leaves_from(1000, [5, 6, 7, 8, 9])          # a range: one device batch per level
    for k, v in enumerate([5, 6, 7, 8, 9]):
        o.set(1000 + k, v)
    assert rln.get_root() == o.root()
    for i in (0, (1 << 39) + 123456789, (1 << 40) - 1, 1002, 12345):
        elems, bits = rln.get_merkle_proof(i)
        oe, ob = o.proof(i)
        assert elems == oe and list(bits) == ob
        assert rln.get_leaf(i) == o.get(i)
    rln.delete_leaf(77)
    o.set(77, 0)
    assert rln.get_root() == o.root()
    with pytest.raises(Exception, match="out of bounds|too many"):
        rln.set_leaf(1 << 40, 1)
    with pytest.raises(Exception, match="must be < 64"):
        rln.set_tree(64)
    rln.set_tree(20)                                     # back to the dense tree
    assert rln.get_root() == SparseMerkleTree(20).root()


def test_reference_tree_tests_and_kat_on_the_sparse_tree():
    """the FFI tree tests of this file and of the V3 file (the reference's rln/tests/ffi.rs tree cases replayed) with the
    sparse tree forced at every depth (RLNAMD_TREE_SPARSE_ABOVE=0): same roots, paths and errors as the dense HBM tree --
    including the deferred-update stream with reads at random points (SparseTree::set_many behind the same queue)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RLNAMD_TREE_SPARSE_ABOVE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", os.path.join(root, "tests", "test_gpu_ffi.py"),
                        os.path.join(root, "tests", "test_gpu_ffi_v3.py"), "-k",
                        "(merkle_operations or leaf_setting or atomic_operation or bad_index or out_of_bounds or "
                        "get_leaf_and_metadata or persistent_tree or stateful_tree or c_program_links or deferred_tree_updates) "
                        "and not sparse"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout
