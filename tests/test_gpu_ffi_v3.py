"""The V3 surface (ffi_rln_v3_*, what the reference's shipped C / Nim callers use) on the GPU: rln/tests/proof.rs replayed
through the C ABI, the golden proofs through the V3 wire formats, and -- when oracle/_ref/examples holds them -- the
reference's own C example programs (built unmodified by oracle/ref_examples.mk) executed against this librln."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _res(multi=False):
    from zerokit_amd.batch import resource_paths
    zp, gp = resource_paths(20, multi)
    return open(zp, "rb").read(), open(gp, "rb").read()


@pytest.fixture(scope="module")
def single():
    from zerokit_amd.public_v3 import RLNV3
    return RLNV3.stateless()


@pytest.fixture(scope="module")
def multi():
    from zerokit_amd.public_v3 import RLNV3
    return RLNV3.stateless(*_res(True))


def _sw(secret, message_id=1, x=1, ext=1, depth=20):
    from zerokit_amd.public_v3 import RLNWitnessInputV3
    return RLNWitnessInputV3.new_single(secret, 10, message_id, [0] * depth, [0] * depth, x, ext)


def _mw(secret, ids=(1, 2, 3, 4), sel=(True,) * 4, x=42, ext=100):
    from zerokit_amd.public_v3 import RLNWitnessInputV3
    return RLNWitnessInputV3.new_multi(secret, 10, list(ids), [0] * 20, [0] * 20, x, ext, list(sel))


def test_v3_generate_verify_and_errors(single, multi):
    """rln/tests/proof.rs:68-158, :271-298"""
    from zerokit_amd._native import RLNError
    p = single.generate_proof(_sw(1234, x=77))
    assert single.verify(p, 77)
    assert single.verify(p, 78) is False                         # signal mismatch: plain false, no error text
    p2 = single.generate_proof(_sw(5678, x=77))
    from zerokit_amd.public_v3 import RLNProofV3
    swapped = RLNProofV3.from_bytes_le(p.to_bytes_le()[:128] + p2.to_bytes_le()[128:])
    assert single.verify(swapped, 77) is False                   # proof 1 with the values of proof 2 (:82-101)
    with pytest.raises(RLNError, match="Field `path_elements` has length 21, but circuit tree_depth is 20"):
        single.generate_proof(_sw(1, depth=21))
    for sel in ((True,) * 4, (True, False, True, False)):
        q = multi.generate_proof(_mw(4321, sel=sel))
        assert multi.verify(q, 42)
        v = q.values
        assert v.selector_used == list(sel) and [y != 0 for y in v.ys] == list(sel)
    with pytest.raises(RLNError, match="Field `message_ids` has length 2, but circuit max_out is 4"):
        multi.generate_proof(_mw(1, ids=(1, 2), sel=(True, True)))
    with pytest.raises(RLNError, match="Field `message_ids` has length 1, but circuit max_out is 4"):
        multi.generate_proof(_sw(1))
    with pytest.raises(RLNError, match="Field `message_ids` has length 4, but circuit max_out is 1"):
        single.generate_proof(_mw(1))
    # stateless: tree calls are refused, root reads as zero (ffi_rln_v3.rs:19, :1425, :1531)
    with pytest.raises(RLNError, match="tree op unsupported on stateless RLN"):
        single.set_leaf(0, 1)
    with pytest.raises(RLNError, match="tree op unsupported on stateless RLN"):
        single.get_merkle_proof(0)
    assert single.get_root() == 0 and single.leaves_set() == 0


def test_v3_calls_from_several_threads_are_gathered(single, multi):
    """ffi_rln_v3_generate_proof goes through the same door as the V1 entry point (ffi.cpp: prove_one): six threads on one
    RLNV3 object, single and multi message-id, every proof verifies with its own x; a witness of the wrong depth among them
    gets the V3 error text while the calls gathered with it get their proofs; the multi-message-id object's batches keep the
    public signals of every call apart."""
    import threading
    from zerokit_amd._native import RLNError
    for obj, mk in ((single, _sw), (multi, _mw)):
        outs, errors, refused = {}, [], []

        def work(tid):
            try:
                for j in range(5):
                    x = 100 * tid + j + 1
                    if obj is single and tid == 2 and j == 2:
                        try:
                            obj.generate_proof(_sw(7, depth=21, x=x))
                            refused.append(None)
                        except RLNError as e:
                            refused.append(str(e))
                        continue
                    outs[(tid, j)] = (x, obj.generate_proof(mk(1000 + tid, x=x)))
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=work, args=(t,)) for t in range(6)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert len(outs) == (29 if obj is single else 30)
        for x, p in outs.values():
            assert obj.verify(p, x) and obj.verify(p, x + 1) is False
        if obj is single:
            assert refused and refused[0] and "has length 21, but circuit tree_depth is 20" in refused[0], refused


def test_v3_golden_proofs_and_wire_formats(single, multi):
    """fixed (r, s): the committed oracle proofs through the V3 objects; LE / mixed serialisations round-trip"""
    from zerokit_amd.public_v3 import RLNProofV3, RLNWitnessInputV3
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))
    case = next(c for c in g["cases"] if c["name"] == "survey_appendix_d")
    w = case["witness"]
    wi = RLNWitnessInputV3.new_single(int(w["identity_secret"]), int(w["user_message_limit"]), int(w["message_id"]),
                                      [int(t) for t in w["path_elements"]], [int(t) for t in w["identity_path_index"]],
                                      int(w["x"]), int(w["external_nullifier"]))
    p = single.generate_proof_with_rs(wi, int(case["r"]), int(case["s"]))
    le, mixed = p.to_bytes_le(), p.to_bytes_mixed()
    pub = [int(t) for t in case["public_inputs"]]                # y, root, nullifier, x, ext
    assert le[:128].hex() == case["proof_compressed"] and len(le) == 128 + 1 + 160
    assert le[128:] == b"\0" + b"".join(v.to_bytes(32, "little") for v in pub)
    assert mixed[:129] == le[:129] and mixed[129:] == b"".join(v.to_bytes(32, "big") for v in pub)
    for q in (RLNProofV3.from_bytes_le(le + b"junk"), RLNProofV3.from_bytes_mixed(mixed)):
        assert q.to_bytes_le() == le and single.verify_with_roots(q, pub[3], [pub[1]])
    o = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_other_circuits.json")))
    mc = next(c for c in o["cases"] if c["multi"])
    i = {k: [int(t) for t in v] for k, v in mc["inputs"].items()}
    mw = RLNWitnessInputV3.new_multi(i["identitySecret"][0], i["userMessageLimit"][0], i["messageId"], i["pathElements"],
                                     i["identityPathIndex"], i["x"][0], i["externalNullifier"][0],
                                     [bool(b) for b in i["selectorUsed"]])
    mp = multi.generate_proof_with_rs(mw, int(mc["r"]), int(mc["s"]))
    assert mp.to_bytes_le()[:128].hex() == mc["proof_compressed"]
    v = mp.values
    assert v.ys + [v.root] + v.nullifiers + [v.x, v.external_nullifier] + [int(b) for b in v.selector_used] == \
        [int(t) for t in mc["public"]]
    assert multi.verify(RLNProofV3.from_bytes_mixed(mp.to_bytes_mixed()), i["x"][0])


def test_v3_verify_with_roots_and_recovery(single, multi):
    """rln/tests/proof.rs:449-620"""
    from zerokit_amd._native import RLNError
    a, b = single.generate_proof(_sw(999, x=11, ext=5)), single.generate_proof(_sw(999, x=22, ext=5))
    root = a.values.root
    assert single.verify_with_roots(a, 11, []) and single.verify_with_roots(a, 11, [123, root])
    with pytest.raises(RLNError, match="^Expected one of the provided roots$"):
        single.verify_with_roots(a, 11, [123])
    with pytest.raises(RLNError, match="^Signal value does not match$"):
        single.verify_with_roots(a, 12, [root])
    assert a.values.recover_secret(b.values) == 999
    c = single.generate_proof(_sw(999, message_id=2, x=22, ext=5))
    with pytest.raises(RLNError, match="No matching nullifier"):
        a.values.recover_secret(c.values)                        # different message id -> different nullifier
    m1 = multi.generate_proof(_mw(999, ids=(1, 7, 8, 9), x=33, ext=5))
    m2 = multi.generate_proof(_mw(999, ids=(4, 5, 6, 7), x=44, ext=5))
    assert m1.values.recover_secret(m2.values) == 999            # shared message id 7
    assert a.values.recover_secret(m1.values) == 999 == m1.values.recover_secret(a.values)   # cross-mode on id 1
    with pytest.raises(RLNError, match="^Expected one of the provided roots$"):
        multi.verify_with_roots(m1, 33, [1])
    assert multi.verify_with_roots(m1, 33, [m1.values.root])


def test_v3_partial_finish_and_stateful_tree():
    """rln/tests/proof.rs:343-447 + the stateful constructors (one device tree behind full / optimal / pm)"""
    from zerokit_amd import hashers
    from zerokit_amd._native import RLNError
    from zerokit_amd.public_v3 import PartialProofV3, RLNPartialWitnessInputV3, RLNV3, RLNWitnessInputV3
    z, g = _res()
    rln = RLNV3.stateful(20, z, g, tree="pm")
    secret = 31337
    rc = hashers.poseidon_hash_pair(hashers.poseidon_hash([secret]), 10)
    rln.set_next_leaf(5)
    rln.set_next_leaf(rc)
    assert rln.leaves_set() == 2 and rln.get_leaf(1) == rc
    elems, bits = rln.get_merkle_proof(1)
    assert bits == [1] + [0] * 19 and elems[0] == 5
    pw = RLNPartialWitnessInputV3.new(secret, 10, elems, bits)
    pp = rln.generate_partial_proof(pw)
    raw = pp.to_bytes_le()
    assert len(raw) == 6011 and raw[:8] == (5843).to_bytes(8, "little")      # the V1 form minus its version byte
    pp2 = PartialProofV3.from_bytes_le(raw)
    assert pp2.to_bytes_le() == raw
    with pytest.raises(RLNError):
        PartialProofV3.from_bytes_le(raw[:-1])
    for mid, x in ((0, 5), (9, 6)):
        w = RLNWitnessInputV3.new_single(secret, 10, mid, elems, bits, x, 808)
        p = rln.finish_proof(pp2, w)
        assert rln.verify(p, x) and rln.verify_with_roots(p, x, [rln.get_root()])
    with pytest.raises(RLNError, match="Field `path_elements` has length 19, but circuit tree_depth is 20"):
        rln.generate_partial_proof(RLNPartialWitnessInputV3.new(secret, 10, elems[:19], bits[:19]))
    with pytest.raises(RLNError, match="circuit tree_depth is 20"):
        rln.finish_proof(pp2, RLNWitnessInputV3.new_single(secret, 10, 0, elems + [0], bits + [0], 5, 808))
    rln.set_metadata(b"meta")
    assert rln.get_metadata() == b"meta"
    rln.flush()
    other = RLNV3.stateful(tree="optimal")
    other.init_tree_with_leaves([5, rc])
    assert other.get_root() == rln.get_root()


EXAMPLES = ["basic_proof", "multi_message_id", "partial_proof", "recover_secret", "stateless", "type_serialization"]


@pytest.mark.parametrize("name", EXAMPLES)
def test_reference_c_examples_run_unmodified(name, tmp_path):
    """rln/ffi_c_examples/<name>.c, compiled as-is against include/rln.h (oracle/ref_examples.mk), run against this
    librln: exit status 0 means every call succeeded and every proof the example made verified."""
    exe = os.path.join(ROOT, "oracle", "_ref", "examples", name)
    if not os.path.exists(exe):
        pytest.skip("reference example binaries not built (need /root/reference at build time)")
    res = os.path.join(ROOT, "zerokit_amd", "resources")
    d20 = tmp_path / "resources" / "tree_depth_20"
    (d20 / "multi_message_id" / "max_out_4").mkdir(parents=True)
    for f in ("rln_final.arkzkey", "graph.bin"):      # the paths common.c opens, relative to its working directory
        os.symlink(os.path.join(res, "tree_depth_20", f), d20 / f)
        os.symlink(os.path.join(res, "tree_depth_20_multi_max_out_4", f), d20 / "multi_message_id" / "max_out_4" / f)
    cwd = tmp_path / "ffi_c_examples"
    cwd.mkdir()
    r = subprocess.run([exe], cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "error" not in r.stderr.lower()
