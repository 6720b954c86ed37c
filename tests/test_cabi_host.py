"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/*.h declares, fails loudly
without a GPU, and its host-side logic (parsers, Keccak, point compression, the Groth16 verifier, wire
formats) agrees with the oracle / the reference's vectors.  No compute call here needs a GPU."""
import ctypes as C
import json
import os
import re

import pytest

from oracle.pyref import arkzkey as o_zkey
from oracle.pyref.keccak import hash_to_field_le as o_htf
from zerokit_amd import _native, hashers
from zerokit_amd._native import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "zerokit_amd", "resources", "tree_depth_20")


def _decls(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:ffi|rlnamd)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = lib()
    names = _decls("rln.h") + _decls("rln_amd.h")
    assert len(names) > 100
    for n in names:
        assert hasattr(L, n), "missing export " + n
        assert n in _native.SIGNATURES, "no ctypes signature for " + n


def test_no_cpu_fallback_without_gpu():
    if lib().rlnamd_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_native.RLNError, match="no HIP device"):
        hashers.poseidon_hash([1, 2])
    h = C.c_void_p()
    z = open(os.path.join(RES, "rln_final.arkzkey"), "rb").read()
    g = open(os.path.join(RES, "graph.bin"), "rb").read()
    assert lib().rlnamd_prover_new(z, len(z), g, len(g), 64, 0, C.byref(h)) == 2  # RLNAMD_ERR_NO_DEVICE
    assert lib().rlnamd_tree_new(4, C.byref(h)) == 2


def test_hash_to_field_matches_oracle():
    for s in (b"", b"test-merkle-proof", b"hey hey", b"x" * 135, b"y" * 136, b"z" * 137, bytes(range(256)) * 3):
        assert hashers.hash_to_field_le(s) == o_htf(s)
        assert hashers.hash_to_field_be(s) == o_htf(s)


def test_parse_resources_counts():
    z = open(os.path.join(RES, "rln_final.arkzkey"), "rb").read()
    g = open(os.path.join(RES, "graph.bin"), "rb").read()
    counts = (C.c_uint64 * 13)()
    assert lib().rlnamd_parse_resources(z, len(z), g, len(g), counts) == 0, _native.last_error()
    assert list(counts) == [6, 5839, 5820, 9658, 13282, 5844, 8192, 5838, 23414, 5844, 20, 1, 46]
    # loader error cases (circuit/mod.rs:333-368)
    assert lib().rlnamd_parse_resources(b"", 0, g, len(g), counts) != 0
    assert "Empty" in _native.last_error()
    assert lib().rlnamd_parse_resources(z, len(z), b"not a graph file....", 20, counts) != 0
    assert "magic" in _native.last_error().lower()
    assert lib().rlnamd_parse_resources(z[:1000], 1000, g, len(g), counts) != 0


def _vec(name):
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))
    return next(c for c in d["cases"] if c["name"] == name)


SNARKJS_PROOF = (  # rln/tests/public.rs:84-135
    (606446415626469993821291758185575230335423926365686267140465300918089871829,
     14881534001609371078663128199084130129622943308489025453376548677995646280161),
    ((18053812507994813734583839134426913715767914942522332114506614735770984570178,
      11219916332635123001710279198522635266707985651975761715977705052386984005181),
     (17371289494006920912949790045699521359436706797224428511776122168520286372970,
      14038575727257298083893642903204723310279435927688342924358714639926373603890)),
    (17701377127561410274754535747274973758826089226897242202671882899370780845888,
     12608543716397255084418384146504333522628400182843246910626782513289789807030))
SNARKJS_PUBLIC = [  # y, root, nullifier, x, external_nullifier
    16401008481486069296141645075505218976370369489687327284155463920202585288271,
    8502402278351299594663821509741133196466235670407051417832304486953898514733,
    9102791780887227194595604713537772536258726662792598131262022534710887343694,
    20645213238265527935869146898028115621427162613172918400241870500502509785943,
    21074405743803627666274838159589343934394162804826017440941339048886754734203]


def _verify(proof128, public):
    z = open(os.path.join(RES, "rln_final.arkzkey"), "rb").read()
    ok = C.c_int(-1)
    rc = lib().rlnamd_verify_with_zkey(z, len(z), proof128, b"".join(v.to_bytes(32, "little") for v in public),
                                       C.byref(ok))
    assert rc == 0, _native.last_error()
    return bool(ok.value)


def test_host_verifier_snarkjs_kat_and_goldens():
    proof = o_zkey.proof_compress(*SNARKJS_PROOF)
    assert _verify(proof, SNARKJS_PUBLIC)
    wrong = [SNARKJS_PUBLIC[i] for i in (1, 4, 3, 0, 2)]  # serialisation order must fail
    assert not _verify(proof, wrong)
    bad = list(SNARKJS_PUBLIC)
    bad[0] = (bad[0] + 1) % hashers.R
    assert not _verify(proof, bad)
    v = _vec("config1_bench_witness")
    assert _verify(bytes.fromhex(v["proof_compressed"]), [int(x) for x in v["public_inputs"]])
    # swapping A and C keeps both on the curve but breaks the equation
    p = bytes.fromhex(v["proof_compressed"])
    assert not _verify(p[96:128] + p[32:96] + p[0:32], [int(x) for x in v["public_inputs"]])


def test_verify_many_on_host_threads_matches_single_verification():
    """rlnamd_verify_many_with_zkey (EXT, no GPU): the golden proofs, a changed public input, a malformed proof and a
    non-canonical public input in one call; the verdicts do not depend on the thread count."""
    from zerokit_amd.batch import verify_many_with_zkey
    z = open(os.path.join(RES, "rln_final.arkzkey"), "rb").read()
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]
    proofs = [bytes.fromhex(c["proof_compressed"]) for c in cases] * 3
    pubs = [[int(v) for v in c["public_inputs"]] for c in cases] * 3
    want = [True] * len(proofs)
    pubs[1] = [pubs[1][0] ^ 1] + pubs[1][1:]
    want[1] = False
    proofs[2] = bytes([proofs[2][0] ^ 1]) + proofs[2][1:]
    want[2] = False
    proofs[4] = proofs[4][96:128] + proofs[4][32:96] + proofs[4][0:32]
    want[4] = False
    for threads in (1, 3, 0):
        assert verify_many_with_zkey(z, proofs, pubs, threads=threads) == want, threads
    assert verify_many_with_zkey(z, [], []) == []
    for i, (pr, pi) in enumerate(zip(proofs, pubs)):
        if i != 2:   # the malformed proof is an error for the single call (as ffi_verify_rln_proof), False here
            assert _verify(pr, pi) == want[i]


def test_point_compression_matches_oracle():
    for name in ("config1_bench_witness", "survey_appendix_d", "config2_1"):
        v = _vec(name)
        coords = [int(v["a"][0]), int(v["a"][1]), int(v["b"][0][0]), int(v["b"][0][1]), int(v["b"][1][0]),
                  int(v["b"][1][1]), int(v["c"][0]), int(v["c"][1])]
        raw = b"".join(c.to_bytes(32, "little") for c in coords)
        out = C.create_string_buffer(128)
        assert lib().rlnamd_proof_compress(raw, out) == 0
        assert out.raw.hex() == v["proof_compressed"]
        back = C.create_string_buffer(256)
        assert lib().rlnamd_proof_decompress(out.raw, back) == 0
        assert back.raw == raw
    junk = C.create_string_buffer(256)
    assert lib().rlnamd_proof_decompress(b"\x04" + bytes(127), junk) != 0  # x = 4 is not on the curve


def test_ffi_field_and_vector_helpers():
    from zerokit_amd.public import _cfr, _take_bytes, _vec_cfr, _vec_u8, _take_cfr
    L = lib()
    x = 0x1234567890ABCDEF << 100
    le = _take_bytes(L.ffi_cfr_to_bytes_le(C.byref(_cfr(x))))
    be = _take_bytes(L.ffi_cfr_to_bytes_be(C.byref(_cfr(x))))
    assert le == x.to_bytes(32, "little") and be == x.to_bytes(32, "big")
    v, _k = _vec_u8(be)
    r = L.ffi_bytes_be_to_cfr(C.byref(v))
    assert r.ok and _take_cfr(C.cast(C.c_void_p(r.ok), C.POINTER(_native.CFr))) == x
    # non-canonical element is rejected (serialize.rs:108-110)
    v, _k = _vec_u8(hashers.R.to_bytes(32, "little"))
    r = L.ffi_bytes_le_to_cfr(C.byref(v))
    assert not r.ok and b"Non-canonical" in C.string_at(r.err.ptr, r.err.len)
    # Vec<Fr>: 8-byte length prefix + elements (utils.rs:123-156)
    vals = [1, 2, hashers.R - 1]
    vc, _k = _vec_cfr(vals)
    b_le = _take_bytes(L.ffi_vec_cfr_to_bytes_le(C.byref(vc)))
    assert b_le == (3).to_bytes(8, "little") + b"".join(t.to_bytes(32, "little") for t in vals)
    b_be = _take_bytes(L.ffi_vec_cfr_to_bytes_be(C.byref(vc)))
    assert b_be == (3).to_bytes(8, "big") + b"".join(t.to_bytes(32, "big") for t in vals)
    vb, _k = _vec_u8(b_be)
    rr = L.ffi_bytes_be_to_vec_cfr(C.byref(vb))
    assert rr.ok.len == 3 and int.from_bytes(bytes(rr.ok.ptr[2].le), "little") == hashers.R - 1
    L.ffi_vec_cfr_free(rr.ok)
    # push / get / len
    vec = L.ffi_vec_cfr_new(0)
    for t in range(10):
        L.ffi_vec_cfr_push(C.byref(vec), C.byref(_cfr(t * t)))
    assert L.ffi_vec_cfr_len(C.byref(vec)) == 10
    assert int.from_bytes(bytes(L.ffi_vec_cfr_get(C.byref(vec), 7).contents.le), "little") == 49
    assert not L.ffi_vec_cfr_get(C.byref(vec), 10)
    L.ffi_vec_cfr_free(vec)
    dbg = L.ffi_cfr_debug(C.byref(_cfr(12345)))
    assert C.string_at(dbg.ptr) == b"12345"
    L.ffi_c_string_free(dbg)


def test_witness_and_proof_wire_formats():
    """rln/tests/serialize.rs round trips + the 837 / 290 byte sizes of SURVEY Appendix B."""
    from zerokit_amd.public import RLNProof, RLNWitnessInput, RLNError
    v = _vec("config1_bench_witness")
    w = v["witness"]
    wi = RLNWitnessInput(int(w["identity_secret"]), 100, 1, [int(t) for t in w["path_elements"]],
                         [int(t) for t in w["identity_path_index"]], int(w["x"]), int(w["external_nullifier"]))
    le, be = wi.to_bytes_le(), wi.to_bytes_be()
    assert len(le) == 837 and len(be) == 837 and le[0] == 0
    assert le[1:33] == int(w["identity_secret"]).to_bytes(32, "little")
    assert be[1:33] == int(w["identity_secret"]).to_bytes(32, "big")
    assert le[97:105] == (20).to_bytes(8, "little") and be[97:105] == (20).to_bytes(8, "big")
    assert RLNWitnessInput.from_bytes_le(le).to_bytes_be() == be
    assert RLNWitnessInput.from_bytes_be(be).to_bytes_le() == le
    with pytest.raises(RLNError, match="too short"):
        RLNWitnessInput.from_bytes_le(le[:-1])
    with pytest.raises(RLNError, match="Expected to read"):
        RLNWitnessInput.from_bytes_le(le + b"\0")
    # validation (witness.rs:87-107)
    with pytest.raises(RLNError, match="cannot be zero"):
        RLNWitnessInput(1, 0, 0, [1], [0], 1, 1)
    with pytest.raises(RLNError, match="length mismatch"):
        RLNWitnessInput(1, 10, 0, [1, 2], [0], 1, 1)
    with pytest.raises(RLNError, match="not within user_message_limit"):
        RLNWitnessInput(1, 10, 10, [1], [0], 1, 1)
    # proof: golden 290-byte LE form parses, re-serialises identically, BE form differs only in the values
    raw = bytes.fromhex(v["rln_proof_le"])
    assert len(raw) == 290
    p = RLNProof.from_bytes_le(raw)
    assert p.to_bytes_le() == raw
    be = p.to_bytes_be()
    assert be[:130] == raw[:130] and be[130:162] == raw[130:162][::-1]
    assert RLNProof.from_bytes_be(be).to_bytes_le() == raw
    vals = p.values
    assert [vals.y, vals.root, vals.nullifier, vals.x, vals.external_nullifier] == [int(t) for t in v["public_inputs"]]
    with pytest.raises(RLNError):
        RLNProof.from_bytes_le(raw[:200])
    with pytest.raises(RLNError, match="version byte"):
        RLNProof.from_bytes_le(b"\x07" + raw[1:])
    with pytest.raises(RLNError, match="invalid data"):
        RLNProof.from_bytes_le(raw[:1] + b"\x04" + bytes(31) + raw[33:])  # x = 4 is not on the curve


def test_header_compiles_as_c_and_program_fails_loudly_without_gpu(tmp_path):
    """include/rln.h is plain C: a C caller (tests/host/ffi_smoke.c) builds against it with gcc -Wall and links
    -lrln; without a device ffi_rln_new returns the error string instead of a handle (no CPU fallback)."""
    import subprocess
    if lib().rlnamd_device_count() > 0:
        pytest.skip("GPU present")
    exe = str(tmp_path / "ffi_smoke")
    libdir = os.path.join(ROOT, "zerokit_amd", "lib")
    res = subprocess.run(["gcc", "-Wall", "-Wextra", "-Werror", "-std=c11", "-I", os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "tests", "host", "ffi_smoke.c"), "-L", libdir, "-lrln",
                          "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "no HIP device" in out.stderr


def test_multi_message_witness_validation_and_wire_format():
    """rln/tests/public.rs:1528-1676 (validation rules) + the 0x01 wire layout of protocol/witness.rs:402-432"""
    from zerokit_amd.public import RLNError, RLNWitnessInput
    pe, pi = list(range(100, 120)), [i & 1 for i in range(20)]
    mk = lambda limit, ids, sel: RLNWitnessInput.new_multi(7, limit, ids, pe, pi, 11, 13, sel)
    with pytest.raises(RLNError, match="at least one message_id"):
        mk(10, [], [])
    with pytest.raises(RLNError, match="message_ids has length 2, but the field selector_used has length 1"):
        mk(10, [0, 1], [True])
    with pytest.raises(RLNError, match=r"Message id \(10\) is not within user_message_limit \(10\)"):
        mk(10, [0, 10], [True, True])
    mk(10, [0, 10], [True, False])  # inactive slot above the limit is fine
    with pytest.raises(RLNError, match="cannot be zero"):
        mk(0, [0], [True])
    with pytest.raises(RLNError, match="Duplicate message ID"):
        mk(10, [5, 5, 1, 2], [True, True, False, False])
    mk(10, [0, 0, 1, 2], [False, False, True, True])  # duplicates only matter when active
    with pytest.raises(RLNError, match="At least one selector_used"):
        mk(10, [0, 1, 2, 3], [False] * 4)
    w = mk(10, [0, 1, 2, 3], [True, True, False, False])
    assert w.version_byte == 1 and w.message_ids == [0, 1, 2, 3] and w.selector_used == [True, True, False, False]
    le, be = w.to_bytes_le(), w.to_bytes_be()
    # 1 + secret + limit + (8 + 20*32) + (8 + 20) + x + ext + (8 + 4*32) + (8 + 4)
    assert len(le) == len(be) == 1 + 64 + 648 + 28 + 64 + 136 + 12 and le[0] == 1
    tail = (4).to_bytes(8, "little") + b"\x01\x01\x00\x00"
    assert le.endswith(tail) and be.endswith((4).to_bytes(8, "big") + b"\x01\x01\x00\x00")
    w2 = RLNWitnessInput.from_bytes_le(le)
    assert w2.version_byte == 1 and w2.to_bytes_be() == be and RLNWitnessInput.from_bytes_be(be).to_bytes_le() == le
    with pytest.raises(RLNError, match="Non-canonical bool"):
        RLNWitnessInput.from_bytes_le(le[:-1] + b"\x02")
    with pytest.raises(RLNError, match="version byte"):
        RLNWitnessInput.from_bytes_le(b"\x02" + le[1:])
    # the witness-calculator JSON: keys sorted, compact, decimal strings (serde_json default map)
    j = json.loads(w.to_bigint_json())
    assert list(j) == sorted(j) and j["messageId"] == ["0", "1", "2", "3"] and j["selectorUsed"] == ["1", "1", "0", "0"]
    assert j["pathElements"] == [str(v) for v in pe] and j["identityPathIndex"] == [str(v) for v in pi]
    assert " " not in w.to_bigint_json()
    single = RLNWitnessInput(7, 10, 3, pe, pi, 11, 13)
    js = json.loads(single.to_bigint_json())
    assert js["messageId"] == "3" and "selectorUsed" not in js and js["identitySecret"] == "7"


def test_partial_witness_wire_format():
    """protocol/witness.rs:631-760"""
    from zerokit_amd.public import RLNError, RLNPartialWitnessInput
    pe, pi = list(range(1, 21)), [1] * 20
    p = RLNPartialWitnessInput(5, 9, pe, pi)
    le, be = p.to_bytes_le(), p.to_bytes_be()
    assert len(le) == 1 + 64 + 8 + 640 + 8 + 20 and le[0] == 0
    q = RLNPartialWitnessInput.from_bytes_le(le)
    assert (q.identity_secret, q.user_message_limit, q.path_elements, q.identity_path_index) == (5, 9, pe, pi)
    assert RLNPartialWitnessInput.from_bytes_be(be).to_bytes_le() == le
    with pytest.raises(RLNError, match="Expected to read"):
        RLNPartialWitnessInput.from_bytes_le(le + b"\0")
    with pytest.raises(RLNError, match="cannot be zero"):
        RLNPartialWitnessInput.from_bytes_le(le[:33] + bytes(32) + le[65:])


def test_slashing_host_math_and_multi_proof_values_bytes():
    """protocol/slashing.rs:12-100 and the MultiV1 proof-values layout (protocol/proof.rs:205-236)"""
    from oracle.pyref.bn254 import R
    from oracle.pyref.keygen import compute_id_secret as o_secret
    from zerokit_amd.public import RLNError, RLNProofValues, compute_id_secret, recover_id_secret
    a0, a1 = 123456789, 987654321987654321
    sh = lambda x: (x, (a0 + x * a1) % R)
    assert compute_id_secret(sh(5), sh(R - 3)) == a0 == o_secret(sh(5), sh(R - 3))
    with pytest.raises(RLNError, match="division by zero"):
        compute_id_secret(sh(5), sh(5))

    def multi_bytes(root, ext, x, ys, nulls, sel, order="little"):
        f = lambda v: int(v).to_bytes(32, order)
        n8 = lambda k: k.to_bytes(8, order)
        return (b"\x01" + f(root) + f(ext) + f(x) + n8(len(ys)) + b"".join(map(f, ys)) + n8(len(nulls)) +
                b"".join(map(f, nulls)) + n8(len(sel)) + bytes(sel))

    x1, x2 = 1111, 2222
    v1 = RLNProofValues.from_bytes_le(multi_bytes(9, 77, x1, [0, sh(x1)[1], 5, 0], [0, 42, 43, 0], [0, 1, 1, 0]))
    v2 = RLNProofValues.from_bytes_le(multi_bytes(9, 77, x2, [sh(x2)[1], 0, 0, 6], [42, 0, 0, 99], [1, 0, 0, 1]))
    assert v1.version_byte == 1 and v1.ys[1] == sh(x1)[1] and v1.selector_used == [False, True, True, False]
    assert v1.to_bytes_le() == multi_bytes(9, 77, x1, [0, sh(x1)[1], 5, 0], [0, 42, 43, 0], [0, 1, 1, 0])
    assert v1.to_bytes_be() == multi_bytes(9, 77, x1, [0, sh(x1)[1], 5, 0], [0, 42, 43, 0], [0, 1, 1, 0], "big")
    assert recover_id_secret(v1, v2) == a0  # slot 1 of v1 and slot 0 of v2 share nullifier 42
    v3 = RLNProofValues.from_bytes_le(multi_bytes(9, 77, x2, [1, 2, 3, 4], [42, 1, 2, 3], [0, 1, 1, 1]))
    with pytest.raises(RLNError, match="No matching nullifier"):
        recover_id_secret(v1, v3)  # 42 sits in an inactive slot
    v4 = RLNProofValues.from_bytes_le(multi_bytes(9, 78, x2, [1, 2, 3, 4], [42, 1, 2, 3], [1, 1, 1, 1]))
    with pytest.raises(RLNError, match="External nullifiers mismatch: 77 != 78"):
        recover_id_secret(v1, v4)
    with pytest.raises(RLNError, match="does not exist on the `MultiV1` variant"):
        v1.y
    with pytest.raises(RLNError, match="ys has length 2, but the field nullifiers has length 1"):
        RLNProofValues.from_bytes_le(multi_bytes(9, 77, 1, [1, 2], [3], [1, 0]))


def test_poseidon_parameter_derivation_dense_and_sparse_forms():
    """The library's own Grain-LFSR derivation, checked on the host for every width t = 2..9: the dense rounds of
    the reference (poseidon_hash.rs:97-135) and the equivalent sparse partial rounds the device kernels use must
    both give the oracle's hash (reference KATs: utils/tests/poseidon_hash_test.rs:21-130).  No device involved."""
    import ctypes as C
    import random
    from oracle.pyref.poseidon import poseidon
    from zerokit_amd import lib
    from zerokit_amd._native import check
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    rnd = random.Random(5)
    for arity in range(1, 9):
        for row in ([0] * arity, [R - 1] * arity, [rnd.randrange(R) for _ in range(arity)]):
            buf = b"".join(v.to_bytes(32, "little") for v in row)
            d, s = C.create_string_buffer(32), C.create_string_buffer(32)
            check(lib().rlnamd_poseidon_params_check(buf, arity, d, s))
            want = poseidon(row)
            assert int.from_bytes(d.raw, "little") == want, ("dense", arity)
            assert int.from_bytes(s.raw, "little") == want, ("sparse", arity)
    d, s = C.create_string_buffer(32), C.create_string_buffer(32)
    assert lib().rlnamd_poseidon_params_check((1).to_bytes(32, "little") + (2).to_bytes(32, "little"), 2, d, s) == 0
    assert int.from_bytes(s.raw, "little") == \
        7853200120776062878684798364095072458815029376092732009249414926327459813530   # circomlib's published vector for poseidon([1, 2]) (not from the reference tree)


def test_verifier_rejects_g2_point_outside_the_order_r_subgroup():
    """ark-serialize Validate::Yes (the reference's proof deserialiser, protocol/proof.rs:434-449 ->
    Proof::deserialize_compressed) checks [r]B == 0; the twist has a large cofactor, so an x picked at random
    gives a curve point outside G2 with overwhelming probability.  Every entry that takes raw proof bytes must
    refuse it as a serialisation error, not feed it to the pairing."""
    from oracle.pyref.bn254 import G2, R
    v = _vec("config1_bench_witness")
    good = bytes.fromhex(v["proof_compressed"])
    public = [int(x) for x in v["public_inputs"]]
    found = None
    for k in range(1, 200):
        cand = bytearray(good[32:96])
        cand[0:8] = k.to_bytes(8, "little")          # perturb x.c0
        try:
            B = o_zkey.g2_decompress(bytes(cand))
        except ValueError:                            # x^3 + b' is not a square for this x
            continue
        if B is None or not G2.on_curve(B):
            continue
        if G2.mul(B, R) is not None:                  # on the curve, not in the r-torsion
            found = bytes(cand)
            break
    assert found is not None
    z = open(os.path.join(RES, "rln_final.arkzkey"), "rb").read()
    ok = C.c_int(-1)
    rc = lib().rlnamd_verify_with_zkey(z, len(z), good[:32] + found + good[96:], b"".join(
        x.to_bytes(32, "little") for x in public), C.byref(ok))
    assert rc != 0 and ok.value == 0
    assert "Proof serialization error" in _native.last_error()
    assert _verify(good, public)                      # the untouched proof still verifies


def test_python_mirrors_refuse_non_canonical_integers():
    """CFr values of the C ABI are canonical by construction (ffi_bytes_*_to_cfr); the Python mirrors write the
    bytes directly, so they apply the same rule instead of letting x + r pass the host-side comparisons."""
    from zerokit_amd import batch, public
    for bad in (hashers.R, hashers.R + 5, 1 << 255, -1):
        with pytest.raises(_native.RLNError, match="Non-canonical field element"):
            public._cfr(bad)
        with pytest.raises(_native.RLNError, match="Non-canonical field element"):
            public._vec_cfr([1, bad])
        with pytest.raises(_native.RLNError, match="Non-canonical field element"):
            batch._b(bad)
    assert bytes(public._cfr(hashers.R - 1).le) == (hashers.R - 1).to_bytes(32, "little")
    assert batch._b(batch._Q - 1, batch._Q) == (batch._Q - 1).to_bytes(32, "little")
    with pytest.raises(_native.RLNError):
        batch._b(batch._Q, batch._Q)


def test_vectorised_workload_generator_matches_the_oracle_generator():
    """zerokit_amd/workload.py (numpy, index ranges; used by bench.py and the size tests) against the sequential
    restatement in oracle/pyref/workload.py, including a range that starts inside the stream and the packed form"""
    from oracle.pyref import workload as o
    from zerokit_amd import workload as w
    ws, rs = o.config2_witnesses(150)
    assert w.config2_range(0, 150) == (ws, rs)
    assert w.config2_range(97, 53) == (ws[97:], rs[97:])
    slots = {"identitySecret": (1, 1), "userMessageLimit": (2, 1), "messageId": (3, 1), "pathElements": (4, 20),
             "identityPathIndex": (24, 20), "x": (44, 1), "externalNullifier": (45, 1)}
    inp, rsb = w.config2_packed(slots, 46, 97, 53)
    assert len(inp) == 53 * 46 * 32 and len(rsb) == 53 * 64
    for i in (0, 52):
        row = [int.from_bytes(inp[(46 * i + k) * 32:(46 * i + k + 1) * 32], "little") for k in range(46)]
        wi = ws[97 + i]
        assert row == [1, wi["identity_secret"], 100, wi["message_id"]] + wi["path_elements"] + \
            wi["identity_path_index"] + [wi["x"], wi["external_nullifier"]]
        assert (int.from_bytes(rsb[64 * i:64 * i + 32], "little"),
                int.from_bytes(rsb[64 * i + 32:64 * i + 64], "little")) == rs[97 + i]


def test_config_path_is_parsed_before_any_device_is_needed(tmp_path):
    """ffi_rln_new reads config_path first (RLN::new parses the config before anything else, public.rs:113): a malformed
    file is the reference's configuration error on a box without a GPU too, and a well-formed one -- including this
    backend's own keys window_bits / max_batch / devices beside the PmTreeConfig keys -- gets as far as the device check"""
    import ctypes as C
    from zerokit_amd import lib
    from zerokit_amd._native import VecU8

    def new(cfg_text):
        p = tmp_path / "cfg.json"
        p.write_text(cfg_text)
        r = lib().ffi_rln_new(20, str(p).encode())
        assert not r.ok
        msg = C.string_at(r.err.ptr, r.err.len).decode()
        lib().ffi_c_string_free(r.err)
        return msg

    assert "Error while reading pmtree config" in new('{"devices": [0, }')
    # the devices list is read as strictly as serde_json reads a Vec: separators required, no trailing comma, no bare sign
    for bad_list in ('[1 2]', '[1,]', '[-]', '[1,,2]', '["0"]', '[1.5]', '[0'):
        assert "Error while reading pmtree config" in new('{"devices": %s}' % bad_list), bad_list
    assert "negative device ordinal" in new('{"devices": [0, -1]}')
    assert "Error while reading pmtree config" in new('{"max_batch": }')
    assert "missing path" in new('{"temporary": false}')
    # "profile" names an operating point instead of the schedule's number; an unknown name is an error, not a default
    assert 'expected "latency", "throughput" or "small", got "fast"' in new('{"profile": "fast"}')
    assert "revive_after: expected" in new('{"devices": [0, 1], "failover": 1, "revive_after": -3}')
    assert "partial_cache: expected" in new('{"partial_cache": -1}')
    assert "auto_partial: expected" in new('{"auto_partial": 100000}')
    assert "gather_calls: expected" in new('{"gather_calls": -2}')
    assert "gather_window_us: expected" in new('{"gather_window_us": 1000000}')
    if lib().rlnamd_device_count() == 0:
        for ok_cfg in ('{"window_bits": 7150114, "max_batch": 1024}', '{"profile": "throughput"}', '{"profile": "small", "max_batch": 128}',
                       '{"devices": [0, 1, 2, 3], "temporary": true}', '{"devices": [0, 1], "failover": 2, "revive_after": 0}', '{"partial_cache": 1024}', '{"auto_partial": 16}', '{"gather_calls": 0}', '{"gather_calls": 32, "gather_window_us": 0}',
                       '{"cache_capacity": 1073741824, "flush_every_ms": 500, "mode": "HighThroughput", "use_compression": false}',
                       # unknown keys carry any JSON value (PmTreeConfig::from_str ignores them)
                       '{"foo": ["a]", {"b": [1, 2]}], "bar": {"x": {"y": "}"}}, "devices": [0]}', '{"devices": []}'):
            assert "no HIP device" in new(ok_cfg)
