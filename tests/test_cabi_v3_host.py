"""Host-side checks of the ffi_rln_v3_* mirror (no GPU): object construction, V3 error texts and the V3 wire formats
(rln/tests/serialize.rs:262-560 replayed through the C ABI)."""
import pytest

from zerokit_amd._native import RLNError
from zerokit_amd.public_v3 import (RLNPartialWitnessInputV3, RLNProofValuesV3, RLNWitnessInputV3, compute_id_secret)

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def f32(v, order="little"):
    return int(v).to_bytes(32, order)


def n8(v, order="little"):
    return int(v).to_bytes(8, order)


def single():
    return RLNWitnessInputV3.new_single(42, 10, 3, [1, 2], [0, 1], 5, 7)        # serialize.rs:264-275


def multi():
    return RLNWitnessInputV3.new_multi(99, 10, [0, 1], [1, 2], [0, 1], 5, 7, [True, False])   # :277-289


def test_witness_v3_layouts_and_roundtrips():
    w = single()
    le, be = w.to_bytes_le(), w.to_bytes_be()
    # LE = tag + the struct declaration order (message_id LAST); BE = tag + the hand-written order (message_id third)
    assert le == (b"\0" + f32(42) + f32(10) + n8(2) + f32(1) + f32(2) + n8(2) + b"\0\1" + f32(5) + f32(7) + f32(3))
    assert be == (b"\0" + f32(42, "big") + f32(10, "big") + f32(3, "big") + n8(2, "big") + f32(1, "big") +
                  f32(2, "big") + n8(2, "big") + b"\0\1" + f32(5, "big") + f32(7, "big"))
    w2 = RLNWitnessInputV3.from_bytes_le(le)
    assert (w2.identity_secret, w2.user_message_limit, w2.message_id, w2.path_elements, w2.identity_path_index, w2.x,
            w2.external_nullifier) == (42, 10, 3, [1, 2], [0, 1], 5, 7)
    assert RLNWitnessInputV3.from_bytes_be(be).to_bytes_le() == le
    m = multi()
    mle, mbe = m.to_bytes_le(), m.to_bytes_be()
    assert mle == (b"\1" + f32(99) + f32(10) + n8(2) + f32(1) + f32(2) + n8(2) + b"\0\1" + f32(5) + f32(7) + n8(2) +
                   f32(0) + f32(1) + n8(2) + b"\1\0")
    assert mbe[-10:] == n8(2, "big") + b"\1\0"
    m2 = RLNWitnessInputV3.from_bytes_be(mbe)
    assert m2.message_ids == [0, 1] and m2.selector_used == [True, False] and m2.to_bytes_le() == mle
    with pytest.raises(RLNError, match="witness is Multi; use get_message_ids"):
        m.message_id
    with pytest.raises(RLNError, match="witness is Single; use get_message_id"):
        w.message_ids
    with pytest.raises(RLNError, match="selector_used is Multi-only"):
        w.selector_used
    # invalid tag / truncation / trailing bytes (serialize.rs:413-487)
    for frm, raw in ((RLNWitnessInputV3.from_bytes_le, le), (RLNWitnessInputV3.from_bytes_be, be)):
        with pytest.raises(RLNError, match="invalid data"):
            frm(b"\x02" + raw[1:])
        with pytest.raises(RLNError, match="I/O error"):
            frm(raw[:-1])
        with pytest.raises(RLNError, match="I/O error"):
            frm(b"")
        assert frm(raw + b"\xAA\xBB").x == 5
    with pytest.raises(RLNError, match="Non-canonical field element"):
        RLNWitnessInputV3.from_bytes_be(b"\0" + f32(R, "big") + be[33:])
    with pytest.raises(RLNError, match="invalid data"):
        RLNWitnessInputV3.from_bytes_le(b"\0" + f32(R) + le[33:])
    with pytest.raises(RLNError, match="Non-canonical bool byte: expected 0x00 or 0x01, got 0x02"):
        RLNWitnessInputV3.from_bytes_be(mbe[:-1] + b"\x02")
    with pytest.raises(RLNError, match="invalid data"):
        RLNWitnessInputV3.from_bytes_le(mle[:-1] + b"\x02")


def test_witness_v3_validation_texts():
    """error.rs:125-165"""
    with pytest.raises(RLNError, match="User message limit cannot be zero"):
        RLNWitnessInputV3.new_single(1, 0, 0, [1], [0], 1, 1)
    with pytest.raises(RLNError, match="Field `path_elements` has length 2, but field `identity_path_index` has length 1"):
        RLNWitnessInputV3.new_single(1, 10, 0, [1, 2], [0], 1, 1)
    with pytest.raises(RLNError, match=r"Message id \(10\) is not within user_message_limit \(10\)"):
        RLNWitnessInputV3.new_single(1, 10, 10, [1], [0], 1, 1)
    mk = lambda ids, sel, limit=10: RLNWitnessInputV3.new_multi(1, limit, ids, [1], [0], 1, 1, sel)
    with pytest.raises(RLNError, match="`message_ids` must contain at least one"):
        mk([], [])
    with pytest.raises(RLNError, match="Field `message_ids` has length 2, but field `selector_used` has length 1"):
        mk([0, 1], [True])
    with pytest.raises(RLNError, match="At least one value in `selector_used` must be true"):
        mk([0, 1], [False, False])
    with pytest.raises(RLNError, match="Duplicate message ID found in `message_ids`"):
        mk([5, 5], [True, True])
    mk([5, 5], [True, False])
    with pytest.raises(RLNError, match="not within user_message_limit"):
        mk([0, 10], [True, True])
    mk([0, 10], [True, False])
    with pytest.raises(RLNError, match="cannot be zero"):
        RLNPartialWitnessInputV3.new(1, 0, [1], [0])
    with pytest.raises(RLNError, match="Field `path_elements` has length 1, but field `identity_path_index` has length 2"):
        RLNPartialWitnessInputV3.new(1, 5, [1], [0, 1])


def test_partial_witness_v3_layout():
    p = RLNPartialWitnessInputV3.new(42, 10, [1, 2], [0, 1])                       # serialize.rs:291-299
    le, be = p.to_bytes_le(), p.to_bytes_be()
    assert le == f32(42) + f32(10) + n8(2) + f32(1) + f32(2) + n8(2) + b"\0\1"     # no tag
    assert be == f32(42, "big") + f32(10, "big") + n8(2, "big") + f32(1, "big") + f32(2, "big") + n8(2, "big") + b"\0\1"
    q = RLNPartialWitnessInputV3.from_bytes_le(le + b"\xff")                       # extra bytes accepted (:507-513)
    assert (q.identity_secret, q.user_message_limit, q.path_elements) == (42, 10, [1, 2])
    assert RLNPartialWitnessInputV3.from_bytes_be(be).to_bytes_le() == le
    with pytest.raises(RLNError, match="I/O error"):
        RLNPartialWitnessInputV3.from_bytes_le(le[:-1])
    assert single().to_partial().to_bytes_le() == le


def test_proof_values_v3_layout_and_recovery():
    """proof.rs:981-1140: y | root | nullifier | x | ext and the Multi form; RecoverSecret in every pairing"""
    sle = b"\0" + f32(4) + f32(1) + f32(5) + f32(2) + f32(3)                       # serialize.rs:301-309
    v = RLNProofValuesV3.from_bytes_le(sle)
    assert (v.y, v.root, v.nullifier, v.x, v.external_nullifier) == (4, 1, 5, 2, 3)
    assert v.to_bytes_le() == sle and v.to_bytes_be() == b"\0" + b"".join(f32(t, "big") for t in (4, 1, 5, 2, 3))
    mle = (b"\1" + n8(2) + f32(40) + f32(50) + f32(10) + n8(2) + f32(60) + f32(70) + f32(20) + f32(30) + n8(2) +
           b"\1\0")                                                                 # :311-320
    m = RLNProofValuesV3.from_bytes_le(mle)
    assert (m.ys, m.root, m.nullifiers, m.x, m.external_nullifier, m.selector_used) == (
        [40, 50], 10, [60, 70], 20, 30, [True, False])
    assert RLNProofValuesV3.from_bytes_be(m.to_bytes_be()).to_bytes_le() == mle
    with pytest.raises(RLNError, match="values are Multi; use get_ys"):
        m.y
    with pytest.raises(RLNError, match="values are Single; use get_nullifier"):
        v.nullifiers
    with pytest.raises(RLNError, match="invalid data"):
        RLNProofValuesV3.from_bytes_le(b"\x09" + sle[1:])
    # recovery: shares of y = a0 + x a1
    a0, a1 = 777, 123456789
    sh = lambda x: (a0 + x * a1) % R
    sv = lambda x, null, ext=9: RLNProofValuesV3.from_bytes_le(b"\0" + f32(sh(x)) + f32(1) + f32(null) + f32(x) + f32(ext))
    mv = lambda x, nulls, sel, ext=9: RLNProofValuesV3.from_bytes_le(
        b"\1" + n8(len(nulls)) + b"".join(f32(sh(x)) for _ in nulls) + f32(1) + n8(len(nulls)) +
        b"".join(map(f32, nulls)) + f32(x) + f32(ext) + n8(len(sel)) + bytes(sel))
    assert sv(11, 55).recover_secret(sv(22, 55)) == a0 == compute_id_secret((11, sh(11)), (22, sh(22)))
    with pytest.raises(RLNError, match="No matching nullifier"):
        sv(11, 55).recover_secret(sv(22, 56))                                      # rln/tests/proof.rs:181-198
    with pytest.raises(RLNError, match="External nullifiers mismatch: 9 != 8"):
        sv(11, 55).recover_secret(sv(22, 55, ext=8))
    assert mv(11, [1, 55], [1, 1]).recover_secret(mv(22, [55, 2], [1, 1])) == a0   # :201-220
    with pytest.raises(RLNError, match="No matching nullifier"):
        mv(11, [1, 55], [1, 0]).recover_secret(mv(22, [55, 2], [1, 1]))
    assert sv(11, 55).recover_secret(mv(22, [3, 55], [1, 1])) == a0                # cross-mode, both orders (:248-268)
    assert mv(22, [3, 55], [1, 1]).recover_secret(sv(11, 55)) == a0
    with pytest.raises(RLNError, match="division by zero"):
        sv(11, 55).recover_secret(sv(11, 55))
