"""rln::hashers (/root/reference/rln/src/hashers.rs:32-93) over the C ABI.  Field elements are Python ints."""
import ctypes as C

from ._native import check, lib

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _b(x: int) -> bytes:
    return int(x).to_bytes(32, "little")


def poseidon_hash(inputs):
    """poseidon_hash (hashers.rs:32-36): one hash of 1..8 field elements, on the GPU."""
    return poseidon_hash_batch([list(inputs)])[0]


def poseidon_hash_pair(a, b):
    """hashers.rs:49-54"""
    return poseidon_hash([a, b])


def poseidon_hash_batch(rows):
    """n independent hashes of equal arity in one launch."""
    if not rows:
        return []
    arity = len(rows[0])
    if arity == 0:
        raise ValueError("Empty input provided")   # PoseidonError::EmptyInput
    buf = b"".join(_b(v) for row in rows for v in row)
    out = C.create_string_buffer(32 * len(rows))
    check(lib().rlnamd_poseidon_hash(buf, len(rows), arity, out))
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(len(rows))]


def hash_to_field_le(signal: bytes) -> int:
    """hashers.rs:73-81"""
    out = C.create_string_buffer(32)
    check(lib().rlnamd_hash_to_field_le(signal, len(signal), out))
    return int.from_bytes(out.raw, "little")


def hash_to_field_be(signal: bytes) -> int:
    """hashers.rs:84-93"""
    out = C.create_string_buffer(32)
    check(lib().rlnamd_hash_to_field_be(signal, len(signal), out))
    return int.from_bytes(out.raw, "little")
