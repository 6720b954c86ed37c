"""ORACLE (test infrastructure): parser for the reference's `.arkzkey` (arkworks uncompressed, unchecked
ProvingKey<Bn254> followed by SerializableConstraintMatrices).

Follows /root/reference/rln/src/circuit/mod.rs:256-305 (struct order) and the ark-serialize 0.5.0
derive layout (field order of ark_groth16::{VerifyingKey, ProvingKey}); byte-exact consumption of the
shipped files is asserted in tests/test_oracle_kats.py.
"""
import struct
from dataclasses import dataclass, field

from .bn254 import Q


def _u64(b, o):
    return struct.unpack_from("<Q", b, o)[0], o + 8


def _fq(b, o):
    return int.from_bytes(b[o:o + 32], "little"), o + 32


def _g1(b, o):
    x, o = _fq(b, o)
    yraw = b[o:o + 32]
    o += 32
    flags = yraw[31] & 0xC0
    if flags & 0x40:
        return None, o
    y = int.from_bytes(yraw[:31] + bytes([yraw[31] & 0x3F]), "little")
    return (x, y), o


def _g2(b, o):
    x0, o = _fq(b, o)
    x1, o = _fq(b, o)
    y0, o = _fq(b, o)
    yraw = b[o:o + 32]
    o += 32
    if yraw[31] & 0x40:
        return None, o
    y1 = int.from_bytes(yraw[:31] + bytes([yraw[31] & 0x3F]), "little")
    return ((x0, x1), (y0, y1)), o


def _vec(b, o, rd):
    n, o = _u64(b, o)
    out = []
    for _ in range(n):
        v, o = rd(b, o)
        out.append(v)
    return out, o


@dataclass
class Zkey:
    alpha_g1: tuple = None
    beta_g2: tuple = None
    gamma_g2: tuple = None
    delta_g2: tuple = None
    gamma_abc_g1: list = field(default_factory=list)
    beta_g1: tuple = None
    delta_g1: tuple = None
    a_query: list = field(default_factory=list)
    b_g1_query: list = field(default_factory=list)
    b_g2_query: list = field(default_factory=list)
    h_query: list = field(default_factory=list)
    l_query: list = field(default_factory=list)
    num_instance_variables: int = 0
    num_witness_variables: int = 0
    num_constraints: int = 0
    a_nnz: int = 0
    b_nnz: int = 0
    c_nnz: int = 0
    a: list = field(default_factory=list)  # rows of [(coeff, col)]
    b: list = field(default_factory=list)
    c: list = field(default_factory=list)
    consumed: int = 0


def parse(data: bytes) -> Zkey:
    if not data:
        raise ValueError("empty arkzkey")
    z = Zkey()
    o = 0
    z.alpha_g1, o = _g1(data, o)
    z.beta_g2, o = _g2(data, o)
    z.gamma_g2, o = _g2(data, o)
    z.delta_g2, o = _g2(data, o)
    z.gamma_abc_g1, o = _vec(data, o, _g1)
    z.beta_g1, o = _g1(data, o)
    z.delta_g1, o = _g1(data, o)
    z.a_query, o = _vec(data, o, _g1)
    z.b_g1_query, o = _vec(data, o, _g1)
    z.b_g2_query, o = _vec(data, o, _g2)
    z.h_query, o = _vec(data, o, _g1)
    z.l_query, o = _vec(data, o, _g1)
    (z.num_instance_variables, o) = _u64(data, o)
    (z.num_witness_variables, o) = _u64(data, o)
    (z.num_constraints, o) = _u64(data, o)
    (z.a_nnz, o) = _u64(data, o)
    (z.b_nnz, o) = _u64(data, o)
    (z.c_nnz, o) = _u64(data, o)

    def entry(b, o):
        v, o = _fq(b, o)
        c, o = _u64(b, o)
        return (v, c), o

    def row(b, o):
        return _vec(b, o, entry)

    z.a, o = _vec(data, o, row)
    z.b, o = _vec(data, o, row)
    z.c, o = _vec(data, o, row)
    z.consumed = o
    return z


# ----------------------------------------------------------------- arkworks point (de)compression
def _fq_is_neg(y):
    return y > (Q - 1) // 2


def _fq2_is_neg(y):
    """lexicographic, c1 first"""
    c0, c1 = y
    n0, n1 = (-c0) % Q, (-c1) % Q
    return (c1, c0) > (n1, n0)


def g1_compress(P) -> bytes:
    if P is None:
        return bytes(31) + bytes([0x40])
    b = bytearray(P[0].to_bytes(32, "little"))
    if _fq_is_neg(P[1]):
        b[31] |= 0x80
    return bytes(b)


def g2_compress(P) -> bytes:
    if P is None:
        return bytes(63) + bytes([0x40])
    b = bytearray(P[0][0].to_bytes(32, "little") + P[0][1].to_bytes(32, "little"))
    if _fq2_is_neg(P[1]):
        b[63] |= 0x80
    return bytes(b)


def proof_compress(A, B, C) -> bytes:
    """ark-serialize compressed Proof{a,b,c} = 128 bytes (COMPRESS_PROOF_SIZE, circuit/mod.rs:82)."""
    return g1_compress(A) + g2_compress(B) + g1_compress(C)


def _sqrt_fq(a):
    r = pow(a, (Q + 1) // 4, Q)
    return r if r * r % Q == a % Q else None


def _sqrt_fq2(a):
    from .bn254 import f2_pow, f2_sqr, f2_mul, F2_ONE
    # q^2 = 9 mod 16 is awkward; use the norm method: sqrt(a0 + a1 u)
    a0, a1 = a
    if a1 == 0:
        r = _sqrt_fq(a0)
        if r is not None:
            return (r, 0)
        r = _sqrt_fq((-a0) % Q)
        return (0, r)
    n = _sqrt_fq((a0 * a0 + a1 * a1) % Q)
    if n is None:
        return None
    inv2 = pow(2, -1, Q)
    for nn in (n, (-n) % Q):
        t = (a0 + nn) * inv2 % Q
        x0 = _sqrt_fq(t)
        if x0 is None or x0 == 0:
            continue
        x1 = a1 * pow(2 * x0, -1, Q) % Q
        if f2_sqr((x0, x1)) == (a0 % Q, a1 % Q):
            return (x0, x1)
    return None


def g1_decompress(b: bytes):
    if b[31] & 0x40:
        return None
    x = int.from_bytes(b[:31] + bytes([b[31] & 0x3F]), "little")
    y = _sqrt_fq((x * x * x + 3) % Q)
    if y is None:
        raise ValueError("not on curve")
    if _fq_is_neg(y) != bool(b[31] & 0x80):
        y = (-y) % Q
    return (x, y)


def g2_decompress(b: bytes):
    from .bn254 import G2_B, f2_add, f2_mul, f2_sqr, f2_neg
    if b[63] & 0x40:
        return None
    x0 = int.from_bytes(b[:32], "little")
    x1 = int.from_bytes(b[32:63] + bytes([b[63] & 0x3F]), "little")
    x = (x0, x1)
    y = _sqrt_fq2(f2_add(f2_mul(f2_sqr(x), x), G2_B))
    if y is None:
        raise ValueError("not on curve")
    if _fq2_is_neg(y) != bool(b[63] & 0x80):
        y = f2_neg(y)
    return (x, y)


def proof_decompress(b: bytes):
    return g1_decompress(b[:32]), g2_decompress(b[32:96]), g1_decompress(b[96:128])
