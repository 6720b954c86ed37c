"""ORACLE (test infrastructure): deterministic identity generation.

Follows /root/reference/rln/src/protocol/keygen.rs:50-94.  The RNG is rand_chacha 0.3.1 `ChaCha20Rng`
(third-party, pinned in /root/reference/Cargo.lock; RFC 7539 block function, 64-bit counter, stream 0) and
the sampler is ark-ff 0.5.0 `<Fp as UniformRand>::rand` (four u64 limbs, top 2 bits shaved, rejection; the
accepted limbs are taken AS the Montgomery representation).  Pinned by the known answers of
rln/tests/protocol.rs:463-517 and rln/tests/ffi_utils.rs:8-66 (tests/test_oracle_kats.py).
"""
import struct

from .bn254 import R
from .keccak import keccak256
from .poseidon import poseidon

_RINV = pow(1 << 256, -1, R)


def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & 0xFFFFFFFF


def chacha20_block(key, counter):
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(struct.unpack("<8I", key)) + [
        counter & 0xFFFFFFFF, counter >> 32, 0, 0]
    x = st[:]

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF
        x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF
        x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF
        x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF
        x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12), qr(1, 5, 9, 13), qr(2, 6, 10, 14), qr(3, 7, 11, 15)
        qr(0, 5, 10, 15), qr(1, 6, 11, 12), qr(2, 7, 8, 13), qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & 0xFFFFFFFF for a, b in zip(x, st)])


class ChaCha20Rng:
    def __init__(self, seed32):
        self.key, self.counter, self.buf = bytes(seed32), 0, b""

    def fill(self, n):
        while len(self.buf) < n:
            self.buf += chacha20_block(self.key, self.counter)
            self.counter += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out

    def next_fr(self):
        while True:
            raw = int.from_bytes(self.fill(32), "little") & ((1 << 254) - 1)
            if raw < R:
                return raw * _RINV % R


def seeded_keygen(signal: bytes):
    """keygen.rs:50-65 -> (identity_secret, id_commitment)"""
    rng = ChaCha20Rng(keccak256(signal))
    secret = rng.next_fr()
    return secret, poseidon([secret])


def extended_seeded_keygen(signal: bytes):
    """keygen.rs:72-94 -> (trapdoor, nullifier, identity_secret, id_commitment)"""
    rng = ChaCha20Rng(keccak256(signal))
    trapdoor, nullifier = rng.next_fr(), rng.next_fr()
    secret = poseidon([trapdoor, nullifier])
    return trapdoor, nullifier, secret, poseidon([secret])


def compute_id_secret(share1, share2):
    """slashing.rs:12-36"""
    (x1, y1), (x2, y2) = share1, share2
    if (x1 - x2) % R == 0:
        raise ZeroDivisionError("DivisionByZero")
    a1 = (y1 - y2) * pow(x1 - x2, -1, R) % R
    return (y1 - x1 * a1) % R
