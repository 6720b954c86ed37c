"""ORACLE (test infrastructure): Poseidon over BN254 Fr exactly as the reference computes it.

Follows /root/reference/utils/src/poseidon/poseidon_hash.rs:97-135 (hash), :63-95 (ark / sbox / mix_2)
and poseidon_constants.rs:15-261 (Grain-LFSR round constants + Cauchy MDS), with the parameter table of
/root/reference/rln/src/hashers.rs:14-23.  Pinned by the KATs in utils/tests/poseidon_hash_test.rs:21-130
(tests/test_oracle_kats.py).
"""
from functools import lru_cache

from .bn254 import R

# (t, R_F, R_P, skip_matrices) -- rln/src/hashers.rs:14-23
ROUND_PARAMS = [(2, 8, 56, 0), (3, 8, 57, 0), (4, 8, 56, 0), (5, 8, 60, 0),
                (6, 8, 60, 0), (7, 8, 63, 0), (8, 8, 64, 0), (9, 8, 63, 0)]


class _Grain:
    """poseidon_constants.rs:15-205"""

    def __init__(self, prime_bits, t, rf, rp):
        st = [0] * 80
        st[1] = 1  # field
        # st[2..5]: sbox = x^alpha (not inverse) -> zeros

        def put(lo, hi, v):
            for i in range(hi, lo - 1, -1):
                st[i] = v & 1
                v >>= 1
        put(6, 17, prime_bits)
        put(18, 29, t)
        put(30, 39, rf)
        put(40, 49, rp)
        for i in range(50, 80):
            st[i] = 1
        self.st, self.head, self.n = st, 0, prime_bits
        for _ in range(160):
            self._update()

    def _update(self):
        s, h = self.st, self.head
        b = s[(h + 62) % 80] ^ s[(h + 51) % 80] ^ s[(h + 38) % 80] ^ s[(h + 23) % 80] ^ s[(h + 13) % 80] ^ s[h]
        s[h] = b
        self.head = (h + 1) % 80
        return b

    def _bits_value(self):
        # get_bits(n) yields MSB first (after the `reverse()` + LE packing in the reference)
        v = 0
        for _ in range(self.n):
            b = self._update()
            while not b:
                self._update()
                b = self._update()
            v = (v << 1) | self._update()
        return v

    def rejection(self, k):
        out = []
        while len(out) < k:
            v = self._bits_value()
            if v < R:
                out.append(v)
        return out

    def mod_p(self, k):
        return [self._bits_value() % R for _ in range(k)]


@lru_cache(maxsize=None)
def constants(t):
    """(ark list of (R_F+R_P)*t, mds t x t) -- poseidon_constants.rs:207-261"""
    _, rf, rp, skip = next(p for p in ROUND_PARAMS if p[0] == t)
    g = _Grain(254, t, rf, rp)
    ark = []
    for _ in range(rf + rp):
        ark += g.rejection(t)
    for _ in range(skip):
        g.mod_p(2 * t)
    xs = g.mod_p(t)
    ys = g.mod_p(t)
    mds = [[pow((xs[i] + ys[j]) % R, -1, R) for j in range(t)] for i in range(t)]
    return ark, mds, rf, rp


def poseidon(inputs):
    """poseidon_hash.rs:97-135"""
    t = len(inputs) + 1
    ark, mds, rf, rp = constants(t)
    state = [0] + [x % R for x in inputs]
    for i in range(rf + rp):
        state = [(s + ark[i * t + j]) % R for j, s in enumerate(state)]
        if i < rf // 2 or i >= rf // 2 + rp:
            state = [pow(s, 5, R) for s in state]
        else:
            state[0] = pow(state[0], 5, R)
        state = [sum(mds[r][c] * state[c] for c in range(t)) % R for r in range(t)]
    return state[0]
