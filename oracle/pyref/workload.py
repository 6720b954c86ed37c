"""ORACLE (test infrastructure): the seeded synthetic workloads of BASELINE.md §5 / SURVEY.md §8(d).

config 1: the witness of /root/reference/rln/benches/partial_proof.rs:5-38 with the secret fixed to
          hash_to_field_le("test-merkle-proof") (ties it to the tree KAT) and (r, s) = (44, 77).
config 2: `n` independent witnesses from a SplitMix64(0xC0FFEE) stream (random path elements / bits --
          proof generation does not need the path to open a real tree, cf. rln/tests/public.rs:45-75).
"""
from .bn254 import R
from .keccak import hash_to_field_le
from .poseidon import poseidon

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def fr(self):
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % R


def config1_witness():
    from .rln import FullMerkleTree
    secret = hash_to_field_le(b"test-merkle-proof")
    rate_commitment = poseidon([poseidon([secret]), 100])
    tree = FullMerkleTree(20)
    tree.set(3, rate_commitment)
    elems, bits = tree.proof(3)
    x = hash_to_field_le(b"hey hey")
    ext = poseidon([hash_to_field_le(b"test-epoch"), hash_to_field_le(b"test-rln-identifier")])
    w = dict(identity_secret=secret, user_message_limit=100, message_id=1, path_elements=elems,
             identity_path_index=bits, x=x, external_nullifier=ext)
    return w, (44, 77), tree.root()


def config2_witnesses(n, seed=0xC0FFEE, depth=20):
    """Cheap generator (no Poseidon on the host): every field is drawn from the stream."""
    g = SplitMix64(seed)
    ws, rs = [], []
    for i in range(n):
        w = dict(identity_secret=g.fr(), user_message_limit=100, message_id=i % 100,
                 path_elements=[g.fr() for _ in range(depth)],
                 identity_path_index=[g.next() & 1 for _ in range(depth)],
                 x=g.fr(), external_nullifier=g.fr())
        ws.append(w)
        rs.append((g.fr(), g.fr()))
    return ws, rs


# ---------------------------------------------------------------------------------------------- config 5 workload
def _sm64_at(seed, j):
    z = (seed + (j + 1) * 0x9E3779B97F4A7C15) & MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def msm_item(seed, i, mode=0):
    """(k_i, s_i) of the config-5 workload: point i is k_i G, its scalar s_i; 253-bit values from words
    8i..8i+3 / 8i+4..8i+7 of the SplitMix64 stream.  mode bit 0: every scalar is s_0; bit 1: k_i = k_(i mod 4)."""
    j = (i & 3) if mode & 2 else i
    k = sum(_sm64_at(seed, 8 * j + q) << (64 * q) for q in range(4)) & ((1 << 253) - 1)
    j = 0 if mode & 1 else i
    s = sum(_sm64_at(seed, 8 * j + 4 + q) << (64 * q) for q in range(4)) & ((1 << 253) - 1)
    return k, s


def msm_expected(seed, first, n, mode=0):
    """msm_bigint(bases, scalars) on that workload (ark-ec 0.5.0; rln/src/partial_proof.rs:98-104) equals
    (sum k_i s_i mod r) G -- Python ints; 2^18 terms take seconds"""
    from .bn254 import G1, G1_GEN
    acc = 0
    for i in range(first, first + n):
        k, s = msm_item(seed, i, mode)
        acc = (acc + k * s) % R
    return G1.mul(G1_GEN, acc) if acc else None
