"""CPU build of the product's __host__ __device__ field / curve code (zerokit_amd/csrc/{field,curve}.h, the
same functions the HIP kernels inline) checked against the Python oracle.  No GPU needed."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle.pyref.bn254 import (G1, G2, G1_GEN, G2_GEN, Q, R, f2_add, f2_inv, f2_mul, f2_neg, f2_sqr, f2_sub)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    so = os.path.join(ROOT, "tests", "host", "libhostmath.so")
    src = os.path.join(ROOT, "tests", "host", "hostmath.cpp")
    hdrs = [os.path.join(ROOT, "zerokit_amd", "csrc", h) for h in ("field.h", "curve.h", "pairing.h")]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I",
                               os.path.join(ROOT, "zerokit_amd", "csrc"), src, "-o", so])
    return ctypes.CDLL(so)


def b(x):
    return x.to_bytes(32, "little")


def i(bs):
    return int.from_bytes(bs, "little")


def g1b(P):
    return b(0) * 2 if P is None else b(P[0]) + b(P[1])


def g1i(bs):
    x, y = i(bs[:32]), i(bs[32:])
    return None if x == 0 and y == 0 else (x, y)


def g2b(P):
    return b(0) * 4 if P is None else b(P[0][0]) + b(P[0][1]) + b(P[1][0]) + b(P[1][1])


def g2i(bs):
    v = [i(bs[32 * k:32 * k + 32]) for k in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def test_prime_fields(L):
    rnd = random.Random(1)
    for field, mod in ((0, R), (1, Q)):
        edge = [(0, 0), (1, mod - 1), (mod - 1, mod - 1), (mod - 1, 1), (0, 5), (2, (mod + 1) // 2)]
        for n in range(300):
            x, y = edge[n] if n < len(edge) else (rnd.randrange(mod), rnd.randrange(mod))
            out = ctypes.create_string_buffer(32)
            want = [(x + y) % mod, (x - y) % mod, x * y % mod, pow(x, -1, mod) if x else 0, (-x) % mod, x * x % mod]
            for op, w in enumerate(want):
                if op == 3 and n > 24:
                    continue
                L.hm_fp_op(field, op, b(x), b(y), out)
                assert i(out.raw) == w, (field, op, x, y)


def test_fq2(L):
    rnd = random.Random(2)
    for _ in range(60):
        x = (rnd.randrange(Q), rnd.randrange(Q))
        y = (rnd.randrange(Q), rnd.randrange(Q))
        out = ctypes.create_string_buffer(64)
        want = [f2_add(x, y), f2_sub(x, y), f2_mul(x, y), f2_inv(x), f2_neg(x), f2_sqr(x)]
        for op, w in enumerate(want):
            L.hm_fq2_op(op, b(x[0]) + b(x[1]), b(y[0]) + b(y[1]), out)
            assert (i(out.raw[:32]), i(out.raw[32:])) == w, op


def test_g1_group_law_incl_special_cases(L):
    rnd = random.Random(3)
    out = ctypes.create_string_buffer(64)
    for _ in range(6):
        k1, k2 = rnd.randrange(R), rnd.randrange(R)
        P, S = G1.mul(G1_GEN, k1), G1.mul(G1_GEN, k2)
        L.hm_g1_add(g1b(P), g1b(S), out)
        assert g1i(out.raw) == G1.add(P, S)
        L.hm_g1_add(g1b(P), g1b(P), out)                      # doubling branch of madd
        assert g1i(out.raw) == G1.add(P, P)
        L.hm_g1_add(g1b(P), g1b(G1.neg(P)), out)              # cancellation
        assert g1i(out.raw) is None
        L.hm_g1_add(g1b(None), g1b(P), out)
        assert g1i(out.raw) == P
        L.hm_g1_add(g1b(P), g1b(None), out)
        assert g1i(out.raw) == P
        L.hm_g1_add_full(g1b(P), g1b(S), out)                 # add-2008-s
        assert g1i(out.raw) == G1.add(G1.mul(P, 2), G1.mul(S, 4))
        half = G1.mul(P, (R + 1) // 2)                        # 2P + 4(P/2) = 4P: equal-operand branch
        L.hm_g1_add_full(g1b(P), g1b(half), out)
        assert g1i(out.raw) == G1.mul(P, 4)
        L.hm_g1_mul(g1b(P), b(k2), out)
        assert g1i(out.raw) == G1.mul(P, k2)
    L.hm_g1_mul(g1b(G1_GEN), b(R), out)
    assert g1i(out.raw) is None
    L.hm_g1_mul(g1b(G1_GEN), b(0), out)
    assert g1i(out.raw) is None


def test_g2_group_law(L):
    rnd = random.Random(4)
    out = ctypes.create_string_buffer(128)
    for _ in range(3):
        k1, k2 = rnd.randrange(R), rnd.randrange(R)
        P, S = G2.mul(G2_GEN, k1), G2.mul(G2_GEN, k2)
        L.hm_g2_add(g2b(P), g2b(S), out)
        assert g2i(out.raw) == G2.add(P, S)
        L.hm_g2_add(g2b(P), g2b(P), out)
        assert g2i(out.raw) == G2.add(P, P)
        L.hm_g2_mul(g2b(P), b(k2), out)
        assert g2i(out.raw) == G2.mul(P, k2)


def test_structured_final_exponentiation_equals_definition(L):
    """pairing.h: (q^6-1)(q^2+1) + the u-chain for the hard part == f^((q^12-1)/r) by square-and-multiply; Fq12
    inverse through Fq6; result in the cyclotomic subgroup"""
    for seed in (1, 12345, 99991):
        assert L.hm_final_exp_check(seed) == 7
