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
    hdrs = [os.path.join(ROOT, "zerokit_amd", "csrc", h) for h in ("field.h", "curve.h", "pairing.h", "glv.h",
                                                                     "glv_constants.h", "modinv30.h")]
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


def test_inversion_by_division_steps(L):
    """Fp::inv (modinv30.h: batched division steps) against Python's pow and against the bit-by-bit binary Euclid it
    replaced, both fields: structured values (powers of two, their neighbours, p - small, values whose Montgomery image
    is small) and 3 000 random ones"""
    rnd = random.Random(30)
    for field, mod in ((0, R), (1, Q)):
        rinv = pow(1 << 256, -1, mod)
        xs = [1, 2, 3, mod - 1, mod - 2, (mod + 1) // 2, (mod - 1) // 2]
        xs += [(1 << k) % mod for k in range(1, 256, 7)] + [((1 << k) - 1) % mod for k in range(2, 256, 11)]
        xs += [(mod - (1 << k)) % mod for k in range(0, 254, 13)]
        xs += [(k * rinv) % mod for k in (1, 2, 3, (1 << 30) - 1, 1 << 30, (1 << 60) + 1, mod - 1)]   # stored integer = k
        xs += [rnd.randrange(1, mod) for _ in range(3000)]
        out, out2 = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        for x in xs:
            L.hm_fp_op(field, 3, b(x), b(0), out)
            assert i(out.raw) == pow(x, -1, mod), (field, x)
        for x in xs[:400]:
            L.hm_fp_op(field, 3, b(x), b(0), out)
            L.hm_fp_op(field, 6, b(x), b(0), out2)
            assert out.raw == out2.raw, (field, x)
        L.hm_fp_op(field, 3, b(0), b(0), out)
        assert i(out.raw) == 0


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
    inverse through Fq6; result in the cyclotomic subgroup; the Karatsuba tower product, the complex squaring and the
    sparse line product == the schoolbook product"""
    for seed in (1, 12345, 99991):
        assert L.hm_final_exp_check(seed) == 31


def _glv_constants():
    import re
    txt = open(os.path.join(ROOT, "zerokit_amd", "csrc", "glv_constants.h")).read()
    out = {}
    for name, body in re.findall(r"(\w+)\[\d+\]\s*=\s*\{([^}]*)\}", txt):
        out[name] = sum(int(w.strip().rstrip("u"), 16) << (32 * k) for k, w in enumerate(body.split(",")))
    return out


def test_glv_split_and_endomorphism(L):
    """k = +-k1 + lambda (+-k2) mod r with both halves below 2^126 (what lets the comb tables stop at 127 bits), and
    (beta x, y) = [lambda](x, y) on G1 and G2 for the committed constants."""
    c = _glv_constants()
    lam = c["LAMBDA"]
    assert (lam * lam + lam + 1) % R == 0 and pow(c["BETA_G1"], 3, Q) == 1 and pow(c["BETA_G2"], 3, Q) == 1
    assert c["A1"] * c["B2"] + c["A2"] * c["B1ABS"] == R
    rnd = random.Random(11)
    worst = 0
    ks = [0, 1, 2, R - 1, R - 2, R // 2, (R + 1) // 2, lam, R - lam, lam - 1, c["A2"], c["B1ABS"], R - c["A2"]]
    ks += [rnd.randrange(R) for _ in range(20000)]
    ks += [(rnd.randrange(1 << 40) * c["B1ABS"] + rnd.randrange(1 << 20)) % R for _ in range(2000)]  # near rounding ties
    for k in ks:
        out = ctypes.create_string_buffer(34)
        L.hm_glv_split(b(k), out)
        k1, n1 = int.from_bytes(out.raw[:16], "little"), out.raw[16]
        k2, n2 = int.from_bytes(out.raw[17:33], "little"), out.raw[33]
        assert ((-k1 if n1 else k1) + lam * (-k2 if n2 else k2) - k) % R == 0, k
        worst = max(worst, k1, k2)
    assert worst < 1 << 126
    out = ctypes.create_string_buffer(64)
    for kk in (1, 5, 0xDEADBEEF):
        P = G1.mul(G1_GEN, kk)
        L.hm_glv_phi_g1(g1b(P), out)
        assert g1i(out.raw) == G1.mul(P, lam)
    out = ctypes.create_string_buffer(128)
    for kk in (1, 7, 0xC0FFEE):
        T = G2.mul(G2_GEN, kk)
        L.hm_glv_phi_g2(g2b(T), out)
        assert g2i(out.raw) == G2.mul(T, lam)


def test_fq29_column_bounds_of_every_call_site():
    """fq29.h accumulates 64-bit columns without carry capture; since round 2 most reductions run "wide" rounds (a
    32-bit digit, no mask / 64-bit shift), which raises a column by up to 2^32 * sum(p[j]).  tools/check_fq29_bounds.py
    replays every call site (G1 / G2 mixed addition and doubling, Fq2 products, Poseidon rounds, conversions) with all
    limbs and digits at their maxima and fails on any column that reaches 2^64."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_fq29_bounds", os.path.join(ROOT, "tools", "check_fq29_bounds.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.main(verbose=False)
    assert len(rows) >= 25 and all(r[3] < 16.0 for r in rows)
    assert max(r[3] for r in rows) > 15.0   # the G2 Y3 dot products sit at 15.53: the check is not vacuous
    # the constants in the generated header are the ones the replay used
    hdr = open(os.path.join(ROOT, "zerokit_amd", "csrc", "fq29_constants.h")).read()
    f = mod.Field(mod.Q)
    assert "INV32 = 0x%08xu" % ((-pow(f.P[0], -1, 1 << 32)) % (1 << 32)) in hdr
    assert ("{" + ", ".join("0x%08xu" % l for l in f.K6) + "}") in hdr


@pytest.fixture(scope="module")
def HV():
    """CPU build of the host-side verifier (zkey.cpp + pairing.h) -- tests/host/hostverify.cpp"""
    so = os.path.join(ROOT, "tests", "host", "libhostverify.so")
    src = os.path.join(ROOT, "tests", "host", "hostverify.cpp")
    csrc = os.path.join(ROOT, "zerokit_amd", "csrc")
    deps = [src] + [os.path.join(csrc, h) for h in ("field.h", "curve.h", "pairing.h", "zkey.cpp", "zkey.h", "common.h")]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I",
                               "/opt/rocm/include", "-I", csrc, src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.hv_verify_us.restype = ctypes.c_double
    z = open(os.path.join(csrc, "..", "resources", "tree_depth_20", "rln_final.arkzkey"), "rb").read()
    assert lib.hv_load_zkey(z, len(z)) == 0
    return lib


def test_host_verifier_on_the_golden_proofs(HV):
    """ffi_verify_rln_proof's arithmetic runs on the host (verification stays on the CPU): arkzkey
    parser, point decompression, subgroup check and the pairing check (Karatsuba tower, sparse line products, Straus
    combination of the public inputs) accept every golden proof and reject it under a changed public input, a changed
    proof byte and a swapped (A, C)."""
    import json
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]
    for c in cases:
        proof = bytes.fromhex(c["proof_compressed"])
        pub = b"".join(b(int(v)) for v in c["public_inputs"])
        assert HV.hv_verify(proof, pub, 5) == 1, c["name"]
        for k in range(5):
            bad = bytearray(pub)
            bad[32 * k] ^= 1
            assert HV.hv_verify(proof, bytes(bad), 5) == 0, (c["name"], k)
        swapped = proof[96:128] + proof[32:96] + proof[0:32]
        assert HV.hv_verify(swapped, pub, 5) == 0
        flipped = bytearray(proof)
        flipped[40] ^= 4
        assert HV.hv_verify(bytes(flipped), pub, 5) in (0,)   # off the curve, outside the subgroup or a failing pairing
        nonc = bytearray(pub)
        nonc[31] = 0xFF                                          # public input >= r
        assert HV.hv_verify(proof, bytes(nonc), 5) == 0
    us = HV.hv_verify_us(bytes.fromhex(cases[0]["proof_compressed"]), b"".join(b(int(v)) for v in cases[0]["public_inputs"]), 5, 5)
    assert 0 < us < 200000


def test_g2_subgroup_check_by_endomorphism_equals_the_definition(HV):
    """g2_in_subgroup tests psi(P) == [6 u^2] P (what ark-ec's BN model does); on points of the twist inside and
    outside the order-r subgroup it must agree with the definition [r] P == O."""
    from oracle.pyref import arkzkey
    from oracle.pyref.bn254 import G2_B, f2_add, f2_mul, f2_sqr

    def g2bytes(P):
        return b(P[0][0]) + b(P[0][1]) + b(P[1][0]) + b(P[1][1])
    rnd = random.Random(99)
    for _ in range(3):
        assert HV.hv_g2_checks(g2bytes(G2.mul(G2_GEN, rnd.randrange(1, R)))) == 7
    found = 0
    while found < 4:
        x = (rnd.randrange(Q), rnd.randrange(Q))
        y = arkzkey._sqrt_fq2(f2_add(f2_mul(f2_sqr(x), x), G2_B))
        if y is None:
            continue
        found += 1
        assert HV.hv_g2_checks(g2bytes((x, y))) == 1   # on the twist, in neither form's subgroup (cofactor > 1)
