#!/usr/bin/env python3
"""Column-bound proof for zerokit_amd/csrc/fq29.h: replays the interleaved Montgomery product (redc_dot / sqr_add) of
every call site in fq29.h, walk29.h, poseidon.h and the NTT's mul_mont with ALL operand limbs at the maximum of their
representation class and the reduction digit at its maximum (2^32 - 1 in a wide round, 2^29 - 1 in a masked one), and
asserts that no 64-bit column accumulator reaches 2^64.  Every term of a column is non-negative and monotone in each
operand limb and in the digit, so the all-maximum replay bounds every real execution.  Run by tests/test_host_math.py.

Classes (limb vectors of 9 entries):
  N(B)      normalised value < B q: limbs 0..7 <= 2^29 - 1, limb 8 <= (B q) >> 232
  K - b     lazy difference, b normalised: limb j <= K[j]
  x + y     sums of the above, limb by limb
"""
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
B29 = 29
M = (1 << B29) - 1
LIMIT = 1 << 64


def limbs(v):
    return [(v >> (B29 * j)) & M for j in range(8)] + [v >> (B29 * 8)]


def biased(v, lo):
    out, rem = [], v
    for _ in range(8):
        d = rem & M
        l = lo + ((d - lo) % (1 << B29))
        out.append(l)
        rem = (rem - l) >> B29
    out.append(rem)
    return out


class Field:
    def __init__(self, p):
        self.p = p
        self.P = limbs(p)
        self.K2, self.K4, self.K6, self.K8 = (biased(k * p, 1 << B29) for k in (2, 4, 6, 8))
        self.K4T = biased(4 * p, 3 << B29)

    def N(self, bq=10):
        return [M] * 8 + [(bq * self.p) >> 232]


def add(a, b):
    return [x + y for x, y in zip(a, b)]


def scale(a, k):
    return [k * x for x in a]


def replay(f, prods, wide, addend=None, square=False):
    """prods: list of (a, b) limb-maximum vectors; returns the maximum column value seen"""
    t = [0] * 10
    worst = 0

    def chk():
        nonlocal worst
        worst = max(worst, max(t))
        assert max(t) < LIMIT, "column overflow: %.3f x 2^60" % (max(t) / 2**60)

    for i in range(9):
        for a, b in prods:
            if square:   # row i of sqr_add: a_i^2 into t[i], (2 a_i) a_l into t[l]
                assert 2 * a[i] < 1 << 32
                t[i] += a[i] * a[i]
                for l in range(i + 1, 9):
                    t[l] += 2 * a[i] * a[l]
            else:
                for j in range(9):
                    assert a[j] < 1 << 32 and b[i] < 1 << 32
                    t[j] += a[j] * b[i]
            chk()
        w = wide and i < 8
        m = (1 << 32) - 1 if w else M
        for j in range(9):
            t[j] += m * f.P[j]
        chk()
        carry = t[0] >> 29
        t = t[1:] + [0]
        t[0] += carry
        chk()
    if addend:
        for j in range(9):
            assert addend[j] < 1 << 32
            t[j] += addend[j]
    for j in range(8):
        t[j + 1] += t[j] >> 29
        chk()
    assert t[8] < 1 << 32, "top limb does not fit"
    return worst


def main(verbose=True):
    rows = []
    for name, f in (("Fq", Field(Q)), ("Fr", Field(R))):
        N = f.N()
        lazy2 = scale(N, 2)                       # sums of two normalised values, 2 x value
        sites = []
        if name == "Fq":
            D = add(N, f.K6)                      # Q + K6 - X3, un-normalised
            sites += [
                ("g1.U2  mul_add(px, ZZ, K6 - X)", [(N, N)], True, f.K6, False),
                ("g1.S2  mul_add(K2 - py, ZZZ, K4 - Y)", [(f.K2, N)], True, f.K4, False),
                ("g1.PP  sqr(P)", [(N, N)], True, None, True),
                ("g1.ZZ3/Q/PPP/ZZZ3  mul(N, N)", [(N, N)], True, None, False),
                ("g1.X3  sqr_add(R, K4T - PPP - 2Q)", [(N, N)], True, f.K4T, True),
                ("g1.Y3  dot2(R, Q + K6 - X3, K4 - Y, PPP)", [(N, D), (f.K4, N)], True, None, False),
                ("g1.dbl sqr(2 py)", [(lazy2, lazy2)], True, None, True),
                ("g1.dbl mul(2 py, V)", [(lazy2, N)], True, None, False),
                ("g1.dbl dot2(M, D, K4 - py, W)", [(N, N), (f.K4, N)], True, None, False),
                ("fq2.mul c0 dot2_add(a0, b0, K8 - a1, b1, K - x)", [(N, N), (f.K8, N)], True, f.K6, False),
                ("fq2.mul c1 dot2_add(a0, b1, a1, b0, K - x)", [(N, N), (N, N)], True, f.K6, False),
                ("fq2.sqr mul_add(a0 + a1, d, K4T - ..)", [(lazy2, N)], True, f.K4T, False),
                ("g2.Y3  dot4 (R0 D0, K8-R1 D1, K4-Y0 P0, Y1 P1)", [(N, N), (f.K8, N), (f.K4, N), (N, N)], True, None, False),
                ("g2.Y3  dot4 (R0 D1, R1 D0, K4-Y0 P1, K4-Y1 P0)", [(N, N), (N, N), (f.K4, N), (f.K4, N)], True, None, False),
                ("g2.dbl dot4 (M0 D0, K8-M1 D1, W0 K4-y0, W1 y1)", [(N, N), (f.K8, N), (N, f.K4), (N, N)], True, None, False),
                ("g2.dbl dot4 (M0 D1, M1 D0, W0 K4-y1, W1 K4-y0)", [(N, N), (N, N), (N, f.K4), (N, f.K4)], True, None, False),
            ]
        else:
            sites += [
                # partial-round lanes 1.. stay normalised but grow to < 72 r (poseidon.h): N(80); "+ ark" adds < r
                ("poseidon sqr(st + ark)", [(add(f.N(80), f.N(1)), None)], True, None, True),
                ("poseidon mul(x4, st + ark)", [(N, add(f.N(80), f.N(1)))], True, None, False),
                ("poseidon t=2 dot2(mds, st), st[1] lazy", [(N, N), (N, lazy2)], True, None, False),
                ("poseidon full round dot3(mds, st)", [(N, N)] * 3, True, None, False),
                ("poseidon full round dot4(mds, st)", [(N, N)] * 4, True, None, False),
                ("poseidon dense partial round dot3 masked, two lazy", [(N, N), (N, lazy2), (N, lazy2)], False, None, False),
                ("poseidon row (sparse partial round) dotn<4>", [(N, f.N(80))] * 4, True, None, False),
                ("poseidon dotn<5> masked", [(N, f.N(80))] * 5, False, None, False),
                ("poseidon mul_add(u_i, st[0], st[i])  (st[i] < 72 r)", [(N, N)], True, f.N(80), False),
                # four lanes per hash (poseidon_hash4_lanes): x = s_0 + k with s_0 < 4 r; p_i = row0[i] s_i < 1.5 r
                ("poseidon 4-lane mul(x, x | u_i | row0[0]), x = s_0 + k", [(add(N, f.N(1)), add(N, f.N(1)))], True, None, False),
                ("poseidon 4-lane mul(row0[i], s_i)  (s_i < 72 r)", [(N, f.N(80))], True, None, False),
                ("poseidon 4-lane s_0' = mul_add(b, x^4, p_1 + p_2)", [(N, N)], True, lazy2, False),
                ("witness interpreter a * b + c (mul_add, every value below 7.5 r)", [(N, N)], True, N, False),
            ]
        sites += [
            ("%s from_fq / to_fq / mul_mont  mul(N, const)" % name, [(N, limbs(f.p - 1))], True, None, False),
            ("%s mul(lazy, lazy)" % name, [(lazy2, lazy2)], True, None, False),
        ]
        for label, prods, wide, addend, square in sites:
            w = replay(f, prods, wide, addend, square)
            rows.append((name, label, "wide" if wide else "masked", w / 2**60))
    if verbose:
        for r in rows:
            print("%-3s %-62s %-6s max column %6.3f x 2^60 (limit 16)" % r)
    return rows


if __name__ == "__main__":
    main()
