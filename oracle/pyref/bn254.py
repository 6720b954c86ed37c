"""ORACLE (test infrastructure, not product): BN254 fields, groups and optimal-ate pairing in
pure-Python integers.

Restates the third-party arithmetic the reference links (ark-bn254 / ark-ec / ark-ff 0.5.0, pinned in
/root/reference/Cargo.lock, sources not vendored).  Conventions are the published BN254 (alt_bn128,
EIP-196/197) ones; they are pinned against the reference by tests/test_oracle_kats.py:
  * the snarkjs proof hard-coded in rln/tests/public.rs:84-135 must verify under the shipped vk,
  * every point in the shipped arkzkey must be on-curve.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr modulus (iden3calc/graph.rs:14-15)
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq modulus
BN_U = 4965661367192848881  # BN parameter u; ate loop count 6u+2
ATE_LOOP = 6 * BN_U + 2

# ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1), tuples (c0, c1)
F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (9, 1)  # non-residue used for the sextic twist


def f2_add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def f2_sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def f2_neg(a):
    return ((-a[0]) % Q, (-a[1]) % Q)


def f2_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    return ((a0 * b0 - a1 * b1) % Q, (a0 * b1 + a1 * b0) % Q)


def f2_sqr(a):
    a0, a1 = a
    return ((a0 + a1) * (a0 - a1) % Q, 2 * a0 * a1 % Q)


def f2_muls(a, s):
    return (a[0] * s % Q, a[1] * s % Q)


def f2_conj(a):
    return (a[0], (-a[1]) % Q)


def f2_inv(a):
    a0, a1 = a
    t = pow((a0 * a0 + a1 * a1) % Q, -1, Q)
    return (a0 * t % Q, (-a1) * t % Q)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


# ---------------------------------------------------------------- curves
G1_B = 3
G2_B = f2_mul((3, 0), f2_inv(XI))  # D-twist: y^2 = x^3 + 3/(9+u)
G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


class _Fq:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % Q)
    sub = staticmethod(lambda a, b: (a - b) % Q)
    mul = staticmethod(lambda a, b: a * b % Q)
    sqr = staticmethod(lambda a: a * a % Q)
    neg = staticmethod(lambda a: (-a) % Q)
    inv = staticmethod(lambda a: pow(a, -1, Q))
    b = G1_B


class _Fq2:
    zero, one = F2_ZERO, F2_ONE
    add = staticmethod(f2_add)
    sub = staticmethod(f2_sub)
    mul = staticmethod(f2_mul)
    sqr = staticmethod(f2_sqr)
    neg = staticmethod(f2_neg)
    inv = staticmethod(f2_inv)
    b = G2_B


class Curve:
    """Short-Weierstrass y^2 = x^3 + b over field F; affine points are (x, y) or None (infinity);
    Jacobian points are (X, Y, Z) with Z == zero for infinity."""

    def __init__(self, F):
        self.F = F

    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sqr(y) == F.add(F.mul(F.sqr(x), x), F.b)

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    # --- affine group law (one inversion each); used by the naive reference paths
    def add(self, P, Qp):
        F = self.F
        if P is None:
            return Qp
        if Qp is None:
            return P
        x1, y1 = P
        x2, y2 = Qp
        if x1 == x2:
            if y1 != y2 or y1 == F.zero:
                return None
            three_x2 = F.mul(F.sqr(x1), (3 if F is _Fq else (3, 0)))
            lam = F.mul(three_x2, F.inv(F.add(y1, y1)))
        else:
            lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.sqr(lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    # --- Jacobian (a = 0)
    def jac_inf(self):
        return (self.F.one, self.F.one, self.F.zero)

    def to_jac(self, P):
        return self.jac_inf() if P is None else (P[0], P[1], self.F.one)

    def jac_dbl(self, P):
        F = self.F
        X, Y, Z = P
        if Z == F.zero:
            return P
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        t = F.sub(F.sub(F.sqr(F.add(X, B)), A), C)
        D = F.add(t, t)
        E = F.add(F.add(A, A), A)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.add(D, D))
        C8 = F.add(C, C)
        C8 = F.add(C8, C8)
        C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        YZ = F.mul(Y, Z)
        Z3 = F.add(YZ, YZ)
        return (X3, Y3, Z3)

    def jac_add_mixed(self, P, A):
        """P Jacobian + A affine (or None)."""
        F = self.F
        if A is None:
            return P
        X1, Y1, Z1 = P
        if Z1 == F.zero:
            return (A[0], A[1], F.one)
        x2, y2 = A
        Z1Z1 = F.sqr(Z1)
        U2 = F.mul(x2, Z1Z1)
        S2 = F.mul(F.mul(y2, Z1), Z1Z1)
        if U2 == X1:
            if S2 == Y1:
                return self.jac_dbl(P)
            return self.jac_inf()
        H = F.sub(U2, X1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        r = F.sub(S2, Y1)
        V = F.mul(X1, HH)
        X3 = F.sub(F.sub(F.sqr(r), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.mul(Y1, HHH))
        Z3 = F.mul(Z1, H)
        return (X3, Y3, Z3)

    def jac_add(self, P, Qp):
        F = self.F
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Qp
        if Z1 == F.zero:
            return Qp
        if Z2 == F.zero:
            return P
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
        S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        if U1 == U2:
            if S1 == S2:
                return self.jac_dbl(P)
            return self.jac_inf()
        H = F.sub(U2, U1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        r = F.sub(S2, S1)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(r), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def to_affine(self, P):
        F = self.F
        X, Y, Z = P
        if Z == F.zero:
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def mul(self, P, k):
        """scalar multiple of an affine point -> affine."""
        acc = self.jac_inf()
        if P is None or k == 0:
            return None
        for bit in bin(k)[2:]:
            acc = self.jac_dbl(acc)
            if bit == "1":
                acc = self.jac_add_mixed(acc, P)
        return self.to_affine(acc)

    def msm_naive(self, points, scalars):
        acc = self.jac_inf()
        for P, k in zip(points, scalars):
            if P is None or k == 0:
                continue
            acc = self.jac_add(acc, self.to_jac(self.mul(P, k)))
        return self.to_affine(acc)

    def msm(self, points, scalars, c=8):
        """Windowed Pippenger (the published algorithm ark-ec's VariableBaseMSM::msm_bigint follows;
        call sites rln/src/partial_proof.rs:103,256).  The result is a group element, so any correct
        bucket method yields the same affine coordinates."""
        pts = [(P, k) for P, k in zip(points, scalars) if P is not None and k != 0]
        if not pts:
            return None
        nwin = (254 + c - 1) // c
        total = self.jac_inf()
        for w in reversed(range(nwin)):
            for _ in range(c):
                total = self.jac_dbl(total)
            buckets = [None] * ((1 << c) - 1)
            for P, k in pts:
                d = (k >> (w * c)) & ((1 << c) - 1)
                if d:
                    b = buckets[d - 1]
                    buckets[d - 1] = self.to_jac(P) if b is None else self.jac_add_mixed(b, P)
            run = self.jac_inf()
            wsum = self.jac_inf()
            for b in reversed(buckets):
                if b is not None:
                    run = self.jac_add(run, b)
                wsum = self.jac_add(wsum, run)
            total = self.jac_add(total, wsum)
        return self.to_affine(total)


G1 = Curve(_Fq)
G2 = Curve(_Fq2)

# ---------------------------------------------------------------- Fq12 = Fq2[w]/(w^6 - xi): lists of 6 Fq2


def f12_one():
    return [F2_ONE] + [F2_ZERO] * 5


def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        ai = a[i]
        if ai == F2_ZERO:
            continue
        for j in range(6):
            bj = b[j]
            if bj == F2_ZERO:
                continue
            t[i + j] = f2_add(t[i + j], f2_mul(ai, bj))
    for k in range(10, 5, -1):
        t[k - 6] = f2_add(t[k - 6], f2_mul(t[k], XI))
    return t[:6]


def f12_pow(a, e):
    r = f12_one()
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_mul(a, a)
        e >>= 1
    return r


_GAMMA12 = f2_pow(XI, (Q - 1) // 3)   # xi^((q-1)/3): x-coordinate Frobenius factor on the twist
_GAMMA13 = f2_pow(XI, (Q - 1) // 2)   # xi^((q-1)/2): y-coordinate factor
_GAMMA22 = f2_pow(XI, (Q * Q - 1) // 3)
_GAMMA23 = f2_pow(XI, (Q * Q - 1) // 2)


def _line(T, Qp, P):
    """Line through twist points T,Qp (affine Fq2) evaluated at P in G1 after the untwist
    (x',y') -> (x' w^2, y' w^3).  Returns (sparse Fq12, T+Qp)."""
    xt, yt = T
    xq, yq = Qp
    if xt == xq and yt == yq:
        lam = f2_mul(f2_muls(f2_sqr(xt), 3), f2_inv(f2_add(yt, yt)))
    else:
        lam = f2_mul(f2_sub(yq, yt), f2_inv(f2_sub(xq, xt)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), xt), xq)
    y3 = f2_sub(f2_mul(lam, f2_sub(xt, x3)), yt)
    xp, yp = P
    l = [F2_ZERO] * 6
    l[0] = (yp % Q, 0)
    l[1] = f2_neg(f2_muls(lam, xp))
    l[3] = f2_sub(f2_mul(lam, xt), yt)
    return l, (x3, y3)


def miller_loop(P, Qp):
    """Optimal ate Miller loop f_{6u+2,Q}(P) * l_{[6u+2]Q,pi(Q)} * l_{.., -pi^2(Q)} (no final exp)."""
    if P is None or Qp is None:
        return f12_one()
    f = f12_one()
    T = Qp
    for bit in bin(ATE_LOOP)[3:]:
        l, T2 = _line(T, T, P)
        f = f12_mul(f12_mul(f, f), l)
        T = T2
        if bit == "1":
            l, T = _line(T, Qp, P)
            f = f12_mul(f, l)
    Q1 = (f2_mul(f2_conj(Qp[0]), _GAMMA12), f2_mul(f2_conj(Qp[1]), _GAMMA13))
    Q2 = (f2_mul(Qp[0], _GAMMA22), f2_neg(f2_mul(Qp[1], _GAMMA23)))  # -pi^2(Q)
    l, T = _line(T, Q1, P)
    f = f12_mul(f, l)
    l, T = _line(T, Q2, P)
    f = f12_mul(f, l)
    return f


_FINAL_EXP = (Q ** 12 - 1) // R


def final_exp(f):
    return f12_pow(f, _FINAL_EXP)


def pairing(P, Qp):
    return final_exp(miller_loop(P, Qp))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 with one shared final exponentiation."""
    f = f12_one()
    for P, Qp in pairs:
        f = f12_mul(f, miller_loop(P, Qp))
    return final_exp(f) == f12_one()
