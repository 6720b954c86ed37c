"""ORACLE (test infrastructure): snarkjs-style QAP witness map, Groth16 prover assembly and verifier.

* `ntt`/`intt`: radix-2 transforms as ark-poly 0.5.0 Radix2EvaluationDomain defines them (third-party,
  pinned in /root/reference/Cargo.lock; call sites circuit/qap.rs:37-90).  Root: W = 5^((r-1)/2^28).
* `witness_map`: /root/reference/rln/src/circuit/qap.rs:30-98.
* `prove`: /root/reference/rln/src/partial_proof.rs:182-274 with an all-unknown mask (== ark-groth16's
  create_proof_with_reduction_and_matrices, equality shown by rln/tests/partial_proof.rs:110-180).
* `verify`: /root/reference/rln/src/protocol/proof.rs:856-894 (public input order) + the Groth16 equation.
Parity unpinned for proof BYTES (the reference holds no golden proof for any (witness, r, s)); pinned by:
the snarkjs proof of rln/tests/public.rs:84-135 verifying under `verify`, and every `prove` output
verifying (A, B are closed forms; C is then unique).
"""
from .bn254 import R, G1, G2, pairing_product_is_one

TWO_ADICITY = 28
W_2_28 = pow(5, (R - 1) >> TWO_ADICITY, R)


def root_of_unity(n):
    lg = n.bit_length() - 1
    assert 1 << lg == n and lg <= TWO_ADICITY
    return pow(W_2_28, 1 << (TWO_ADICITY - lg), R)


def ntt(a, inverse=False):
    """out[i] = sum_j a[j] w^(ij); inverse scales by n^-1."""
    n = len(a)
    w = root_of_unity(n)
    if inverse:
        w = pow(w, -1, R)
    a = list(a)
    # bit reversal
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    ln = 2
    while ln <= n:
        wl = pow(w, n // ln, R)
        half = ln >> 1
        tw = [1] * half
        for k in range(1, half):
            tw[k] = tw[k - 1] * wl % R
        for s in range(0, n, ln):
            for k in range(half):
                u = a[s + k]
                v = a[s + k + half] * tw[k] % R
                a[s + k] = (u + v) % R
                a[s + k + half] = (u - v) % R
        ln <<= 1
    if inverse:
        ninv = pow(n, -1, R)
        a = [x * ninv % R for x in a]
    return a


def next_pow2(n):
    p = 1
    while p < n:
        p <<= 1
    return p


def matvec(rows, w):
    """evaluate_constraint per row (qap.rs:45-52)"""
    return [sum(c * w[col] for c, col in row) % R for row in rows]


def witness_map(zk, w):
    """qap.rs:30-98 -> h (domain_size entries)"""
    ni, nc = zk.num_instance_variables, zk.num_constraints
    n = next_pow2(nc + ni)
    a = matvec(zk.a, w) + [0] * (n - nc)
    b = matvec(zk.b, w) + [0] * (n - nc)
    a[nc:nc + ni] = [x % R for x in w[:ni]]
    c = [a[i] * b[i] % R if i < nc else 0 for i in range(n)]
    g = root_of_unity(2 * n)

    def coset(v):
        v = ntt(v, inverse=True)
        p = 1
        for i in range(n):
            v[i] = v[i] * p % R
            p = p * g % R
        return ntt(v)

    a, b, c = coset(a), coset(b), coset(c)
    return [(a[i] * b[i] - c[i]) % R for i in range(n)]


def prove(zk, w, r, s, msm_window=8):
    """partial_proof.rs:182-274 with mask = all unknown.  Returns affine (A, B, C)."""
    ni = len(zk.gamma_abc_g1)
    if len(w) != ni + len(zk.l_query):
        raise ValueError("MalformedVerifyingKey")
    h = witness_map(zk, w)
    J = G1.to_jac
    a_msm = G1.msm(zk.a_query[1:], w[1:], msm_window)
    g_a = G1.to_affine(_sum(G1, [zk.alpha_g1, zk.a_query[0], a_msm, G1.mul(zk.delta_g1, r)]))
    if r % R != 0:
        b1_msm = G1.msm(zk.b_g1_query[1:], w[1:], msm_window)
        g1_b = G1.to_affine(_sum(G1, [zk.beta_g1, zk.b_g1_query[0], b1_msm, G1.mul(zk.delta_g1, s)]))
    else:
        g1_b = None
    b2_msm = G2.msm(zk.b_g2_query[1:], w[1:], msm_window)
    g2_b = G2.to_affine(_sum(G2, [zk.beta_g2, zk.b_g2_query[0], b2_msm, G2.mul(zk.delta_g2, s)]))
    l_acc = G1.msm(zk.l_query, w[ni:], msm_window)
    h_acc = G1.msm(zk.h_query, h, msm_window)
    g_c = G1.to_affine(_sum(G1, [G1.mul(g_a, s), G1.mul(g1_b, r), G1.neg(G1.mul(zk.delta_g1, r * s % R)),
                                 l_acc, h_acc]))
    return g_a, g2_b, g_c


def _sum(C, pts):
    acc = C.jac_inf()
    for P in pts:
        acc = C.jac_add_mixed(acc, P)
    return acc


def verify(zk, proof, public_inputs):
    """e(A,B) == e(alpha,beta) e(IC,gamma) e(C,delta); inputs exclude the constant 1."""
    A, B, C = proof
    if len(public_inputs) + 1 != len(zk.gamma_abc_g1):
        raise ValueError("MalformedVerifyingKey")
    if not (G1.on_curve(A) and G2.on_curve(B) and G1.on_curve(C)):
        return False
    ic = G1.to_jac(zk.gamma_abc_g1[0])
    for x, P in zip(public_inputs, zk.gamma_abc_g1[1:]):
        ic = G1.jac_add_mixed(ic, G1.mul(P, x % R))
    ic = G1.to_affine(ic)
    return pairing_product_is_one([
        (A, B), (G1.neg(zk.alpha_g1), zk.beta_g2), (G1.neg(ic), zk.gamma_g2), (G1.neg(C), zk.delta_g2)])


def known_mask(graph, unknown_inputs=("messageId", "selectorUsed", "x", "externalNullifier")):
    """evaluate_partial knownness (iden3calc/graph.rs:274-312) per witness signal for the partial witness of
    protocol/witness.rs:887-937"""
    unk = set()
    for name in unknown_inputs:
        if name in graph.input_mapping:
            off, ln = graph.input_mapping[name]
            unk.update(range(off, off + ln))
    known = []
    for n in graph.nodes:
        k = n[0]
        if k == "Const":
            known.append(True)
        elif k == "Input":
            known.append(n[1] not in unk)
        elif k == "Duo":
            known.append(known[n[2]] and known[n[3]])
        elif k == "Uno":
            known.append(known[n[2]])
        else:
            known.append(known[n[2]] and known[n[3]] and known[n[4]])
    return [known[s] for s in graph.signals]


def prove_partial(zk, w, mask):
    """create_partial_proof_from_assignment (partial_proof.rs:108-179); mask/w indexed by witness signal
    (entry 0 = the constant 1).  -> (pi_a, rho, pi_b, pi_c) affine"""
    ni = len(zk.gamma_abc_g1)
    idx = [i for i in range(1, len(w)) if mask[i]]
    a = G1.msm([zk.a_query[i] for i in idx], [w[i] for i in idx])
    b1 = G1.msm([zk.b_g1_query[i] for i in idx], [w[i] for i in idx])
    b2 = G2.msm([zk.b_g2_query[i] for i in idx], [w[i] for i in idx])
    lidx = [i for i in idx if i >= ni]
    lc = G1.msm([zk.l_query[i - ni] for i in lidx], [w[i] for i in lidx])
    pi_a = G1.to_affine(_sum(G1, [zk.alpha_g1, zk.a_query[0], a]))
    rho = G1.to_affine(_sum(G1, [zk.beta_g1, zk.b_g1_query[0], b1]))
    pi_b = G2.to_affine(_sum(G2, [zk.beta_g2, zk.b_g2_query[0], b2]))
    return pi_a, rho, pi_b, lc


def finish_partial(zk, partial, w, mask, r, s):
    """finish_partial_proof_with_assignment (partial_proof.rs:182-274) -> (A, B, C)"""
    ni = len(zk.gamma_abc_g1)
    pi_a, rho, pi_b, pi_c = partial
    h = witness_map(zk, w)
    idx = [i for i in range(1, len(w)) if not mask[i]]
    a = G1.msm([zk.a_query[i] for i in idx], [w[i] for i in idx])
    b1 = G1.msm([zk.b_g1_query[i] for i in idx], [w[i] for i in idx])
    b2 = G2.msm([zk.b_g2_query[i] for i in idx], [w[i] for i in idx])
    lidx = [i for i in idx if i >= ni]
    l_rem = G1.msm([zk.l_query[i - ni] for i in lidx], [w[i] for i in lidx])
    g_a = G1.to_affine(_sum(G1, [pi_a, a, G1.mul(zk.delta_g1, r)]))
    g1_b = G1.to_affine(_sum(G1, [rho, b1, G1.mul(zk.delta_g1, s)])) if r % R else None
    g2_b = G2.to_affine(_sum(G2, [pi_b, b2, G2.mul(zk.delta_g2, s)]))
    h_acc = G1.msm(zk.h_query, h)
    g_c = G1.to_affine(_sum(G1, [G1.mul(g_a, s), G1.mul(g1_b, r), G1.neg(G1.mul(zk.delta_g1, r * s % R)),
                                 pi_c, l_rem, h_acc]))
    return g_a, g2_b, g_c
