"""Synthetic witness streams for bench.py and the size tests (BASELINE configs 2 and 4).

The generator is the seeded one SURVEY 8(d) defines: one SplitMix64(0xC0FFEE) stream, 120 draws per witness (secret,
20 path elements, 20 path bits, x, external nullifier, r, s; field elements = 4 draws reduced mod r), so witness i
starts at draw 120 i -- config 4's shard of GPU g is simply the index range [8192 g, 8192 (g + 1)).  SplitMix64 is a
counter generator (state_k = seed + k * gamma), which is what makes an index range computable without the prefix; the
draws are vectorised with numpy.  oracle/pyref/workload.py restates the same generator sequentially; the CPU suite
checks that the two agree (tests/test_cabi_host.py), the product never imports the oracle.
"""
import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
GAMMA = 0x9E3779B97F4A7C15
DRAWS = 120        # per witness at depth 20: 4 + 80 + 20 + 4 + 4 + 4 + 4


def _draws(seed, first_draw, count):
    """SplitMix64 outputs number first_draw .. first_draw + count (0-based) of the stream seeded with `seed`"""
    with np.errstate(over="ignore"):
        k = np.arange(first_draw + 1, first_draw + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * np.uint64(GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _fr_bytes(limbs4):
    """limbs4: uint64 array [..., 4] (little-endian limbs) -> list of canonical 32-byte LE values mod r"""
    raw = np.ascontiguousarray(limbs4.astype("<u8")).tobytes()
    return [(int.from_bytes(raw[o:o + 32], "little") % R).to_bytes(32, "little") for o in range(0, len(raw), 32)]


def config2_range(first, n, seed=0xC0FFEE, depth=20):
    """witnesses first .. first + n of the config-2 / config-4 stream as (list of witness dicts, list of (r, s))"""
    assert depth == 20
    d = _draws(seed, DRAWS * first, DRAWS * n).reshape(n, DRAWS)
    fr = lambda cols: [int.from_bytes(b, "little") for b in _fr_bytes(cols)]  # noqa: E731
    secret = fr(d[:, 0:4])
    path = fr(d[:, 4:84].reshape(n, 20, 4))
    bits = (d[:, 84:104] & np.uint64(1)).astype(np.uint8)
    x, ext, r, s = fr(d[:, 104:108]), fr(d[:, 108:112]), fr(d[:, 112:116]), fr(d[:, 116:120])
    ws = [dict(identity_secret=secret[i], user_message_limit=100, message_id=(first + i) % 100,
               path_elements=path[20 * i:20 * i + 20], identity_path_index=[int(b) for b in bits[i]], x=x[i],
               external_nullifier=ext[i]) for i in range(n)]
    return ws, list(zip(r, s))


def config2_packed(slots, inputs_size, first, n, seed=0xC0FFEE, depth=20):
    """the same witnesses as the witness-graph inputs buffer (n * inputs_size * 32 bytes, slot 0 = 1,
    iden3calc.rs:122-181) and the (r, s) buffer (n * 64 bytes); slots = {graph signal name: (offset, length)}"""
    assert depth == 20
    d = _draws(seed, DRAWS * first, DRAWS * n).reshape(n, DRAWS)
    buf = np.zeros((n, inputs_size, 32), dtype=np.uint8)
    buf[:, 0, 0] = 1

    def put(name, cols, count):
        off, ln = slots[name]
        assert ln == count
        b = np.frombuffer(b"".join(_fr_bytes(cols)), dtype=np.uint8).reshape(n, count, 32)
        buf[:, off:off + count, :] = b

    put("identitySecret", d[:, 0:4], 1)
    put("pathElements", d[:, 4:84].reshape(n, 20, 4), 20)
    put("x", d[:, 104:108], 1)
    put("externalNullifier", d[:, 108:112], 1)
    off, _ = slots["identityPathIndex"]
    buf[:, off:off + 20, 0] = (d[:, 84:104] & np.uint64(1)).astype(np.uint8)
    off, _ = slots["userMessageLimit"]
    buf[:, off, 0] = 100
    off, _ = slots["messageId"]
    buf[:, off, 0] = ((first + np.arange(n)) % 100).astype(np.uint8)
    rs = np.frombuffer(b"".join(_fr_bytes(d[:, 112:120].reshape(n, 2, 4))), dtype=np.uint8)
    return buf.tobytes(), rs.tobytes()


def tree_update_stream(n, count, seed, tag):
    """[(leaf index, leaf value)] of the tree-mutation workload of bench.py's config 3 (SplitMix64(seed) at position k mod n,
    value tag + k); oracle/c applies the same stream from its own generator (oracle_tree_bench)"""
    M = (1 << 64) - 1
    out = []
    for k in range(count):
        z = (seed + (k + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        out.append(((z ^ (z >> 31)) % n, tag + k))
    return out


def circuit_range(first, n, depth=20, multi=False, seed=0xC1AC0175, max_out=4, limit=100):
    """witnesses first .. first + n for ANY shipped circuit (depth 10 / 20 single, depth 20 multi-message-id), as the
    named inputs of witness.rs:832-881 -> ([{graph signal name: [ints]}], [(r, s)]).  One SplitMix64(seed) stream,
    4 * (depth + 5) + depth + 2 draws per witness in the order secret, path elements, x, external nullifier, r, s
    (4 draws each, reduced mod r), path bits (one draw each), message-id base, selector bits.  Valid by construction
    under RLNWitnessInput::new_single / new_multi (witness.rs:78-180): message ids distinct and below the limit, at
    least one selector set.  Used by the size tests of the other circuits and bench.py's operating points."""
    per = 4 * (depth + 5) + depth + 2
    M = (1 << 64) - 1

    def draw(j):
        z = (seed + (j + 1) * GAMMA) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    named, rs = [], []
    for i in range(first, first + n):
        b = per * i
        fr = lambda k: sum(draw(b + 4 * k + q) << (64 * q) for q in range(4)) % R  # noqa: E731
        secret = fr(0)
        path = [fr(1 + k) for k in range(depth)]
        x, ext, r, s = fr(depth + 1), fr(depth + 2), fr(depth + 3), fr(depth + 4)
        o = b + 4 * (depth + 5)
        bits = [int(draw(o + k) & 1) for k in range(depth)]
        base, selw = draw(o + depth), draw(o + depth + 1)
        w = {"identitySecret": [secret], "userMessageLimit": [limit], "pathElements": path, "identityPathIndex": bits,
             "x": [x], "externalNullifier": [ext]}
        if multi:
            sel = [int((selw >> k) & 1) for k in range(max_out)]
            if not any(sel):
                sel[int(base % max_out)] = 1
            w["messageId"] = [int((base + 7 * k) % limit) for k in range(max_out)]   # 7 k mod 100 distinct for k < 4
            w["selectorUsed"] = sel
        else:
            w["messageId"] = [int(base % limit)]
        named.append(w)
        rs.append((r, s))
    return named, rs
