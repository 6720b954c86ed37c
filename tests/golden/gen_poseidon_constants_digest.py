#!/usr/bin/env python3
"""Reads the hard-coded Poseidon parameters of the reference's own test (utils/tests/poseidon_constants.rs:42-3490:
`c_str` = round constants, `m_str` = MDS matrices for t = 2..9, taken there from the Poseidon reference implementation)
and writes one SHA-256 per t over the 32-byte little-endian values (round constants in order, then the matrix row by
row) to tests/golden/poseidon_constants_digest.json.  Run in the build container only (needs /root/reference)."""
import hashlib
import json
import os
import re

SRC = "/root/reference/utils/tests/poseidon_constants.rs"
HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = [(2, 8, 56), (3, 8, 57), (4, 8, 56), (5, 8, 60), (6, 8, 60), (7, 8, 63), (8, 8, 64), (9, 8, 63)]


def main():
    lines = open(SRC).read().split("\n")
    c0 = next(i for i, l in enumerate(lines) if "let c_str" in l)
    m0 = next(i for i, l in enumerate(lines) if "let m_str" in l)
    end = next(i for i, l in enumerate(lines) if "(c_str, m_str)" in l)
    nums = lambda a, b: [int(x) for x in re.findall(r'"(\d+)"', "\n".join(lines[a:b]))]
    c, m = nums(c0, m0), nums(m0, end)
    out = {"source": "utils/tests/poseidon_constants.rs (c_str, m_str)", "sets": []}
    ci = mi = 0
    for t, rf, rp in PARAMS:
        nc, nm = (rf + rp) * t, t * t
        ark, mds = c[ci:ci + nc], m[mi:mi + nm]
        ci, mi = ci + nc, mi + nm
        h = hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in ark + mds)).hexdigest()
        out["sets"].append({"t": t, "rf": rf, "rp": rp, "n_round_constants": nc, "first_round_constant": str(ark[0]),
                            "mds_00": str(mds[0]), "sha256": h})
    assert ci == len(c) and mi == len(m), (ci, len(c), mi, len(m))
    json.dump(out, open(os.path.join(HERE, "poseidon_constants_digest.json"), "w"), indent=1)
    print("wrote %d sets, %d round constants, %d matrix entries" % (len(PARAMS), len(c), len(m)))


if __name__ == "__main__":
    main()
