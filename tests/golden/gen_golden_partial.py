#!/usr/bin/env python3
"""Generates tests/golden/rln_h20_partial.json with the Python oracle (oracle/pyref): the partial proofs
(generate_partial_zk_proof, /root/reference/rln/src/protocol/proof.rs:783-803 over partial_proof.rs:108-179) of
two witnesses of rln_h20_vectors.json, the known-signal mask of iden3calc/graph.rs:274-312 as a digest, and the
check that finishing them (partial_proof.rs:182-274) gives the full proofs of that file -- the equality the
reference tests in rln/tests/protocol.rs:222-248.  Run from the repo root: python tests/golden/gen_golden_partial.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyref import arkzkey, groth16, rln, wtns_graph  # noqa: E402


def le(v):
    return int(v).to_bytes(32, "little")


def main():
    zk, g = rln.load_circuit(20)
    mask = groth16.known_mask(g)
    cases = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]}
    out = []
    for name in ("survey_appendix_d", "config2_0_r_zero"):
        c = cases[name]
        w = c["witness"]
        wi = rln.WitnessInput(int(w["identity_secret"]), int(w["user_message_limit"]), int(w["message_id"]),
                              [int(t) for t in w["path_elements"]], [int(t) for t in w["identity_path_index"]],
                              int(w["x"]), int(w["external_nullifier"]))
        full = wtns_graph.calc_witness(g, wi.named_inputs())
        pi_a, rho, pi_b, pi_c = groth16.prove_partial(zk, full, mask)
        A, B, C = groth16.finish_partial(zk, (pi_a, rho, pi_b, pi_c), full, mask, int(c["r"]), int(c["s"]))
        assert arkzkey.proof_compress(A, B, C).hex() == c["proof_compressed"], name
        blob = le(pi_a[0]) + le(pi_a[1]) + le(rho[0]) + le(rho[1]) + le(pi_b[0][0]) + le(pi_b[0][1]) + le(pi_b[1][0]) + \
            le(pi_b[1][1]) + le(pi_c[0]) + le(pi_c[1])
        out.append(dict(name=name, partial320=blob.hex()))
        print(name, "ok")
    json.dump(dict(circuit="tree_depth_20/rln_final.arkzkey + graph.bin",
                   layout="pi_a x,y | rho x,y | pi_b x.c0,x.c1,y.c0,y.c1 | pi_c x,y, canonical 32-byte LE (rlnamd_prover_download_partial)",
                   mask_known=sum(mask), mask_len=len(mask), mask_sha256=hashlib.sha256(bytes(int(b) for b in mask)).hexdigest(),
                   cases=out),
              open(os.path.join(ROOT, "tests", "golden", "rln_h20_partial.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
