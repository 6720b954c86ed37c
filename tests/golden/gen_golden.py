#!/usr/bin/env python3
"""Generates tests/golden/rln_h20_vectors.json with the Python oracle (oracle/pyref), which is pinned to the
reference's KATs by tests/test_oracle_kats.py.  Run from the repo root: python tests/golden/gen_golden.py
Vectors: config 1 (bench witness, r=44 s=77), SURVEY Appendix D, and the first 3 config-2 witnesses.
Each holds inputs, blinding, public values, digests of the full witness / h, and the proof."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyref import arkzkey, groth16, rln, workload, wtns_graph  # noqa: E402


def digest(vals):
    return hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in vals)).hexdigest()


def main():
    zk, g = rln.load_circuit(20)
    cases = []
    w1, rs1, root1 = workload.config1_witness()
    cases.append(("config1_bench_witness", w1, rs1))
    cases.append(("survey_appendix_d", dict(identity_secret=12345, user_message_limit=100, message_id=1,
                                            path_elements=[0] * 20, identity_path_index=[0] * 20, x=42,
                                            external_nullifier=100), (44, 77)))
    ws, rs = workload.config2_witnesses(3)
    for i in range(3):
        cases.append(("config2_%d" % i, ws[i], rs[i]))
    # r == 0 exercises the g1_b = 0 branch of partial_proof.rs:242-248
    cases.append(("config2_0_r_zero", ws[0], (0, rs[0][1])))
    out = []
    for name, w, (r, s) in cases:
        wi = rln.WitnessInput(w["identity_secret"], w["user_message_limit"], w["message_id"], w["path_elements"],
                              w["identity_path_index"], w["x"], w["external_nullifier"])
        full = wtns_graph.calc_witness(g, wi.named_inputs())
        h = groth16.witness_map(zk, full)
        proof = groth16.prove(zk, full, r, s)
        vals = rln.proof_values_from_witness(wi)
        pub = rln.public_inputs(vals)
        assert full[1:6] == pub
        assert groth16.verify(zk, proof, pub), name
        A, B, C = proof
        out.append(dict(
            name=name,
            witness={k: (list(map(str, v)) if isinstance(v, list) else str(v)) for k, v in w.items()},
            r=str(r), s=str(s),
            public_inputs=[str(v) for v in pub],
            witness_sha256=digest(full), h_sha256=digest(h),
            h_first=[str(h[0]), str(h[1])], h_last=str(h[-1]),
            proof_compressed=arkzkey.proof_compress(A, B, C).hex(),
            a=[str(A[0]), str(A[1])], b=[[str(B[0][0]), str(B[0][1])], [str(B[1][0]), str(B[1][1])]],
            c=[str(C[0]), str(C[1])],
            rln_proof_le=rln.rln_proof_to_bytes_le(proof, vals).hex(),
        ))
        print(name, "ok")
    json.dump(dict(circuit="tree_depth_20/rln_final.arkzkey + graph.bin", tree_root_config1=str(root1), cases=out),
              open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
