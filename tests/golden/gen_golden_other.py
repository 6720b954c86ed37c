#!/usr/bin/env python3
"""Generates tests/golden/rln_other_circuits.json: one proof each for the depth-10 single-message circuit and
the depth-20 multi-message-id (max_out 4) circuit, with the Python oracle.  Run from the repo root."""
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
    g = workload.SplitMix64(0xD10)
    out = []
    # depth 10, single
    zk, gr = rln.load_circuit(10)
    w = dict(identitySecret=[g.fr()], userMessageLimit=[50], messageId=[7], pathElements=[g.fr() for _ in range(10)],
             identityPathIndex=[g.next() & 1 for _ in range(10)], x=[g.fr()], externalNullifier=[g.fr()])
    r, s = g.fr(), g.fr()
    full = wtns_graph.calc_witness(gr, w)
    proof = groth16.prove(zk, full, r, s)
    pub = full[1:zk.num_instance_variables]
    vals = rln.proof_values_from_witness(rln.WitnessInput(w["identitySecret"][0], 50, 7, w["pathElements"],
                                                          w["identityPathIndex"], w["x"][0], w["externalNullifier"][0]))
    assert rln.public_inputs(vals) == pub and groth16.verify(zk, proof, pub)
    out.append(dict(name="depth10_single", depth=10, multi=False, inputs={k: [str(v) for v in vs] for k, vs in w.items()},
                    r=str(r), s=str(s), public=[str(v) for v in pub], witness_sha256=digest(full),
                    proof_compressed=arkzkey.proof_compress(*proof).hex()))
    print("depth10 ok")
    # depth 20, multi max_out 4: two of four message ids active
    zk, gr = rln.load_circuit(20, multi=True)
    sel = [1, 0, 1, 0]
    w = dict(identitySecret=[g.fr()], userMessageLimit=[100], messageId=[3, 0, 9, 0], selectorUsed=sel,
             pathElements=[g.fr() for _ in range(20)], identityPathIndex=[g.next() & 1 for _ in range(20)],
             x=[g.fr()], externalNullifier=[g.fr()])
    r, s = g.fr(), g.fr()
    full = wtns_graph.calc_witness(gr, w)
    pub = full[1:zk.num_instance_variables]
    want = rln.proof_values_multi(w["identitySecret"][0], 100, w["messageId"], sel, w["pathElements"],
                                  w["identityPathIndex"], w["x"][0], w["externalNullifier"][0])
    assert pub == want, "multi circuit outputs differ from witness.rs:777-802"
    proof = groth16.prove(zk, full, r, s)
    assert groth16.verify(zk, proof, pub)
    out.append(dict(name="depth20_multi_max_out_4", depth=20, multi=True,
                    inputs={k: [str(v) for v in vs] for k, vs in w.items()}, r=str(r), s=str(s),
                    public=[str(v) for v in pub], witness_sha256=digest(full),
                    proof_compressed=arkzkey.proof_compress(*proof).hex()))
    print("multi ok")
    json.dump(dict(cases=out), open(os.path.join(ROOT, "tests", "golden", "rln_other_circuits.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
