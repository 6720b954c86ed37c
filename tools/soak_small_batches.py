#!/usr/bin/env python3
"""Soak of the small-batch shapes (needs the GPU): thousands of DISTINCT witnesses proved one, two, three ... at a time
on the default tables, every proof verified on the host (pairing, all cores) and every public-input vector compared with
the same witnesses proved in one throughput-shaped batch.  Prints one JSON line.  N=<proofs> (default 6000), WBITS=<schedule> (default: the default tables)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zerokit_amd import workload  # noqa: E402
from zerokit_amd.batch import BatchProver, resource_paths, verify_many_with_zkey  # noqa: E402

N = int(os.environ.get("N", "6000"))
p = BatchProver(max_batch=1024, window_bits=int(os.environ.get("WBITS", "0")))   # WBITS=7150114: the bench tables
inp, rsb = workload.config2_packed(p.slots, p.inputs_size, 100000, N)
isz = p.inputs_size * 32
t0 = time.time()
# reference: the throughput shape, 1024 at a time
ref_proofs, ref_vals = [], []
for lo in range(0, N, 1024):
    n = min(1024, N - lo)
    t, _ = p.submit(inp[lo * isz:(lo + n) * isz], rsb[lo * 64:(lo + n) * 64])
    pr, va, er = p.collect_raw(t, n)
    assert not any(er)
    ref_proofs += [pr[128 * i:128 * i + 128] for i in range(n)]
    ref_vals += [va[160 * i:160 * i + 160] for i in range(n)]
sizes = [1, 1, 1, 2, 1, 3, 1, 4, 1, 5, 1, 1, 7, 1, 16, 1, 33]
lo, k, mism = 0, 0, 0
while lo < N:
    n = min(sizes[k % len(sizes)], N - lo)
    k += 1
    t, _ = p.submit(inp[lo * isz:(lo + n) * isz], rsb[lo * 64:(lo + n) * 64])
    pr, va, er = p.collect_raw(t, n)
    for i in range(n):
        if er[i] or pr[128 * i:128 * i + 128] != ref_proofs[lo + i] or va[160 * i:160 * i + 160] != ref_vals[lo + i]:
            mism += 1
    lo += n
t1 = time.time()
zkey = open(resource_paths(20, False)[0], "rb").read()
pubs = [[int.from_bytes(v[32 * j:32 * j + 32], "little") for j in range(5)] for v in ref_vals]
ok = verify_many_with_zkey(zkey, ref_proofs, pubs, threads=0)
print(json.dumps({"proofs": N, "calls": k, "small_vs_throughput_shape_mismatches": mism, "verified": sum(ok), "not_verified": N - sum(ok),
                  "prove_s": round(t1 - t0, 1), "verify_s": round(time.time() - t1, 1)}))
p.close()
