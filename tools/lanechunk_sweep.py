import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zerokit_amd.batch import BatchProver
sys.path.insert(0, os.path.join(ROOT, "tests"))
cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]
import test_gpu_parity as T
ws = [T._w(c) for c in cases]; rs = [(int(c["r"]), int(c["s"])) for c in cases]
p = BatchProver(max_batch=64)
for n in (8, 16, 24, 32, 48, 64):
    W = (ws * 64)[:n]; R = (rs * 64)[:n]
    p.prove(W, R)
    t0 = time.perf_counter()
    for _ in range(5): p.prove(W, R)
    print("LANECHUNK=%s n=%2d  %.2f ms" % (os.environ.get("RLNAMD_LANECHUNK"), n, (time.perf_counter() - t0) / 5 * 1e3))
p.close()
