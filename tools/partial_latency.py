import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from zerokit_amd import workload
from zerokit_amd.batch import BatchProver
p = BatchProver(max_batch=64)
ws, rs = workload.config2_range(0, 1)
pin = p.pack_inputs([dict(ws[0], message_id=0, x=0, external_nullifier=0)])
ts = []
for i in range(14):
    t0 = time.perf_counter()
    t, _ = p.submit(pin, bytes(64), 1)
    pp, hs, _ = p.collect_partial_cached(t, 1)
    ts.append((time.perf_counter() - t0) * 1e3)
    p.sync()
    p.release_partial(hs)
print(json.dumps({"partial_ms": sorted(ts[3:])[len(ts[3:]) // 2], "min": min(ts)}))
p.close()
