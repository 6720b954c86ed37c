#!/usr/bin/env python3
"""Non-pipelined stage times (each batch runs alone): python tools/stage_times.py [batch] [window_bits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import config2_witnesses
from zerokit_amd.batch import BatchProver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
c = int(sys.argv[2]) if len(sys.argv) > 2 else 13
p = BatchProver(max_batch=B, window_bits=c)
ws, rs = config2_witnesses(B)
n = p.upload(p.pack_inputs(ws), rs)
for _ in range(2):
    p.run(n)
acc = {}
for _ in range(3):
    p.run(n)
    for k, v in p.stage_ms().items():
        acc[k] = acc.get(k, 0) + v / 3
print({k: round(v, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()), 2))
