# streams of mid-size batches: walk form threshold
RLNAMD_LANECHUNK_WALK=128 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
RLNAMD_LANECHUNK_WALK=48 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
RLNAMD_LANECHUNK_WALK=0 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
