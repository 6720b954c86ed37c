# stream of mid-size batches: which of the latency changes cost stream throughput?
timeout 600 python tools/midstream.py 2>/dev/null | tail -1
RLNAMD_MSM_CHUNK_G2_SMALL=4 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
RLNAMD_NTT_FUSE9=0 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
RLNAMD_VALUES_WITNESS=0 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
RLNAMD_WL_REASSOC=0 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
RLNAMD_MARKS_SMALL=1 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
