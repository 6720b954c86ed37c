# stream of mid-size batches: the lone-batch rule (RLNAMD_LONE=-1 detect / 0 never / 1 always) + parity
timeout 600 python tools/midstream.py 2>/dev/null | tail -1
RLNAMD_LONE=0 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
RLNAMD_LONE=1 timeout 600 python tools/midstream.py 2>/dev/null | tail -1
timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1
RLNAMD_LONE=0 timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1
RLNAMD_LONE=0 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py tests/test_gpu_ffi_v3.py tests/test_gpu_stream_pool.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
