# stream of mid-size batches after the lone rule covers the fused plan; lone latency; parity
timeout 600 python tools/midstream.py 2>/dev/null | tail -1
timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py tests/test_gpu_ffi_v3.py tests/test_gpu_stream_pool.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
