timeout 300 python tools/lanechunk_sweep.py 2>/dev/null | tail -1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
