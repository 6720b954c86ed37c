# row-form products over sixteen lanes (wl_row_mul_add16): same-box A/B + parity
run() { echo "$*"; env "$@" timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1 | cut -c1-150; }
run RLNAMD_WITROW16=0
run RLNAMD_WITROW16=1
run RLNAMD_WITROW16=0
run RLNAMD_WITROW16=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py tests/test_gpu_ffi_v3.py -x -q 2>&1 | tail -3
timeout 200 python tools/single_timeline.py 2>&1 | tail -40
