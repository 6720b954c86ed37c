# sums re-associated by arrival time (witness_sched.cpp): same-box A/B of the single-proof latency + parity
run() { echo "$*"; env "$@" timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1 | cut -c1-150; }
run RLNAMD_WL_REASSOC=0
run RLNAMD_WL_REASSOC=1
run RLNAMD_WL_REASSOC=0
run RLNAMD_WL_REASSOC=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py tests/test_gpu_ffi_v3.py -x -q 2>&1 | tail -3
timeout 200 python tools/single_timeline.py 2>/dev/null | tail -40
