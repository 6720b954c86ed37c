# interpreter without per-step VMEM (staged stores, two descriptor banks): latency + parity
run() { echo "$*"; env "$@" timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1 | cut -c1-150; }
run A=1
run A=2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py tests/test_gpu_ffi_v3.py -x -q 2>&1 | tail -3
