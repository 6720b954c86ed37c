run() { echo "$*"; env "$@" timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1 | cut -c1-150; }
run A=1
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=12
run GPU_MAX_HW_QUEUES=2
run A=1
run GPU_MAX_HW_QUEUES=8
