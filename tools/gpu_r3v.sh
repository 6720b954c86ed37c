# numbers for the docs: FFI latency, timeline, batch-size sweep (default thresholds / lanes = proofs walks)
run() { echo "$*"; env "$@" timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1; }
run A=1
run A=2
bash tools/gpu_r3q.sh 2>&1 | tail -36
timeout 300 python tools/lanechunk_sweep.py 2>/dev/null | tail -1
RLNAMD_LANECHUNK=0 timeout 300 python tools/lanechunk_sweep.py 2>/dev/null | tail -1
RLNAMD_LANECHUNK=64 timeout 300 python tools/lanechunk_sweep.py 2>/dev/null | tail -1
