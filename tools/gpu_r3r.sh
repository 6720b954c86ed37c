# timeline + latency + parity after: marks off for small batches, lanes = values v29_to_fr, wave-per-long-row mat-vec
bash tools/gpu_r3q.sh 2>&1 | tail -40
run() { echo "$*"; env "$@" timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1 | cut -c1-150; }
run A=1
run RLNAMD_MARKS_SMALL=1
run A=2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py tests/test_gpu_ffi_v3.py -x -q 2>&1 | tail -3
