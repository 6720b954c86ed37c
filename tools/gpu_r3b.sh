# round 3, step B: FFI streaming + path-emission kernel
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_gpu_ffi.py tests/test_gpu_ffi_v3.py -x -q > gpurun_out/r3b/ffi_tests.log 2>&1; tail -3 gpurun_out/r3b/ffi_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "merkle or tree or config3 or poseidon" > gpurun_out/r3b/tree_tests.log 2>&1; tail -3 gpurun_out/r3b/tree_tests.log
for m in "RLNAMD_PATHS=1" "RLNAMD_PATHS=2 RLNAMD_PATHS_NT=1" "RLNAMD_PATHS=2 RLNAMD_PATHS_NT=0"; do echo "--- $m"; env $m timeout 120 python bench.py --workload merkle --steps 10 2>/dev/null; done
timeout 600 python tools/ffi_latency.py --batch 8192 2>&1 | tail -2
