mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py -x -q > gpurun_out/r3e/tests.log 2>&1; tail -3 gpurun_out/r3e/tests.log
timeout 300 python tools/ffi_latency.py 2>&1 | head -1
RLNAMD_EARLY_WALK=0 timeout 300 python tools/ffi_latency.py 2>&1 | head -1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3e/prof -- python3 tools/single_latency.py > gpurun_out/r3e/lat.log 2>&1; tail -1 gpurun_out/r3e/lat.log
f=$(find gpurun_out/r3e/prof -name "*kernel_trace.csv" | head -1); python3 tools/single_timeline.py $f > gpurun_out/r3e/timeline.txt; cat gpurun_out/r3e/timeline.txt; find gpurun_out/r3e -name "*.csv" -size +4M -delete
