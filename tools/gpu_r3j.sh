mkdir -p gpurun_out/r3j
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py tests/test_gpu_ffi_v3.py -x -q > gpurun_out/r3j/tests.log 2>&1; grep -E "passed|failed" gpurun_out/r3j/tests.log | tail -2
for v in 1 0; do RLNAMD_FUSED_SMUL=$v timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3j/prof -- python3 tools/single_latency.py > gpurun_out/r3j/lat.log 2>&1
f=$(find gpurun_out/r3j/prof -name "*kernel_trace.csv" | head -1); python3 tools/single_timeline.py $f > gpurun_out/r3j/timeline.txt; grep -v ntt_pass gpurun_out/r3j/timeline.txt
find gpurun_out/r3j -name "*.csv" -size +4M -delete
