#!/bin/bash
# round 2, step J: lane = chunk walks for small batches: parity, single-proof stage times and FFI latency
mkdir -p gpurun_out/r2j
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2j/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
RLNAMD_LANECHUNK=0 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/ffi_latency.py > gpurun_out/r2j/ffi_latency.txt 2>&1; tail -8 gpurun_out/r2j/ffi_latency.txt
RLNAMD_LANECHUNK=0 timeout 300 python tools/ffi_latency.py > gpurun_out/r2j/ffi_latency_off.txt 2>&1; tail -8 gpurun_out/r2j/ffi_latency_off.txt
