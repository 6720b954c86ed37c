mkdir -p gpurun_out/r3g
RLNAMD_WITLANES_INFO=1 timeout 300 python -c "
import __graft_entry__ as g
g.smoke()
" > gpurun_out/r3g/smoke.log 2>&1; tail -3 gpurun_out/r3g/smoke.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or witness or interpreters or circuits or edge or batch_vs" > gpurun_out/r3g/parity.log 2>&1; tail -3 gpurun_out/r3g/parity.log
timeout 300 python tools/ffi_latency.py 2>&1 | head -1
RLNAMD_WITROWS=0 timeout 300 python tools/ffi_latency.py 2>&1 | head -1
